"""Stand-in for reference utils/loss_utils.py:17-63: mean absolute error, and SSIM as five grouped 11x11 convolutions with a
sigma-1.5 Gaussian window (autograd supplies the backward) — the torch formulation the redirect replaces."""
import math

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return (network_output - gt).abs().mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def _window(size, channel, like):
    g = torch.tensor([math.exp(-((i - size // 2) ** 2) / (2 * 1.5 ** 2)) for i in range(size)])
    g = (g / g.sum())[:, None]
    return (g @ g.t())[None, None].expand(channel, 1, size, size).contiguous().to(like)


def ssim(img1, img2, window_size=11, size_average=True):
    c = img1.size(-3)
    w, pad = _window(window_size, c, img1), window_size // 2
    blur = lambda x: F.conv2d(x, w, padding=pad, groups=c)  # noqa: E731
    m1, m2 = blur(img1), blur(img2)
    v1, v2, v12 = blur(img1 * img1) - m1 * m1, blur(img2 * img2) - m2 * m2, blur(img1 * img2) - m1 * m2
    q = ((2 * m1 * m2 + 0.01 ** 2) * (2 * v12 + 0.03 ** 2)) / ((m1 * m1 + m2 * m2 + 0.01 ** 2) * (v1 + v2 + 0.03 ** 2))
    return q.mean() if size_average else q.mean(1).mean(1).mean(1)
