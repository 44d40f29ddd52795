"""Host logic around the FlashSplat contribution render: turning the
additive per-(object, Gaussian) counts into labels, as reference run_3d_seg.py:54-72 does
(pinned by tests/golden/multi_instance_opt.npz) — vectorised over objects instead of a Python
loop with one tqdm step per object."""
import torch


def multi_instance_opt(all_contrib: torch.Tensor, gamma: float = 0.0) -> torch.Tensor:
    """all_contrib (K, P) additive counts -> bool (K, P): S[i, j] = Gaussian j belongs to object i.
    For every object the pair (rest, own) = (sum - own, own) is L2-normalised over the pair,
    `gamma` is added to the rest score, and the larger one wins (ties go to "rest")."""
    total = all_contrib.sum(dim=0, keepdim=True)
    rest = total - all_contrib
    norm = torch.sqrt(rest * rest + all_contrib * all_contrib).clamp_min(1e-12)
    return (all_contrib / norm) > (rest / norm + gamma)


def accumulate_counts(render_fn, cameras, masks, obj_num=1):
    """Sum of `used_count` over views (reference run_3d_seg.py:91-97).  render_fn(cam, mask) must
    return the dict of flashsplat_render."""
    total = None
    for cam, mask in zip(cameras, masks):
        with torch.no_grad():
            uc = render_fn(cam, mask)["used_count"]
        total = uc.clone() if total is None else total + uc
    return total
