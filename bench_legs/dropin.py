"""drop-in legs (--full): the UNMODIFIED reference loop script on this repo's modules, with and without the import redirect."""
import ctypes
import json
import math
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

from .common import ROOT


def time_standin(args, sc, cams, bg, dev, perm, hook, n, warm):
    """tests/standin_checkout/train_loop.py — the import lines and the loop body of reference train_vanilla_3dgs.py:16-18,55-115,
    statement by statement, starting from a checkpoint 13-tuple as --start_checkpoint does (:38-40) — timed as a whole
    (loss.item() and the boolean-mask statistics lines, i.e. the reference loop's host syncs, included).  The GPU box has no
    reference checkout, so the script imports its GaussianModel / render / l1_loss / ssim from the stand-in modules of the same
    names (tests/standin_checkout/README.md: six nn.Parameters with torch activations, torch.optim.Adam over six groups,
    render() marshalling into `diff_gaussian_rasterization`, conv2d SSIM).  hook=False: as it is — only the rasterizer packages
    are this repo's (INTEGRATION.md section 1 without the redirect).  hook=True: under w3d_amd.dropin.install() — the same
    unmodified script and modules, the four names redirected to this repo's fast path.  A tuple: only those modules."""
    from util import standin_checkout, checkpoint_tuple
    from w3d_amd.gaussian_model import OptimizationParams
    from w3d_amd.train import PipelineParams
    opt, pipe = OptimizationParams(), PipelineParams()
    with standin_checkout(hook) as loop:
        owners = {k: getattr(loop, k).__module__ for k in ("GaussianModel", "render", "l1_loss", "ssim")}
        g, _ = loop.training(checkpoint_tuple(sc, device=dev), opt, pipe, cams, bg, perm, 1, warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.training(None, opt, pipe, cams, bg, perm, 1 + warm, n, gaussians=g)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the reference's own timer (TensorBoard `iter_time`, train_vanilla_3dgs.py:56,82,149): CUDA/HIP events around
        # render + loss + backward — a separate short run, so the event pairs do not sit in the timed loop above
        ev = []
        loop.training(None, opt, pipe, cams, bg, perm, 1 + warm + n, min(n, 36), gaussians=g, iter_events=ev)
        torch.cuda.synchronize()
        iter_ms = sorted(a.elapsed_time(b) for a, b in ev)
        del g
    torch.cuda.empty_cache()
    return {"iters_per_s": round(n / dt, 2), "ms_per_step": round(1e3 * dt / n, 4), "steps": n,
            "iter_time_ms_median": round(iter_ms[len(iter_ms) // 2], 4), "resolved": owners}


def time_dropin(args, sc, cams, bg, dev, perm, n=None):
    """The unmodified loop script under the import redirect (w3d_amd.dropin.install())."""
    if n is None:
        n = args.steps if args.dropin_steps < 0 else args.dropin_steps
    if n <= 0:
        return None
    # (W3D_SPATIAL_ORDER: the documented switch of the redirect's GaussianModel, INTEGRATION.md section 1 — the model the script
    #  restores from its checkpoint is put into Morton order, as Trainer(spatial_order=True) does for the fused step)
    prev = os.environ.get("W3D_SPATIAL_ORDER")
    if not args.no_spatial_order:
        os.environ["W3D_SPATIAL_ORDER"] = "2"
    try:
        out = time_standin(args, sc, cams, bg, dev, perm, True, n, max(3, min(args.warmup, 10)))
    finally:
        if prev is None:
            os.environ.pop("W3D_SPATIAL_ORDER", None)
        else:
            os.environ["W3D_SPATIAL_ORDER"] = prev
    out["spatial_order"] = not args.no_spatial_order
    assert all(v.startswith("w3d_amd.") for v in out["resolved"].values()), out["resolved"]
    out["iter_time"] = "HIP events around render + loss + backward, the bracket of train_vanilla_3dgs.py:56,82 (no optimizer step)"
    return out


# ------------------------------------------------------------------------------------------------ modules-only loop
def time_modules_only(args, sc, cams, bg, dev, perm):
    """The same loop script WITHOUT the redirect — INTEGRATION.md section 1's first step alone: only the three rasterizer
    packages on the path are this repo's; the model (six nn.Parameters, torch activations, six-group torch.optim.Adam), render()'s
    marshalling and the conv2d SSIM are the checkout's own Python — then with the redirect on for the loss module only, for the
    model + render modules only, and (time_dropin) for all of them."""
    n = min(args.steps, 60) if args.modules_only_steps < 0 else args.modules_only_steps
    if n <= 0:
        return None
    w = max(3, min(args.warmup, 6))
    base = time_standin(args, sc, cams, bg, dev, perm, False, n, w)
    assert not any(v.startswith("w3d_amd.") for v in base["resolved"].values()), base["resolved"]
    out = dict(base, what="tests/standin_checkout/train_loop.py as it is, no import redirect: only diff_gaussian_rasterization is this "
                          "repo's; six nn.Parameters with torch exp / sigmoid / normalize / cat, torch.optim.Adam (6 groups), torch "
                          "conv2d SSIM, the reference loop's host syncs")
    # which of the other swaps buys what (same script, the redirect switched on for one part at a time)
    try:
        out["redirect_loss_module_only_iters_per_s"] = time_standin(args, sc, cams, bg, dev, perm, ("utils.loss_utils",), n, w)["iters_per_s"]
        out["redirect_model_and_render_only_iters_per_s"] = time_standin(args, sc, cams, bg, dev, perm,
                                                                         ("scene.gaussian_model", "gaussian_renderer"), n, w)["iters_per_s"]
    except Exception as e:
        out["breakdown_error"] = repr(e)
    return out

