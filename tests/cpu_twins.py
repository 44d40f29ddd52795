"""Torch stand-ins for the HIP-backed operations of w3d_amd, for the host-LOGIC tests that run without a GPU (`-m "not gpu"`:
block bookkeeping of FlatAdam / GaussianModel, schedules, the view-parallel protocol over gloo).  The product has no CPU path
(w3d_amd/_host_twins.py is an empty registry; every operation refuses CPU tensors); tests/conftest.py calls install() so that the
CPU suites can drive the host classes on CPU tensors.  The GPU suites hold the kernels against these same formulas
(tests/test_gpu_fused.py, tests/test_gpu_dist.py), so each formula exists once — here.

Each function restates what its kernel computes:
  adam_step                 torch.optim.Adam's update (no weight decay / amsgrad), reference scene/gaussian_model.py:172-182
  photometric_loss          0.8 * L1 + 0.2 * (1 - SSIM), reference train_vanilla_3dgs.py:77-79, utils/loss_utils.py:17-63
  add_densification_stats   reference scene/gaussian_model.py:461-463 as masked arithmetic
  densify_compact           row compaction of the flat parameter buffer and both moments (csrc/w3d_densify.hip)
  sh_adam_lowrank           dL/dSH[k] = sum_v basis_k(dir_v) * dL/dRGB_v in view order, then Adam on f_dc / f_rest
  pack / apply_gradient_rows  the 64-B non-zero gradient rows of the sparse exchange (include/w3d.h)
  track_visibility          per-rank visibility counts and radii maxima (Trainer.track_local)"""
import math

import torch


def adam_step(opt, p, g, m, v, slices, zero_grad):
    b1, b2 = opt.betas
    for name, (a, b, stepped) in slices.items():
        gg = g[a:b]
        if not stepped:
            if zero_grad:
                gg.zero_()
            continue
        bc1, bc2 = opt.bias_corrections(name)
        m[a:b].mul_(b1).add_(gg, alpha=1 - b1)
        v[a:b].mul_(b2).addcmul_(gg, gg, value=1 - b2)
        denom = (v[a:b].sqrt() / math.sqrt(bc2)).add_(opt.eps)
        p[a:b].addcdiv_(m[a:b], denom, value=-opt.lrs[name] / bc1)
        if zero_grad:
            gg.zero_()


def photometric_loss(image, gt, lambda_dssim=0.2):
    from w3d_amd.loss import photometric_loss_torch
    return photometric_loss_torch(image, gt, lambda_dssim)


def add_densification_stats(model, g, update_filter):
    norm = torch.norm(g[:, :2], dim=-1, keepdim=True)
    model.xyz_gradient_accum += torch.where(update_filter.reshape(-1, 1), norm, torch.zeros_like(norm))
    model.denom += update_filter.reshape(-1, 1).to(model.xyz_gradient_accum.dtype)


def densify_compact(model, names, dims, layout_old, layout_new, P_old, P_new, src, n_keep, n_child0, new_store, m_new, v_new,
                    child_xyz, child_scaling):
    opt = model.optimizer
    src64 = src.to(torch.int64)
    for name, d in zip(names, dims):
        off_o, off_n = layout_old[name][0], layout_new[name][0]

        def rows(buf):
            return buf[off_o:off_o + P_old * d].view(P_old, d).index_select(0, src64)
        blk = rows(model.flat.detach())
        if n_child0 < P_new and name == "xyz":
            blk[n_child0:] = child_xyz
        if n_child0 < P_new and name == "scaling":
            blk[n_child0:] = child_scaling
        new_store[off_n:off_n + P_new * d] = blk.reshape(-1)
        if m_new is not None:
            for old, new_ in ((opt.exp_avg, m_new), (opt.exp_avg_sq, v_new)):
                mb = rows(old)
                mb[n_keep:] = 0
                new_[off_n:off_n + P_new * d] = mb.reshape(-1)


def sh_adam_lowrank(model, dcolor_all, campos_all, skip, rows):
    from w3d_amd.fused_step import SH_BLOCKS
    from w3d_amd.sh import sh_basis
    if rows is not None:
        raise RuntimeError("row chunks are a GPU-path feature")
    P, V = model.num_points, int(dcolor_all.shape[0])
    deg = int(model.active_sh_degree)
    xyz = model._p["xyz"].detach()
    grad = torch.zeros(P, 16, 3, dtype=torch.float32)
    for v in range(V):                                   # view order, as in the kernel
        dirs = xyz - campos_all[v][None]
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        basis = sh_basis(deg, dirs)                      # (P, (deg+1)^2)
        grad[:, :basis.shape[1]] += basis[:, :, None] * dcolor_all[v][:, None, :]
    model.grad_view("f_dc").copy_(grad[:, :1])
    model.grad_view("f_rest").copy_(grad[:, 1:])
    # (the bucket was written directly just above: .grad is not consulted — after a densification every p.grad is None)
    model.optimizer.step(only=SH_BLOCKS, skip=skip, advance=False, respect_none_grads=False)


def pack_gradient_rows(model, dcolor, grad2d_norm, norm_scale, rows, count):
    from w3d_amd.fused_step import GEO_BLOCKS
    P = model.num_points
    gn = torch.zeros(P) if grad2d_norm is None else grad2d_norm.reshape(P).float() * norm_scale
    full = torch.cat([gn[:, None], dcolor.reshape(P, 3)] + [model.grad_view(n).reshape(P, -1) for n in GEO_BLOCKS], 1)
    idx = (full != 0).any(1).nonzero()[:, 0]
    n = int(idx.numel())
    rows[:n, 0] = idx.to(torch.int32).view(torch.float32)
    rows[:n, 1:] = full[idx]
    count[0] = n
    return rows, count


def apply_gradient_rows(model, rows, count, max_rows, dcolor_view, norm_sum):
    from w3d_amd.fused_step import GEO_BLOCKS
    P = model.num_points
    n = min(int(count[0]), int(max_rows))
    r = rows[:n]
    idx = r[:, 0].contiguous().view(torch.int32).long()
    if norm_sum is not None:
        norm_sum[idx] += r[:, 1]
    dcolor_view[idx] = r[:, 2:5]
    col = 5
    for name in GEO_BLOCKS:
        blk = model.grad_view(name).view(P, -1)
        blk[idx] += r[:, col:col + blk.shape[1]]
        col += blk.shape[1]


def track_visibility(visible, radii, vis_local, rmax_local):
    vis_local += visible.to(torch.int32)
    torch.maximum(rmax_local, radii.to(torch.int32), out=rmax_local)


def install():
    from w3d_amd import _host_twins
    for name, fn in (("adam_step", adam_step), ("photometric_loss", photometric_loss),
                     ("add_densification_stats", add_densification_stats), ("densify_compact", densify_compact),
                     ("sh_adam_lowrank", sh_adam_lowrank), ("pack_gradient_rows", pack_gradient_rows),
                     ("apply_gradient_rows", apply_gradient_rows), ("track_visibility", track_visibility)):
        _host_twins.register(name, fn)
