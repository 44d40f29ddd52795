"""Stand-in for reference train_vanilla_3dgs.py: the script's own import lines (:16-18) and its loop body (:55-115) statement
by statement — logging, saving and the densification branch (not due in the measured iterations) left out, cameras cycled
through a fixed permutation instead of randint — starting from a checkpoint 13-tuple as `--start_checkpoint` does (:38-40).
Whether `render`, `GaussianModel`, `l1_loss` and `ssim` below are this checkout's torch formulations or this repo's fast path
is decided by the import system alone (w3d_amd.dropin)."""
import torch
from utils.loss_utils import l1_loss, ssim
from gaussian_renderer import render
from scene import GaussianModel


def training(model_params, opt, pipe, cams, background, perm, first_iter, n_steps, gaussians=None, iter_events=None,
             losses=None):
    if gaussians is None:
        gaussians = GaussianModel(3)
        gaussians.restore(model_params, opt)
    ema_loss_for_log = 0.0
    for iteration in range(first_iter, first_iter + n_steps):
        gaussians.update_learning_rate(iteration)
        if iteration % 1000 == 0:
            gaussians.oneupSHdegree()
        viewpoint_cam = cams[perm[(iteration - 1) % len(cams)]]
        if iter_events is not None:                 # iter_start.record(), :56
            iter_events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            iter_events[-1][0].record()
        render_pkg = render(viewpoint_cam, gaussians, pipe, background)
        image, viewspace_point_tensor, visibility_filter, radii = (render_pkg["render"], render_pkg["viewspace_points"],
                                                                   render_pkg["visibility_filter"], render_pkg["radii"])
        gt_image = viewpoint_cam.original_image.cuda()
        Ll1 = l1_loss(image, gt_image)
        loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim(image, gt_image))
        loss.backward()
        if iter_events is not None:                 # iter_end.record(), :82
            iter_events[-1][1].record()
        with torch.no_grad():
            ema_loss_for_log = 0.4 * loss.item() + 0.6 * ema_loss_for_log
            if losses is not None:
                losses.append(loss.item())
            if iteration < opt.densify_until_iter:
                gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter],
                                                                     radii[visibility_filter])
                gaussians.add_densification_stats(viewspace_point_tensor, visibility_filter)
            if iteration < opt.iterations:
                gaussians.optimizer.step()
                gaussians.optimizer.zero_grad(set_to_none=True)
    return gaussians, ema_loss_for_log
