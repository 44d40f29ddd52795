"""Adam over the flat 59-float-per-Gaussian parameter buffer.

Same update rule and hyper-parameters as the reference's optimizer — torch.optim.Adam with
lr=0.0 default, eps=1e-15 and one lr per parameter group (scene/gaussian_model.py:172-182) —
but the two moments live in flat buffers with the layout of GaussianModel.flat, so one sweep
updates every block.  `step()` runs the hand-written HIP kernel (csrc/w3d_adam.hip) when the
buffers are on the GPU; on the CPU (unit tests of the host logic) it uses the identical formula
written with torch ops.
"""
import math

import torch


class FlatAdam:
    def __init__(self, model, lrs, betas=(0.9, 0.999), eps=1e-15, moments=None):
        self.model = model
        self.lrs = dict(lrs)
        self.betas = betas
        self.eps = eps
        self.step_count = 0
        if moments is not None and isinstance(moments[0], torch.Tensor):
            # flat moments with the layout of model.flat, adopted as they are (densification's compaction kernel wrote them)
            self.exp_avg, self.exp_avg_sq, self.step_count = moments[0], moments[1], int(moments[2])
            assert self.exp_avg.shape == model.flat.shape and self.exp_avg_sq.shape == model.flat.shape
            return
        self.exp_avg = torch.zeros_like(model.flat)
        self.exp_avg_sq = torch.zeros_like(model.flat)
        if moments is not None:
            new_m, new_v, steps = moments
            for name, (a, b) in model.block_slices().items():
                self.exp_avg[a:b].copy_(new_m[name].reshape(-1))
                self.exp_avg_sq[a:b].copy_(new_v[name].reshape(-1))
            self.step_count = steps

    # -- the pieces of torch.optim.Optimizer the reference's host code touches
    @property
    def param_groups(self):
        return [{"name": n, "lr": lr, "params": [self.model._p[n]]} for n, lr in self.lrs.items()]

    def set_lr(self, name, lr):
        self.lrs[name] = float(lr)

    def moments(self):
        out = {}
        for name, (a, b) in self.model.block_slices().items():
            shape = self.model._p[name].shape
            out[name] = (self.exp_avg[a:b].view(shape), self.exp_avg_sq[a:b].view(shape))
        return out

    def zero_moments(self, name):
        a, b = self.model.block_slices()[name]
        self.exp_avg[a:b].zero_()
        self.exp_avg_sq[a:b].zero_()

    def zero_grad(self, set_to_none=True):
        # gradients are views of one flat buffer (the all-reduce bucket): zero it in place.
        self.model.flat_grad.zero_()

    def state_dict(self):
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.step_count,
                "lrs": dict(self.lrs)}

    def load_state_dict(self, d):
        self.exp_avg.copy_(d["exp_avg"])
        self.exp_avg_sq.copy_(d["exp_avg_sq"])
        self.step_count = int(d["step"])
        self.lrs.update(d["lrs"])

    def note_fused_step(self):
        """The backward kernel applied this step's update itself (fused_step.backward_raw_adam)."""
        self.step_count += 1

    def bias_corrections(self):
        b1, b2 = self.betas
        return 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count

    @torch.no_grad()
    def step(self, zero_grad=False, skip=(), elem_range=None, only=None, advance=True):
        """One Adam step on every block.  zero_grad=True clears the gradient bucket in the same
        sweep (what `optimizer.zero_grad(set_to_none=True)` achieves at train_vanilla_3dgs.py:115).
        elem_range=(lo, hi): only that slice of the flat buffer is stepped — the shard this rank owns in
        the dense view-parallel exchange (the other shards arrive through the parameter all-gather).
        only=names: step just these blocks; advance=False: the step counter was already advanced for this
        iteration (the low-rank exchange steps the geometry blocks and the SH blocks separately)."""
        if advance:
            self.step_count += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        p, g, m, v = self.model.flat, self.model.flat_grad, self.exp_avg, self.exp_avg_sq
        slices = {}
        for name, (a, b) in self.model.block_slices().items():
            if elem_range is not None:
                a, b = max(a, elem_range[0]), min(b, elem_range[1])
            if a < b and (only is None or name in only):
                slices[name] = (a, b)
        if p.is_cuda:
            from .fused import adam_step
            for name, (a, b) in slices.items():
                if name in skip:
                    if zero_grad:
                        g[a:b].zero_()
                    continue
                adam_step(p[a:b], g[a:b], m[a:b], v[a:b], self.lrs[name], b1, b2, self.eps, bc1, bc2, zero_grad)
            return
        for name, (a, b) in slices.items():
            gg = g[a:b]
            if name in skip:
                if zero_grad:
                    gg.zero_()
                continue
            m[a:b].mul_(b1).add_(gg, alpha=1 - b1)
            v[a:b].mul_(b2).addcmul_(gg, gg, value=1 - b2)
            denom = (v[a:b].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p[a:b].addcdiv_(m[a:b], denom, value=-self.lrs[name] / bc1)
            if zero_grad:
                gg.zero_()
