"""Metamorphic check (round 4): the ORDER of the Gaussians in memory must not matter.  A scene and a random permutation of it
(scenes in which two visible Gaussians share a depth are skipped: there the index breaks the tie, by contract) are rendered and back-propagated with the deterministic
backward: images bit-identical, gradients bit-identical after undoing the permutation.  Exercises the depth sort's index handling
(the first pass takes "the value of key i is i"), the record gather and the per-Gaussian backward on scattered rows.
  usage: python3 profiles/permutation_probe.py [P] [seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from w3d_amd.synth import make_scene, make_cameras
from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
from w3d_amd.fused_step import render_raw, backward_raw

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3001
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
W, H = 800, 608
bad = skipped = 0
for seed in range(seeds):
    sc = make_scene(P, seed=seed, scale_mean=0.03)
    cam = make_cameras(6, W, H)[seed % 6].to(dev)
    bg = torch.tensor([0.1, 0.0, 0.3], device=dev)
    dimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(seed)).to(dev)
    perm = torch.randperm(P, generator=torch.Generator().manual_seed(100 + seed))

    def run(idx):
        m = GaussianModel(3, device=dev)
        t = [x if idx is None else x[idx].contiguous() for x in (sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)]
        m.create_from_tensors(*t)
        m.active_sh_degree = 3
        m.deterministic = True
        m.training_setup(OptimizationParams())
        with torch.no_grad():
            pkg = render_raw(cam, m, bg, sync=True)
            gn, _ = backward_raw(m, pkg["handle"], dimg, want_norm=True)
        return m, pkg, gn
    mA, pA, gA = run(None)
    with torch.no_grad():
        fl = render_raw(cam, mA, bg, sync=True, flash=dict(num_obj=1, gt_mask=torch.zeros(H, W, device=dev)))
    dv = fl["gs_depth"][pA["radii"] > 0]
    if dv.unique().numel() != dv.numel():
        skipped += 1
        continue
    mB, pB, gB = run(perm)
    msgs = []
    for k in ("render", "depth", "alpha"):
        if not torch.equal(pA[k], pB[k]):
            msgs.append(f"{k}: {int((pA[k] != pB[k]).sum())} elements differ")
    pd = perm.to(dev)
    if not torch.equal(pA["radii"][pd], pB["radii"]): msgs.append("radii differ")
    if not torch.equal(gA[pd], gB): msgs.append("densification norms differ")
    for k, (lo, hi) in mA.block_slices().items():
        a = mA.flat_grad[lo:hi].view(P, -1)[pd]
        lo2, hi2 = mB.block_slices()[k]
        b = mB.flat_grad[lo2:hi2].view(P, -1)
        if not torch.equal(a, b):
            msgs.append(f"grad {k}: {int((a != b).any(1).sum())} rows differ (max {float((a - b).abs().max()):.2e})")
    depth = pA["handle"]
    if msgs: print(f"seed {seed}: visible {int((pA['radii'] > 0).sum())}, entries {pA['handle']['num_rendered']}: " + ("ok" if not msgs else "; ".join(msgs)), flush=True)
    bad += bool(msgs)
print("RESULT", "ok" if bad == 0 else f"{bad} seeds differ", f"({skipped} of {seeds} scenes skipped for depth ties)")
