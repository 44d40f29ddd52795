"""Differential fuzzing of the two training-step flavours (round 4): the fused raw-parameter step (activations inside the
kernels, gradients straight into the bucket) against the drop-in render() + autograd step (torch activations, the reference's
marshalling) on random small scenes — random P (ragged on purpose), SH degree, frame, background, shapes from blobs to mild
ellipsoids.  Loss within 1e-6, image within 2e-5, every gradient block of the first step within 2e-4 of its maximum, statistics
(denom, max_radii2D) identical — EXCEPT where a (pixel, Gaussian) pair sits on a blend threshold: the two flavours' activations
differ in the last bit (in-kernel exp / sigmoid / normalize vs torch's), which flips such a pair in a few percent of the scenes
(one to three pixels, image difference up to 1/255 of a colour, one Gaussian's gradient off by a pixel's worth) and, about once
in 400 scenes, the ceil() of a radius.
  usage: python3 profiles/fuzz_step_probe.py [cases] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from w3d_amd.synth import make_scene, make_cameras
from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
from w3d_amd.train import Trainer
import w3d_amd.gaussian_renderer as gr

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
bad = 0
worst = 0.0
for case in range(cases):
    rs = np.random.RandomState(seed0 + case)
    g = torch.Generator().manual_seed(seed0 + case)
    P = int(rs.choice([1, 3, 65, 255, 257, 1000, 5000, 5000 + int(rs.randint(256))]))
    W, H = int(rs.randint(16, 300)), int(rs.randint(16, 220))
    deg = int(rs.randint(4))
    cams = [c.to(dev) for c in make_cameras(4, W, H)]
    for cam in cams:
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
    bg = torch.tensor([float(x) for x in rs.choice([0.0, 0.2], 3)], device=dev)
    sc = make_scene(P, seed=seed0 + case, scale_mean=float(rs.choice([0.01, 0.03, 0.1])))
    if rs.rand() < 0.5:
        sc.scaling[:] = sc.scaling[:, :1] + torch.empty(P, 3).uniform_(-1.0, 1.0, generator=g)
    gr.RAW_AUTOGRAD = False
    res = []
    for fused in (False, True):
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = deg
        m.deterministic = True
        opt = OptimizationParams()
        m.training_setup(opt)
        # (spatial_order=False: the spy reads the gradient bucket before optimizer.step(); after a re-sort the autograd path's first
        #  .grad is a fresh tensor that step() copies into the bucket, as after a densification — the bucket would still be zero here)
        tr = Trainer(m, cams, opt, bg, densify=False, fused=fused, spatial_order=False)
        tr.fused_adam = False
        grads = []
        orig = m.optimizer.step
        def spy(*a, _m=m, _g=grads, _o=orig, **k):
            _g.append(_m.flat_grad.clone())
            return _o(*a, **k)
        m.optimizer.step = spy
        loss = float(tr.step(1))
        res.append((m, loss, tr.last["image"].clone(), grads[0], m.denom.clone(), m.max_radii2D.clone()))
    gr.RAW_AUTOGRAD = True
    (ma, la, ia, ga, da, ra), (mb, lb, ib, gb, db, rb) = res
    msgs = []
    if abs(la - lb) > 1e-6 * max(1.0, abs(la)):
        msgs.append(f"loss {la} vs {lb}")
    if float((ia - ib).abs().max()) > 2e-5:
        msgs.append(f"image max diff {float((ia - ib).abs().max()):.2e} ({int(((ia - ib).abs().max(0)[0] > 2e-5).sum())} pixels beyond 2e-5)")
    radii_same = torch.equal(ra, rb) and torch.equal(da, db)
    for name, (a, b) in ma.block_slices().items():
        ref = ga[a:b]
        e = float((gb[a:b] - ref).abs().max() / (ref.abs().max() + 1e-20))
        worst = max(worst, e)
        if e > 2e-4:
            msgs.append(f"{name}: rel {e:.2e}")
    if not radii_same:
        nd = int((ra != rb).sum())
        msgs.append(f"radii differ on {nd} Gaussians (max |d| {float((ra - rb).abs().max())})")
    if msgs:
        bad += 1
        print(f"case {seed0 + case} (P={P}, {W}x{H}, deg {deg}): " + "; ".join(msgs), flush=True)
print(f"cases {cases} from seed {seed0}: {bad} with differences; worst block error {worst:.2e}")
