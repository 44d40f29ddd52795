"""Do offsets beyond 2^32 bytes work? (round 4)  A 1 M-Gaussian scene A is rendered and back-propagated (deterministic
backward) alone and as the LAST 1 M Gaussians of a 25 M-Gaussian model whose first 24 M sit behind the camera (culled): every
parameter block of the big model puts A's rows behind byte offset 2^32 (f_rest: 24 M x 180 B = 4.3 GB; the flat buffer: 5.9 GB).
Images must be bit-identical, A's gradients bit-identical, the culled Gaussians' gradients exactly zero.
  usage: python3 profiles/big_offset_probe.py [pad_millions]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from w3d_amd.synth import make_scene, make_cameras
from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
from w3d_amd.fused_step import render_raw, backward_raw, backward_raw_adam

pad = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 24_000_000
dev = torch.device("cuda:0")
W, H, PA = 1600, 1200, 1_000_000
cam = make_cameras(36, W, H)[0].to(dev)
a = make_scene(PA, seed=3, scale_mean=0.006)
bg = torch.zeros(3, device=dev)
dimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)


def run(sc_tensors):
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(*sc_tensors)
    m.active_sh_degree = 3
    m.deterministic = True
    m.training_setup(OptimizationParams())
    with torch.no_grad():
        pkg = render_raw(cam, m, bg, sync=True)
        gn, _ = backward_raw(m, pkg["handle"], dimg, want_norm=True)
        # ... and one step of the fused backward + Adam kernel (parameters and both moments updated in place)
        grads = {k: m.flat_grad[lo:hi].clone() for k, (lo, hi) in m.block_slices().items()}
        m.update_learning_rate(1)
        pkg2 = render_raw(cam, m, bg, sync=True, color_only=True)
        backward_raw_adam(m, pkg2["handle"], dimg, want_norm=False, update_stats=True)
        m.optimizer.note_fused_step()
    return m, pkg, gn, grads


tA = (a.xyz, a.features_dc, a.features_rest, a.scaling, a.rotation, a.opacity)
mA, pA, gnA, grA = run(tA)
refs = dict(img=pA["render"].clone(), depth=pA["depth"].clone(), alpha=pA["alpha"].clone(), radii=pA["radii"].clone(), gn=gnA.clone(),
            grads=grA, after={k: (mA.flat[lo:hi].clone(), mA.optimizer.exp_avg[lo:hi].clone(), mA.optimizer.exp_avg_sq[lo:hi].clone())
                              for k, (lo, hi) in mA.block_slices().items()},
            stats=(mA.xyz_gradient_accum.clone(), mA.denom.clone(), mA.max_radii2D.clone()))
R_A = pA["handle"]["num_rendered"]
del mA, pA
torch.cuda.empty_cache()
far = torch.zeros(pad, 3); far[:, 2] = 50.0 + torch.rand(pad)            # behind the overhead cameras (they look down -z)
def cat(x, y): return torch.cat([x, y], 0)
big = (cat(far, a.xyz), cat(torch.zeros(pad, 1, 3), a.features_dc), cat(torch.zeros(pad, 15, 3), a.features_rest),
       cat(torch.full((pad, 3), -5.0), a.scaling), cat(torch.tensor([[1.0, 0, 0, 0]]).repeat(pad, 1), a.rotation), cat(torch.zeros(pad, 1), a.opacity))
mB, pB, gnB, grB = run(big)
P = pad + PA
print(f"P = {P}: flat buffer {mB.flat.numel() * 4 / 2**30:.2f} GiB; visible {int((pB['radii'] > 0).sum())} (alone: {int((refs['radii'] > 0).sum())}); list entries {pB['handle']['num_rendered']} (alone: {R_A})")
ok = True
def check(name, cond):
    global ok
    print(f"  {name}: {'ok' if cond else 'DIFFERENT'}")
    ok = ok and bool(cond)
check("culled prefix invisible", int((pB["radii"][:pad] > 0).sum()) == 0)
check("radii of A", torch.equal(pB["radii"][pad:], refs["radii"]))
for k in ("img", "depth", "alpha"):
    check(f"{k} bit-identical", torch.equal(pB[{"img": "render"}.get(k, k)], refs[k]))
check("densification norm of A bit-identical", torch.equal(gnB[pad:], refs["gn"]))
check("densification norm of the prefix zero", float(gnB[:pad].abs().max()) == 0.0)
for k, (lo, hi) in mB.block_slices().items():
    gB = grB[k].view(P, -1)
    check(f"grad {k}: A's rows bit-identical (byte offset of the first {(lo + pad * gB.shape[1]) * 4 / 2**30:.2f} GiB)", torch.equal(gB[pad:].reshape(-1), refs["grads"][k]))
    check(f"grad {k}: prefix zero", float(gB[:pad].abs().max()) == 0.0)
for k, (lo, hi) in mB.block_slices().items():
    for what, buf, ref in zip(("parameters", "first moment", "second moment"), (mB.flat, mB.optimizer.exp_avg, mB.optimizer.exp_avg_sq), refs["after"][k]):
        check(f"after one fused backward + Adam step, {k} {what}: A's rows bit-identical", torch.equal(buf[lo:hi].view(P, -1)[pad:].reshape(-1), ref))
for name, bufB, bufA in zip(("xyz_gradient_accum", "denom", "max_radii2D"), (mB.xyz_gradient_accum, mB.denom, mB.max_radii2D), refs["stats"]):
    check(f"statistics {name}: A's rows bit-identical, prefix zero", torch.equal(bufB[pad:], bufA) and float(bufB[:pad].abs().max()) == 0.0)
print("RESULT", "ok" if ok else "FAILED")
