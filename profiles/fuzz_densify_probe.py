"""Differential fuzzing of densify_and_prune (round 4): the same GaussianModel code on the CPU (torch gathers: the stand-in pinned
to the reference's own densify_and_prune by tests/golden/densify.npz) and on the GPU (csrc/w3d_densify.hip's one-pass
compaction) with random parameters, moments, statistics (zero denominators included: the reference's NaN -> 0 rule), thresholds
from "nothing is selected" to "everything is split / pruned", ragged P, with and without the screen-size test.  The split
samples come from the CPU random stream on both sides.  Row count, order, parameters, both moments, which_object bit-identical;
the split children's positions and log-scales (a bmm / log on the device) to 2e-6 of the scene's size; statistics reset; flat layout aligned.
  usage: python3 profiles/fuzz_densify_probe.py [cases] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from w3d_amd.synth import make_scene
from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
import cpu_twins                 # the CPU side of this comparison is the torch stand-in of the compaction (tests/cpu_twins.py: the product
cpu_twins.install()              # has no CPU path), pinned to the reference's own densify_and_prune by tests/golden/densify.npz

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
real_normal = torch.normal


def cpu_stream_normal(mean, std, **kw):
    return real_normal(mean=mean.cpu(), std=std.cpu(), **kw).to(std.device)


torch.normal = cpu_stream_normal
bad = 0
sizes = []
for case in range(cases):
    rs = np.random.RandomState(seed0 + case)
    g = torch.Generator().manual_seed(seed0 + case)
    P = int(rs.choice([1, 2, 63, 64, 65, 255, 1000, 4097, 20001]))
    sc = make_scene(P, seed=seed0 + case, scale_mean=float(rs.choice([0.003, 0.03, 0.3])))
    sc.opacity[:] = torch.empty(P, 1).normal_(float(rs.choice([-6.0, 0.0, 4.0])), 2.0, generator=g)
    moments = [torch.randn(P * 59 + 64, generator=g) for _ in range(2)]
    accum = torch.rand(P, 1, generator=g) * float(rs.choice([1e-5, 1e-3, 1.0]))
    denom = torch.randint(0, 3, (P, 1), generator=g).float()
    radii = torch.rand(P, generator=g) * 40
    which = torch.randint(0, 5, (P, 1), generator=g).int()
    max_grad = float(rs.choice([0.0, 1e-6, 2e-4, 1e9]))
    min_opacity = float(rs.choice([0.0, 0.005, 0.5, 2.0]))
    extent = float(rs.choice([0.01, 1.0, 100.0]))
    mss = None if rs.rand() < 0.5 else 20
    out = []
    for dev in ("cpu", "cuda:0"):
        m = GaussianModel(3, device=torch.device(dev))
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.training_setup(OptimizationParams())
        n = m.flat.numel()
        m.optimizer.exp_avg.copy_(moments[0][:n]); m.optimizer.exp_avg_sq.copy_(moments[1][:n].abs())
        m.zero_padding() if hasattr(m, "zero_padding") else None
        m.xyz_gradient_accum.copy_(accum); m.denom.copy_(denom); m.max_radii2D.copy_(radii); m._which_object = which.clone().to(dev)
        torch.manual_seed(seed0 + case)
        try:
            m.densify_and_prune(max_grad, min_opacity, extent, mss)
            err = None
        except Exception as e:                          # (an empty model is a legal outcome only if both sides agree on it)
            err = type(e).__name__ + ": " + str(e)[:100]
        out.append((m, err))
    (mc, ec), (mg, eg) = out
    msgs = []
    if (ec is None) != (eg is None):
        msgs.append(f"cpu error {ec!r} vs gpu error {eg!r}")
    elif ec is None:
        if mc.num_points != mg.num_points:
            msgs.append(f"P {mc.num_points} vs {mg.num_points}")
        else:
            sizes.append((P, mc.num_points))
            momc, momg = mc.optimizer.moments(), mg.optimizer.moments()
            for nme in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
                a, b = mc._p[nme].detach(), mg._p[nme].detach().cpu()
                if nme in ("xyz", "scaling"):
                    if a.numel() and float((a - b).abs().max()) > 2e-6 * max(1.0, float(a.abs().max())): msgs.append(f"{nme}: max diff {float((a - b).abs().max()):.2e}")
                elif not torch.equal(a, b): msgs.append(f"{nme} differs")
                for i, w in enumerate(("m", "v")):
                    if not torch.equal(momc[nme][i], momg[nme][i].cpu()): msgs.append(f"{w}_{nme} differs")
            if not torch.equal(mc._which_object, mg._which_object.cpu()): msgs.append("which_object differs")
            for nme in ("xyz_gradient_accum", "denom", "max_radii2D"):
                if not torch.equal(getattr(mc, nme), getattr(mg, nme).cpu()): msgs.append(f"{nme} differs")
            if mg.num_points and any(a % 4 for a, _ in mg.block_slices().values()): msgs.append("a block is not 16-byte aligned")
    if msgs:
        bad += 1
        print(f"case {seed0 + case} (P={P}, max_grad {max_grad}, min_opacity {min_opacity}, extent {extent}, screen test {mss}): " + "; ".join(msgs[:6]), flush=True)
grow = sum(1 for a, b in sizes if b > a); shrink = sum(1 for a, b in sizes if b < a); same = sum(1 for a, b in sizes if a == b); empty = sum(1 for a, b in sizes if b == 0)
print(f"cases {cases} from seed {seed0}: {bad} with differences ({grow} grew, {shrink} shrank, {same} kept their size, {empty} ended empty)")
