"""Cost of one re-sort of the Gaussians into Morton order (GaussianModel.spatial_permutation + the compaction pass) at 2 M
Gaussians, four times in a row with the positions perturbed in between: python profiles/sort_cost.py
(profiles/r05/ab_spatial_order.txt, block 5)."""
import os
import sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wheat-3dgs_amd"))
from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
from w3d_amd.synth import make_scene
sc = make_scene(2_000_000, seed=0)
m = GaussianModel(3)
m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
m.training_setup(OptimizationParams())
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = m.spatial_permutation()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    m._compact(p, n_keep=p.numel(), reset_stats=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"permutation {1e3*(t1-t0):.2f} ms, compaction {1e3*(t2-t1):.2f} ms")
    m._p["xyz"].data.add_(0.01 * torch.randn_like(m._p["xyz"]))
