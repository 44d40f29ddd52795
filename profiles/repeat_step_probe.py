"""Run-to-run repeatability of the whole training step in every exchange form (round 4): `reps` fresh (model, Trainer) pairs per
form — single GPU with and without the fused Adam, low-rank, low-rank with the early gather, sparse rows, dense — on a one-rank
RCCL group, three steps each; first / second moments, statistics and the first image are compared with the first pair's
(bit for bit in the deterministic mode and for the integer statistics, 2e-3 / 1e-4 of the block maximum otherwise).
  usage: python3 profiles/repeat_step_probe.py [reps] [P]          (measured: 0 anomalies in 12 forms x 20 runs at P = 6 999 and x 12 at P = 300)"""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 6999
from w3d_amd.synth import make_scene, make_cameras
from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
from w3d_amd.train import Trainer
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
W, H = 208, 160
cams = [c.to(dev) for c in make_cameras(4, W, H)]
g = torch.Generator().manual_seed(5)
for cam in cams:
    cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
bg = torch.tensor([0.1, 0.1, 0.0], device=dev)
sc = make_scene(P, seed=13, scale_mean=0.02)
total_bad = 0
for det in (False, True):
    for name in ("single_adam", "single", "lowrank", "lowrank_early", "rows", "dense"):
        ref = None
        bad = 0
        for r in range(reps):
            m = GaussianModel(3, device=dev)
            m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
            m.active_sh_degree = 3
            m.deterministic = det
            opt = OptimizationParams()
            m.training_setup(opt)
            tr = Trainer(m, cams, opt, bg, densify=False, force_exchange=not name.startswith("single"))
            tr.fused_adam = name == "single_adam"
            tr.exchange_mode = "lowrank" if name.startswith("lowrank") else name if name in ("rows", "dense") else tr.exchange_mode
            tr.early_gather = name == "lowrank_early"
            imgs = []
            for it in range(1, 4):
                tr.step(it)
                imgs.append(tr.last["image"].clone())
            if not name.startswith("single"):
                tr.gather_moments()
            cur = dict(m1=m.optimizer.exp_avg.clone(), v1=m.optimizer.exp_avg_sq.clone(), acc=m.xyz_gradient_accum.flatten().clone(),
                       den=m.denom.flatten().clone(), rad=m.max_radii2D.flatten().clone().float(), img0=imgs[0].flatten())
            if ref is None:
                ref = cur; continue
            for k in cur:
                a, b = cur[k], ref[k]
                if det or k in ("den", "rad", "img0"):
                    ok = torch.equal(a, b)
                    rel = float((a - b).abs().max() / (b.abs().max() + 1e-30))
                else:
                    rel = float((a - b).abs().max() / (b.abs().max() + 1e-30))
                    ok = rel <= (2e-3 if k in ("m1", "v1") else 1e-4)      # 3 Adam steps: sign flips of ~0 gradients move params
                if not ok:
                    bad += 1
                    idx = ((a - b).abs() > 1e-4 * b.abs().max()).nonzero().flatten()[:6].tolist()
                    print(f"det={det} {name} rep {r}: {k} differs rel {rel:.3e} at {idx}")
        print(f"det={det} {name}: {reps} runs, anomalies {bad}")
        total_bad += bad
print("TOTAL", total_bad)
dist.destroy_process_group()
