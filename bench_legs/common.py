"""Shared pieces of the measurement: scene construction, workload statistics, per-stage HIP-event timing (StepMeter) and
the roofline arithmetic.  bench.py (the timed step + the compact line) and every leg under
bench_legs/ use these."""
import ctypes
import json
import math
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
# fp32 vector peak of the chip: 256 CUs x 4 SIMD-32 x 2.4 GHz x 2 flop (MI355X_MICROARCH.md 'Peak FP32 (vector)')
# = one wave64 VALU instruction per 2 cycles per SIMD.  The VALU roofline of the blend kernels prices every issued
# wave64 VALU instruction as 128 flop-equivalents against it (the measured sustainable rate is in
# profiles/r02/valu_microbench.json and replaces the spec figure when present).
VALU_SPEC_TFLOPS = 157.3
FLOP_PER_VALU_INSTR = 128.0


# ------------------------------------------------------------------------------------------------ scene
def build_scene(args, dev):
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    sc = make_scene(args.points, seed=0)
    model = GaussianModel(3, device=dev)
    model.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    model.active_sh_degree = 3
    opt = OptimizationParams()
    model.training_setup(opt)
    cams = [c.to(dev) for c in make_cameras(args.views, args.width, args.height)]
    return sc, model, opt, cams


def make_ground_truth(args, cams, dev, bg):
    """GT image of each view = render of a DIFFERENT seed's scene + noise, so the loss gradient is dense."""
    from w3d_amd.synth import make_scene
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.train import render_views
    sc = make_scene(max(args.points // 4, 1000), seed=1, scale_mean=0.009)
    gt_model = GaussianModel(3, device=dev)
    gt_model.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    gt_model.active_sh_degree = 3
    g = torch.Generator(device="cpu").manual_seed(7)
    for cam, img in zip(cams, render_views(gt_model, cams, bg)):
        noise = 0.03 * torch.randn(img.shape, generator=g).to(dev)
        cam.original_image = (img + noise).clamp(0.0, 1.0).contiguous()
    del gt_model
    torch.cuda.empty_cache()


def workload_stats(model, cam, bg, dev):
    """Measured V, R and R_walk (entries the reverse walk must visit) of one view."""
    from w3d_amd.rasterizer import _forward_impl, debug_pixel_state
    from w3d_amd.gaussian_renderer import _settings
    from w3d_amd.rasterizer import GaussianRasterizationSettings
    with torch.no_grad():
        # (one list per tile, whatever list_share the trainer currently runs: R and the walk lengths are then the culled
        #  per-tile figures, comparable between scenes and rounds)
        s = _settings(GaussianRasterizationSettings, cam, model, bg, 1.0, False)._replace(list_share=0)
        _, radii, _, _, saved, _ = _forward_impl(s, model.get_xyz, model.get_features, None, model.get_opacity,
                                                 model.get_scaling, model.get_rotation, None)
        _, nc = debug_pixel_state(saved)
        H, W = nc.shape
        gy, gx = (H + 15) // 16, (W + 15) // 16
        pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
        pad[:H, :W] = nc.to(torch.int64)
        r_walk = int(pad.view(gy, 16, gx, 16).amax(dim=(1, 3)).sum())
        # contributors = entries a pixel actually BLENDS (alpha >= 1/255, before it saturates): the FlashSplat forward counts
        # them (contrib_num); n_contrib above is the list POSITION of the last one — every entry of the tile's list in front of
        # it counts there, whether it touches the pixel or not
        from w3d_amd.rasterizer import FlashSplatRasterizationSettings
        sf = FlashSplatRasterizationSettings(*s[:12], mask_grad=False, num_obj=1, tile_cull=True, deterministic=False, list_share=0)
        ex = _forward_impl(sf, model.get_xyz, model.get_features, None, model.get_opacity, model.get_scaling, model.get_rotation,
                           None, flash=dict(gt_mask=None, num_obj=1))[5]
        return dict(V=saved["num_visible"], R=saved["num_rendered"], R_walk=r_walk,
                    mean_last=float(nc.float().mean()), mean_contrib=float(ex[0].float().mean()))


def _newest(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", pattern)))
    return files[-1] if files else None


# kernels of every stage of the step (names as rocprofv3 prints them, template arguments included where they matter)
STAGE_KERNELS = {
    "preprocess_fwd": ("preprocess_fwd_kernel",),
    "depth_sort": ("depth_bucket_", "depth_grid_"),
    "tile_count_scan": ("chunk_walk_kernel<0", "seg_sum_kernel", "tile_scan_kernel", "chunk_off_kernel", "tile_count_", "band_"),
    "fill_lists": ("chunk_walk_kernel<1", "fill_"),
    "render_fwd": ("render_fwd_kernel",),
    "loss": ("ssim_pass_a", "ssim_pass_b"),
    "render_bwd": ("render_bwd_kernel", "zero_visible_records_kernel", "det_gather_kernel"),
    # (the blend kernels' block -> (tile, part) schedule: launched in front of both, outside their event brackets)
    "tile_schedule": ("tile_schedule_kernel", "tile_order_kernel"),
    "preprocess_bwd": ("preprocess_bwd_kernel",),
}


def _scene_csv(prefix, scene):
    """newest profiles/rNN/<prefix>_<scene>.csv (per-step sums inside the marker window of profiles/scene_step.py)"""
    return _newest(f"{prefix}_{scene}.csv")


def pmc_traffic(stage, scene="untrained"):
    """HBM-side bytes ONE STEP of `scene` moves in `stage`: the SUM over every kernel of the stage (STAGE_KERNELS) and over
    all its launches in a step, from the newest committed profiles/rNN/pmc_hbm_traffic_<scene>.csv (FETCH_SIZE x2 +
    WRITE_SIZE collected in separate rocprofv3 --pmc passes over profiles/scene_step.py, whose K steps sit between two
    marker kernels; profiles/summarize_pmc.py --window).  (bytes, [kernel rows]) or (None, None) when no summary exists —
    the counters cannot be read live from inside the process."""
    import csv
    f = _scene_csv("pmc_hbm_traffic", scene)
    if not f:
        return None, None
    tot, used = 0.0, []
    for r in csv.DictReader(open(f)):
        if any(r["kernel"].startswith(k) for k in STAGE_KERNELS.get(stage, (stage,))):
            mib = float(r["hbm_read_MiB_corrected_x2"]) + float(r["hbm_write_MiB"])
            tot += mib
            used.append({"kernel": r["kernel"], "launches_per_step": float(r["launches_per_step"]), "MiB_per_step": round(mib, 2)})
    return (int(tot * 1024 * 1024), used) if used else (None, None)


def valu_instructions(kernel, scene="untrained"):
    """wave64 VALU instructions `kernel` issues per step of `scene` (SQ_INSTS_VALU summed over its launches inside the marker
    window, newest committed profiles/rNN/sq_counters_<scene>.csv); (count, file) or (None, None)."""
    import csv
    f = _scene_csv("sq_counters", scene)
    if not f:
        return None, None
    for r in csv.DictReader(open(f)):
        if r["kernel"].startswith(kernel) and "<true>" not in r["kernel"]:
            return float(r["SQ_INSTS_VALU"]), os.path.relpath(f, ROOT)
    return None, None


def valu_peak():
    """Sustainable wave64 VALU issue rate of the chip measured by profiles/valu_microbench.hip (v_fma_f32, best over the
    waves-per-SIMD settings), as TFLOP/s-equivalents (x128); falls back to the spec fp32 vector peak."""
    f = _newest("valu_microbench.json")
    if f:
        try:
            res = json.load(open(f))["results"]
            rate = max(r["wave_instr_per_s"] for r in res if r["op"] == "v_fma_f32")
            return rate * FLOP_PER_VALU_INSTR / 1e12, os.path.relpath(f, ROOT)
        except Exception:
            pass
    return VALU_SPEC_TFLOPS, "spec (MI355X_MICROARCH.md, Peak FP32 vector)"



def _progress(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


# ------------------------------------------------------------------------------------------------ roofline helpers
SPATIAL_ORDER = True        # (main() clears it under --no-spatial-order)


def kernel_bytes(P, V, R, Rw, HW, fused_adam):
    """Algorithmic HBM bytes per launch of every stage (what the stage must read and write once)."""
    return {
        # parameters read; packed per-visible records written.  In Morton order the culled Gaussians come in runs and their
        # 180-B SH rows are not requested at all: 56 B of geometry for everyone, the SH rows of the visible
        "preprocess_fwd": (56.0 * P + 180.0 * V + 64.0 * V) if SPATIAL_ORDER else (236.0 * P + 64.0 * V),
        # P keys read twice (histogram, split), V (key, id) pairs written and read once, 24-B records written from a 16-B rect/mask gather
        "depth_sort": 2 * 4.0 * P + 2 * 8.0 * V + (16.0 + 24.0) * V,
        "tile_count_scan": 24.0 * V,                           # the records, once
        "fill_lists": 24.0 * V + 4.0 * R,                      # the records once + the lists
        "render_fwd": 48.0 * Rw + 36.0 * HW,                   # 4-B id + 44-B gather per walked instance; image + aux
        "loss": 2 * 12.0 * HW + 12.0 * HW,                     # image + gt read, gradient written
        "render_bwd": 84.0 * Rw + 20.0 * HW,                   # gather + one 40-B record update; dL/dpixel + aux
        # fused Adam: parameters + both moments read and written, 2-D records read / otherwise gradients written
        "preprocess_bwd": (6 * 236.0 * P + 104.0 * V) if fused_adam else (236.0 * P + 64.0 * V + 252.0 * P),
    }


def kernel_table(stage_ms, kb, scene):
    """Per stage: event-timed ms, algorithmic bytes, achieved GB/s and fraction of the HBM peak, and the PMC-counter traffic of
    the stage on THIS scene — summed over all its kernels and launches (pmc_traffic) — with its ratio to the algorithmic bytes
    (wasted re-reads show up there)."""
    rows = []
    for k, ms in sorted(stage_ms.items(), key=lambda kv: -kv[1]):
        if k not in kb:
            continue
        gbs = kb[k] / (ms * 1e-3) / 1e9
        tr, used = pmc_traffic(k, scene)
        rows.append({"stage": k, "ms": ms, "algorithmic_bytes": int(kb[k]), "achieved_GBps": round(gbs, 1),
                     "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "pmc_traffic_bytes": tr,
                     "traffic_ratio": None if not tr else round(tr / kb[k], 2),
                     "pmc_GBps": None if not tr else round(tr / (ms * 1e-3) / 1e9, 1), "pmc_kernels": used})
    return rows


def step_roofline(P, V, R, HW, it_per_s, stage_ms, kb):
    """SURVEY.md section 8(d) / BASELINE.md section 5 as written: (B_f + B_b + B_adam) * iters/s / peak with the MEASURED
    V and R, and the same with this design's own byte count (sum of the stages' algorithmic bytes — fusion removed the
    gradient round trip and the 64-bit key sort, so it is smaller)."""
    B_f = 236.0 * P + 56.0 * V + 80.0 * R + 28.0 * HW
    B_b = 484.0 * P + 80.0 * V + 84.0 * R + 20.0 * HW
    B_adam = 1652.0 * P
    tot = B_f + B_b + B_adam
    own = sum(kb[k] for k in kb if k in stage_ms)
    return {"formula": "B_f + B_b + B_adam, B_f = 236P + 56V + 80R + 28HW, B_b = 484P + 80V + 84R + 20HW, B_adam = 1652P",
            "P": P, "V": int(V), "R": int(R), "HW": HW, "B_f": int(B_f), "B_b": int(B_b), "B_adam": int(B_adam),
            "algorithmic_bytes": int(tot), "achieved_GBps": round(tot * it_per_s / 1e9, 1),
            "frac": round(tot * it_per_s / 1e9 / HBM_PEAK_GBS, 4),
            "design_bytes": int(own), "design_achieved_GBps": round(own * it_per_s / 1e9, 1),
            "design_frac": round(own * it_per_s / 1e9 / HBM_PEAK_GBS, 4)}


VALU_BOUND_STAGES = {"render_bwd": "render_bwd_kernel", "render_fwd": "render_fwd_kernel"}


def valu_object(stage, ms, scene):
    """`stage` (a blend kernel) against the VALU issue roof: wave64 VALU instructions per launch from
    the committed SQ counter summary of THIS scene x 128 flop-equivalents / the measured launch time."""
    n_valu, src = valu_instructions(VALU_BOUND_STAGES[stage], scene)
    if not n_valu:
        return None
    peak, peak_src = valu_peak()
    ach = n_valu * FLOP_PER_VALU_INSTR / (ms * 1e-3) / 1e12
    return {"bound": "valu", "kernel": stage, "achieved": round(ach, 2), "peak": VALU_SPEC_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / VALU_SPEC_TFLOPS, 4), "peak_measured": round(peak, 1), "frac_of_measured": round(ach / peak, 4),
            "avg_launch_ms": round(ms, 4), "valu_wave_instr_per_launch": int(n_valu), "valu_instr_source": src,
            "peak_source": peak_src, "flop_equiv_per_wave_instr": FLOP_PER_VALU_INSTR}


def roofline_object(meas, P, ws, HW, fused_adam, it_per_s, scene):
    """The `roofline` object of one scene: its DOMINANT stage (the longest one, found by the probe; timed with HIP events on
    its launch stream inside the timed region) against the roof that bounds it — HBM for the per-Gaussian and binning
    stages, VALU issue for the blend kernels (their HBM view is kept beside it) — plus the whole-step formula of SURVEY
    section 8(d) and the per-stage table."""
    V, R, Rw = ws["V"], ws["R"], ws["R_walk"]
    kb = kernel_bytes(P, V, R, Rw, HW, fused_adam)
    dom = meas["dominant"]
    if dom not in kb or not meas.get("live") or meas["live"][0] <= 0:
        return None
    cnt, ms = meas["live"]
    avg_ms = ms / cnt
    hbm_ach = kb[dom] / (avg_ms * 1e-3) / 1e9
    tr, used = pmc_traffic(dom, scene)
    label = "preprocess_bwd+adam" if (dom == "preprocess_bwd" and fused_adam) else dom
    roof = {"bound": "hbm", "kernel": label, "achieved": round(hbm_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(hbm_ach / HBM_PEAK_GBS, 4), "traffic": tr, "traffic_ratio": None if not tr else round(tr / kb[dom], 2),
            "avg_launch_ms": round(avg_ms, 4), "launches": cnt, "algorithmic_bytes_per_launch": int(kb[dom]),
            "scene": scene, "chosen": meas["chosen_by"], "probe_stage_ms": meas["probe_ms"]}
    if dom in VALU_BOUND_STAGES:
        vo = valu_object(dom, avg_ms, scene)
        if vo is not None:
            hbm_view = {k: roof[k] for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_ratio",
                                             "algorithmic_bytes_per_launch")}
            roof.update(vo)
            roof["kernel"] = label
            roof["hbm_view"] = hbm_view
        else:
            roof["note"] = ("a blend kernel: VALU-issue-bound (DESIGN.md section 2.1); no SQ counter summary of this scene is "
                            "committed, so only its HBM view is given")
    roof["step"] = step_roofline(P, V, R, HW, it_per_s, meas["stage_ms"], kb)
    roof["kernels"] = kernel_table(meas["stage_ms"], kb, scene)
    return roof


class StepMeter:
    """Trainer steps between barrier + synchronize brackets (max over ranks), with the library's per-stage event timing."""

    def __init__(self, trainer, world, dev):
        from w3d_amd import _lib
        self.trainer, self.world, self.dev, self.lib = trainer, world, dev, _lib.lib
        self.lib.w3d_profile_enable.argtypes = [ctypes.c_char_p]
        self.lib.w3d_profile_collect.argtypes = [ctypes.c_char_p, ctypes.c_uint64]

    def sync(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, n_steps, it, prof_sel):
        """n_steps trainer steps; returns (seconds, it, {stage: (launches, total ms)})."""
        self.sync()
        self.lib.w3d_profile_enable(prof_sel)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            it += 1
            self.trainer.step(it)
        self.sync()
        t1 = time.perf_counter()
        self.lib.w3d_profile_enable(None)
        buf = ctypes.create_string_buffer(1 << 16)
        self.lib.w3d_profile_collect(buf, len(buf))
        stages = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            stages[name] = (int(cnt), float(ms))
        el = torch.tensor([t1 - t0], device=self.dev, dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el), it, stages

    def measure(self, n_steps, it, profile="auto", all_stages=False, probe_steps=10, stage_steps=20):
        """The measurement protocol of one scene: (1) an untimed probe with every stage timed finds the dominant stage;
        (2) n_steps timed steps with ONLY that stage's events inside the timed region (an event pair around every stage costs
        ~4 % of the step); (3) stage_steps more steps with every stage timed, for the per-stage table."""
        _, it, pr = self.timed(probe_steps, it, b"*")
        probe_ms = {k: round(ms / c, 4) for k, (c, ms) in pr.items() if c > 0}
        known = kernel_bytes(1, 1, 1, 1, 1, True)
        if profile == "auto":
            cand = {k: v for k, v in probe_ms.items() if k in known}
            dominant = max(cand, key=cand.get) if cand else "preprocess_bwd"
            chosen_by = f"longest stage of a {probe_steps}-step probe with every stage timed"
        else:
            dominant, chosen_by = profile, "--profile"
        el, it, st = self.timed(n_steps, it, b"*" if all_stages else dominant.encode())
        if all_stages:
            stage_ms = {k: round(ms / c, 4) for k, (c, ms) in st.items() if c > 0}
        else:
            _, it, st2 = self.timed(stage_steps, it, b"*")
            stage_ms = {k: round(ms / c, 4) for k, (c, ms) in st2.items() if c > 0}
        return {"elapsed": el, "it": it, "dominant": dominant, "chosen_by": chosen_by, "probe_ms": probe_ms,
                "live": st.get(dominant), "stages": st, "stage_ms": stage_ms}



def mean_workload(model, cams, bg, dev):
    ws = [workload_stats(model, cams[i], bg, dev) for i in (0, len(cams) // 2)]
    return {k: sum(w[k] for w in ws) / len(ws) for k in ("V", "R", "R_walk", "mean_contrib", "mean_last")}


# ------------------------------------------------------------------------------------------------ densified scene
def _psnr_db(a, b):
    """reference utils/image_utils.py:17-19 on one image"""
    mse = float(((a - b) ** 2).mean())
    return 99.0 if mse == 0 else 20.0 * math.log10(1.0 / math.sqrt(mse))

