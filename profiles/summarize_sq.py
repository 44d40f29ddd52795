"""Per-kernel SQ counters from rocprofv3 --pmc passes -> csv on stdout.

  summarize_sq.py --window <marker> --steps K pass1.csv pass2.csv ...    (round 4; what bench.py reads)
      per kernel: the SUM over every dispatch between the first two launches of the marker kernel (profiles/scene_step.py),
      divided by K: counts PER STEP with all launches of the kernel added up.
  summarize_sq.py pass1.csv pass2.csv ...                                 (rounds 1-3)
      per kernel: the mean over the second half of its dispatches (the benchmark's full-size launches).

simd_cycles_per_valu_instr = (SQ_BUSY_CYCLES / 32 x 1024 SIMDs) / SQ_INSTS_VALU: SIMD cycles of the kernel's duration per wave64
VALU instruction it issued (SQ_BUSY_CYCLES is summed over the 32 shader engines).  What a SIMD can sustain per instruction
class is measured by profiles/valu_microbench.hip (r02: v_fma/v_mul/v_cndmask_e32 2.3-2.6 cycles with >= 2 waves per SIMD,
DPP / v_cmp / v_max / v_cndmask_e64 4.2-4.5, v_exp / v_rcp 8.2)."""
import collections
import csv
import re
import sys

argv = sys.argv[1:]
marker, steps = None, 1
if argv and argv[0] == "--window":
    marker, steps, argv = argv[1], int(argv[3]), argv[4:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in argv:
    rows = list(csv.DictReader(open(path)))
    if marker:
        ids = sorted({int(r["Dispatch_Id"]) for r in rows if marker in r["Kernel_Name"]})
        if len(ids) < 2:
            sys.exit(f"{path}: fewer than two launches of a kernel named *{marker}*")
        rows = [r for r in rows if ids[0] < int(r["Dispatch_Id"]) < ids[1]]
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        if "at::native" in name or "rocclr" in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
        "SQ_WAIT_INST_ANY"]
print("kernel," + ("launches_per_step" if marker else "dispatches") + "," + ",".join(cols) + ",simd_cycles_per_valu_instr")
out = []
for k, v in acc.items():
    vals = {}
    for c in cols:
        x = v.get(c, [])
        if marker:
            vals[c] = sum(x) / steps
        else:
            x = x[len(x) // 2:]
            vals[c] = sum(x) / len(x) if x else 0.0
    busy = vals["SQ_BUSY_CYCLES"] / 32.0
    frac = busy * 1024.0 / vals["SQ_INSTS_VALU"] if vals["SQ_INSTS_VALU"] > 0 else 0.0
    n = len(v.get("SQ_WAVES", []))
    out.append((vals["SQ_INSTS_VALU"], k, n / steps if marker else n, vals, frac))
for _, k, n, vals, frac in sorted(out, reverse=True):
    print(f"\"{k}\",{n:g}," + ",".join(f"{vals[c]:.0f}" for c in cols) + f",{frac:.2f}")
