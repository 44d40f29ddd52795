"""Per-kernel SQ counters (mean over the second half of a kernel's dispatches = the benchmark's full-size launches) from
rocprofv3 --pmc passes -> sq_counters.csv (stdout).  simd_cycles_per_valu_instr = (SQ_BUSY_CYCLES / 32 x 1024 SIMDs) / SQ_INSTS_VALU:
SIMD cycles of the kernel's duration per wave64 VALU instruction it issued (SQ_BUSY_CYCLES is summed over the 32 shader
engines).  What a SIMD can sustain per instruction class is measured by profiles/valu_microbench.hip (r02: v_fma/v_mul/
v_cndmask_e32 2.3-2.6 cycles with >= 2 waves per SIMD, DPP / v_cmp / v_max / v_cndmask_e64 4.2-4.5, v_exp / v_rcp 8.2)."""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        if "at::native" in name or "rocclr" in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
        "SQ_WAIT_INST_ANY"]
print("kernel,dispatches," + ",".join(cols) + ",simd_cycles_per_valu_instr")
rows = []
for k, v in acc.items():
    vals = {}
    for c in cols:
        x = v.get(c, [])
        x = x[len(x) // 2:]
        vals[c] = sum(x) / len(x) if x else 0.0
    busy = vals["SQ_BUSY_CYCLES"] / 32.0
    frac = busy * 1024.0 / vals["SQ_INSTS_VALU"] if vals["SQ_INSTS_VALU"] > 0 else 0.0
    rows.append((vals["SQ_INSTS_VALU"], k, len(v.get("SQ_WAVES", [])), vals, frac))
for _, k, n, vals, frac in sorted(rows, reverse=True):
    print(f"\"{k}\",{n}," + ",".join(f"{vals[c]:.0f}" for c in cols) + f",{frac:.2f}")
