"""Per-kernel SQ counters (mean over the second half of a kernel's dispatches = the benchmark's full-size launches) from
rocprofv3 --pmc passes -> sq_counters.csv (stdout).  valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (SQ_BUSY_CYCLES/32 x 1024 SIMDs):
the share of the kernel's duration a SIMD spends issuing VALU instructions if they were spread evenly (wave64 on SIMD16:
4 cycles per instruction; SQ_BUSY_CYCLES is summed over the 32 shader engines)."""
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
print("kernel,dispatches," + ",".join(cols) + ",valu_issue_frac")
rows = []
for k, v in acc.items():
    vals = {}
    for c in cols:
        x = v.get(c, [])
        x = x[len(x) // 2:]
        vals[c] = sum(x) / len(x) if x else 0.0
    busy = vals["SQ_BUSY_CYCLES"] / 32.0
    frac = vals["SQ_INSTS_VALU"] * 4.0 / (busy * 1024.0) if busy > 0 else 0.0
    rows.append((vals["SQ_INSTS_VALU"], k, len(v.get("SQ_WAVES", [])), vals, frac))
for _, k, n, vals, frac in sorted(rows, reverse=True):
    print(f"\"{k}\",{n}," + ",".join(f"{vals[c]:.0f}" for c in cols) + f",{frac:.3f}")
