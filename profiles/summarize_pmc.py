"""FETCH_SIZE / WRITE_SIZE per kernel from two rocprofv3 --pmc passes -> pmc_hbm_traffic.csv (stdout).
Per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in KiB and, on gfx950,
FETCH_SIZE under-reports reads by 2x (calibrated here on the Adam sweep, which reads 4 x 4 B x n by construction).
The per-kernel figure is the MAX over its dispatches: the benchmark's own full-size launches, not the smaller
ground-truth renders that share kernels with them."""
import collections
import csv
import re
import sys


def load(path, counter):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        out[name].append(float(r["Counter_Value"]))
    return out


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fetch:
    f, w = max(fetch[k]), max(write.get(k, [0.0]))
    rows.append((k, len(fetch[k]), f, w, 2.0 * f / 1024.0, w / 1024.0))
rows.sort(key=lambda r: -(r[4] + r[5]))
print("kernel,dispatches,FETCH_SIZE_KiB_max,WRITE_SIZE_KiB_max,hbm_read_MiB_corrected_x2,hbm_write_MiB")
for k, n, f, w, rm, wm in rows:
    if "at::native" in k or rm + wm < 1.0:
        continue
    print(f"\"{k}\",{n},{f:.1f},{w:.1f},{rm:.1f},{wm:.1f}")
