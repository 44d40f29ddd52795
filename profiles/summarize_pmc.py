"""FETCH_SIZE / WRITE_SIZE per kernel from two rocprofv3 --pmc passes -> csv on stdout.

Per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in KiB and, on gfx950, FETCH_SIZE
under-reports reads by 2x (calibrated here on the Adam sweep, which reads 4 x 4 B x n by construction); WRITE_SIZE is exact.
Both count the L2's memory-side requests (Infinity-Cache hits included).

  summarize_pmc.py --window <marker> --steps K fetch.csv write.csv      (round 4; what bench.py reads)
      per kernel: the SUM over every dispatch between the first and the second launch of the marker kernel (a kernel whose
      name contains <marker>; profiles/scene_step.py brackets its K measured steps with torch.lgamma), divided by K:
      bytes PER STEP, all launches of the kernel added up (the four radix passes, both SSIM passes, ...).
  summarize_pmc.py fetch.csv write.csv                                   (rounds 1-3)
      per kernel: the MAX over its dispatches."""
import collections
import csv
import re
import sys


def clean(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0]


def load(path, counter, marker=None):
    """{kernel: [values]} — with a marker only the dispatches between its first two launches (by Dispatch_Id)"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    if marker:
        ids = sorted(int(r["Dispatch_Id"]) for r in rows if marker in r["Kernel_Name"])
        if len(ids) < 2:
            sys.exit(f"{path}: fewer than two launches of a kernel named *{marker}*")
        rows = [r for r in rows if ids[0] < int(r["Dispatch_Id"]) < ids[1]]
    out = collections.defaultdict(list)
    for r in rows:
        out[clean(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


def main():
    argv = sys.argv[1:]
    marker, steps = None, 1
    if argv and argv[0] == "--window":
        marker, argv = argv[1], argv[2:]
        assert argv[0] == "--steps"
        steps, argv = int(argv[1]), argv[2:]
    fetch, write = load(argv[0], "FETCH_SIZE", marker), load(argv[1], "WRITE_SIZE", marker)
    rows = []
    if marker:
        for k in sorted(set(fetch) | set(write)):
            f, w = sum(fetch.get(k, [0.0])) / steps, sum(write.get(k, [0.0])) / steps
            rows.append((k, len(fetch.get(k, write.get(k, []))) / steps, f, w, 2.0 * f / 1024.0, w / 1024.0))
        rows.sort(key=lambda r: -(r[4] + r[5]))
        print("kernel,launches_per_step,FETCH_SIZE_KiB_per_step,WRITE_SIZE_KiB_per_step,hbm_read_MiB_corrected_x2,hbm_write_MiB")
        for k, n, f, w, rm, wm in rows:
            print(f"\"{k}\",{n:g},{f:.1f},{w:.1f},{rm:.2f},{wm:.2f}")
        return
    for k in fetch:
        f, w = max(fetch[k]), max(write.get(k, [0.0]))
        rows.append((k, len(fetch[k]), f, w, 2.0 * f / 1024.0, w / 1024.0))
    rows.sort(key=lambda r: -(r[4] + r[5]))
    print("kernel,dispatches,FETCH_SIZE_KiB_max,WRITE_SIZE_KiB_max,hbm_read_MiB_corrected_x2,hbm_write_MiB")
    for k, n, f, w, rm, wm in rows:
        if "at::native" in k or rm + wm < 1.0:
            continue
        print(f"\"{k}\",{n},{f:.1f},{w:.1f},{rm:.1f},{wm:.1f}")


main()
