"""Copy the judged summaries of one GPU session from gpurun_out/<tag>/ (scratch) into profiles/ (tracked):

    python tools/collect_profiles.py <tag>

  <tag>_bench*.json                       the bench lines of the session
  <tag>_kernel_stats.csv / _kernel_trace.csv   rocprofv3 --kernel-trace --stats of the headline command (k_* rows only)
  <tag>_pmc_<workload>_<pass>.csv (+ .json)    counter_collection.csv of every PMC pass, rows of the library's kernels only, and the
                                          bench line of that very run (the evaluation / iteration counts the calibration uses)
"""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def trim(src, dst, col="Kernel_Name"):
    rows = list(csv.DictReader(open(src)))
    if not rows:
        return
    keep = [r for r in rows if "k_" in r.get(col, r.get("Name", "")) and "rocclr" not in r.get(col, r.get("Name", ""))]
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    for f in glob.glob(os.path.join(src, "bench*.json")) + glob.glob(os.path.join(src, "*.txt")):
        if os.path.getsize(f) > 0:
            shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, os.path.basename(f))))
    for kind in ("kernel_stats", "kernel_trace"):
        for f in glob.glob(os.path.join(src, "prof", "*", "*_%s.csv" % kind)):
            trim(f, os.path.join(dst, "%s_%s.csv" % (tag, kind)), "Name" if kind == "kernel_stats" else "Kernel_Name")
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        name = os.path.basename(d)
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            trim(f, os.path.join(dst, "%s_%s.csv" % (tag, name)))
        j = d + ".json"
        if os.path.exists(j) and os.path.getsize(j) > 0:
            shutil.copy(j, os.path.join(dst, "%s_%s.json" % (tag, name)))
    print("copied into profiles/:", len(glob.glob(os.path.join(dst, tag + "_*"))), "files")


main()
