"""Per-wave counter values of the step kernels in a rocprofv3 --pmc output directory:  python tools/pmc_summary.py <dir> [nwaves]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
nw = float(sys.argv[2]) if len(sys.argv) > 2 else 1024.0
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    agg.setdefault((r['Kernel_Name'][:44], r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
last = {}
for (kn, d), v in agg.items():
    if 'k_step' in kn or 'k_adjoint' in kn:
        last[kn] = v
for kn, v in last.items():
    print("  ", kn, " ".join("%s=%.3g" % (c.replace('SQ_', ''), v.get(c, 0) / nw) for c in sorted(v)))
