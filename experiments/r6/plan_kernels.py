"""Kernel time per plan preparation from a rocprofv3 --kernel-trace --stats run of plan_trace.py (23 preparations).  python plan_kernels.py <kernel_stats.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    per = float(r["TotalDurationNs"]) / 23 / 1e3
    tot += per
    name = r["Name"].split("(")[0][:60]
    print("  %-60s calls/plan %5.1f  us/plan %7.1f" % (name, int(r["Calls"]) / 23, per))
print("  kernel time per plan: %.1f us" % tot)
