"""Per kernel of a rocprofv3 kernel trace: launches, median / 90th / 99th percentile / max duration in us — a launch far above
the rest is a wave that ran a serial loop (profiles/r3/ab_coop_folds.txt).  usage: python3 tools/kernel_outliers.py kernel_trace.csv"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wf_" in r["Kernel_Name"] and "<true" not in r["Kernel_Name"]]
by = {}
for r in rows:
    n = r["Kernel_Name"]
    k = "traverse" if "traverse" in n else "shade" if "shade" in n else "generate" if "generate" in n else "tail" if "tail" in n else "fold"
    by.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    v = np.array(v)
    print("%-9s launches %6d  median %8.1f  p90 %8.1f  p99 %8.1f  max %8.1f us" % (k, len(v), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
