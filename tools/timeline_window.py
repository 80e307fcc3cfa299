"""Kernel timeline per hardware queue over the last N frames of a kernel trace (frames overlap when two are in flight).
usage: python3 tools/timeline_window.py kernel_trace.csv [frames=3]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wf_" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
folds = [i for i, r in enumerate(rows) if "fold" in r["Kernel_Name"]]
t_lo = int(rows[folds[-n - 1]]["End_Timestamp"])
t_hi = int(rows[folds[-1]]["End_Timestamp"])
seg = [r for r in rows if int(r["End_Timestamp"]) > t_lo and int(r["Start_Timestamp"]) <= t_hi]
print("window us %.1f, launches %d, folds end at us %s" % ((t_hi - t_lo) / 1e3, len(seg), [round((int(rows[i]["End_Timestamp"]) - t_lo) / 1e3) for i in folds[-n:]]))
for q in sorted(set(r["Queue_Id"] for r in seg)):
    out = []
    for r in seg:
        if r["Queue_Id"] != q:
            continue
        nme = r["Kernel_Name"]
        k = "T" if "traverse" in nme else "S" if "shade" in nme else "G" if "generate" in nme else "X" if "tail" in nme else "F"
        out.append("%s%d@%d" % (k, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - t_lo) / 1e3))
    print("queue", q, " ".join(out))
