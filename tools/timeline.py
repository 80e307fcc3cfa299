import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if "wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
folds=[i for i,r in enumerate(rows) if "fold" in r["Kernel_Name"]]
a,b=folds[-2]+1,folds[-1]+1
seg=rows[a:b]
t0=int(seg[0]["Start_Timestamp"])
print("step span us", (int(seg[-1]["End_Timestamp"])-t0)/1e3, "launches", len(seg))
for q in sorted(set(r["Queue_Id"] for r in seg)):
    s=[r for r in seg if r["Queue_Id"]==q]
    out=[]
    for r in s:
        n=r["Kernel_Name"]; k="T" if "traverse" in n else "S" if "shade" in n else "G" if "generate" in n else "X" if "tail" in n else "F"
        out.append("%s%d@%d"%(k,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,(int(r["Start_Timestamp"])-t0)/1e3))
    print("queue",q," ".join(out[:90]))
    x=[r for r in s if "tail" in r["Kernel_Name"]]
    if x: print("   tail regs", x[0].get("VGPR_Count"), x[0].get("Accum_VGPR_Count"), x[0].get("Scratch_Size"), x[0].get("LDS_Block_Size"))
