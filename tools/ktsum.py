import csv,sys,re
for t in sys.argv[1:]:
    print("==", t)
    for r in csv.DictReader(open("gpurun_out/kt_%s_kernel_stats.csv"%t)):
        n=r["Name"]; n=n[n.find("rtk::"):][:30] if "rtk::" in n else n[:30]
        print("%-32s calls %5s total %9.2f ms avg %8.1f us min %8.1f max %8.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
    m=re.search(r'"ms_per_step": ([0-9.]+)', open("gpurun_out/kt_%s.json"%t).read()); print("ms_per_step", m and m.group(1))
