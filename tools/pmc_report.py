"""Summarise gpurun_out/pmc_TAG_{a,b}: counters of rtk::render_kernel<false> + derived ratios."""
import collections, csv, glob, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(float); meta = {}
for part in "ab":
    for f in glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (tag, part)):
        for r in csv.DictReader(open(f)):
            if "render_kernel<false>" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
b = json.load(open("gpurun_out/pmc_%s_a.json" % tag))
rays = b["rays_per_step"]
d = {"workload": b["config"]["workload"], "dispatch": meta, "counters": dict(agg), "rays": rays,
     "valu_wave_insts_per_ray_x64": agg["SQ_INSTS_VALU"] * 64 / rays,
     "lane_utilisation_pct": 100 * agg["SQ_THREAD_CYCLES_VALU"] / (agg["SQ_ACTIVE_INST_VALU"] * 64) if agg["SQ_ACTIVE_INST_VALU"] else None,
     "valu_share_of_wave_cycles_pct": 100 * agg["SQ_ACTIVE_INST_VALU"] / agg["SQ_WAVE_CYCLES"] if agg["SQ_WAVE_CYCLES"] else None,
     "wait_inst_share_pct": 100 * agg["SQ_WAIT_INST_ANY"] / agg["SQ_WAVE_CYCLES"] if agg["SQ_WAVE_CYCLES"] else None,
     "kernel_ms_profiled": b["roofline"]["kernel_ms"]}
json.dump(d, open(sys.argv[2], "w"), indent=1) if len(sys.argv) > 2 else None
print(json.dumps(d, indent=1))
