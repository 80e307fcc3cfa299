"""Summarise gpurun_out/pmc_TAG_{a,b}: SQ counters per render-path kernel + derived ratios.
usage: pmc_report.py TAG [out.json]"""
import collections, csv, glob, json, re, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}; calls = collections.Counter()
for part in "abcd":
    for f in glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (tag, part)):
        for r in csv.DictReader(open(f)):
            m = re.search(r"rtk::(\w+)<false[,>]|rtk::(wf_fold_kernel)", r["Kernel_Name"])
            if not m: continue
            k = m.group(1) or m.group(2)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
            meta[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
b = json.load(open("gpurun_out/pmc_%s_a.json" % tag))
rays = b["rays_per_step"]
out = {"workload": b["config"]["workload"], "rays": rays, "kernels": {}}
for k, a in agg.items():
    d = {"dispatches": calls[k], "dispatch": meta[k], "counters": dict(a),
         "valu_lane_slots_per_ray": a["SQ_INSTS_VALU"] * 64 / rays,
         "valu_pipe_note": "valu_pipe_busy_pct prices a wave64 VALU op at the 2.2 cycles a SIMD issues one in with several waves resident "
                           "(tools/valu_issue_bench.hip on this box: 2.17 at 8 waves, 2.44 at 4, 4.5 for a wave that is alone; the guide: 2); "
                           "valu_pipe_busy_pct_at_4_cycles is round 2's figure (one wave's own issue rate)",
         "lane_utilisation_pct": 100 * a["SQ_THREAD_CYCLES_VALU"] / (a["SQ_ACTIVE_INST_VALU"] * 64) if a["SQ_ACTIVE_INST_VALU"] else None,
         # SQ_BUSY_CYCLES is summed over 32 shader engines; 1024 SIMDs issue one wave64 VALU op per 4 cycles
         "valu_pipe_busy_pct": 100 * a["SQ_INSTS_VALU"] * 2.2 / (a["SQ_BUSY_CYCLES"] / 32 * 1024) if a["SQ_BUSY_CYCLES"] else None,
         "valu_pipe_busy_pct_at_4_cycles": 100 * a["SQ_INSTS_VALU"] * 4 / (a["SQ_BUSY_CYCLES"] / 32 * 1024) if a["SQ_BUSY_CYCLES"] else None,
         # what a wave's cycles go to (quad-cycles; the three are disjoint and add up to SQ_WAVE_CYCLES): parked at s_waitcnt
         # (memory, LDS), stalled at issue, issuing
         "wave_cycles_waitcnt_pct": 100 * a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] if a["SQ_WAVE_CYCLES"] and "SQ_WAIT_ANY" in a else None,
         "wave_cycles_issuing_pct": 100 * a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"] if a["SQ_WAVE_CYCLES"] and "SQ_ACTIVE_INST_ANY" in a else None,
         "waves_per_simd_avg": a["SQ_WAVE_CYCLES"] * 4 / (a["SQ_BUSY_CYCLES"] / 32 * 1024) if a["SQ_BUSY_CYCLES"] else None,
         "wait_inst_share_pct": 100 * a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"] if a["SQ_WAVE_CYCLES"] else None,
         "busy_ms": a["SQ_BUSY_CYCLES"] / 32 / 2.4e6,
         # KB as rocprofv3 reports them; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (guide), other widths uncalibrated
         "fetch_size_kb": a.get("FETCH_SIZE"), "write_size_kb": a.get("WRITE_SIZE")}
    out["kernels"][k] = d
if len(sys.argv) > 2: json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, {x: (round(v, 2) if isinstance(v, float) else v) for x, v in d.items() if x not in ("counters", "dispatch")}, "VGPR", d["dispatch"]["VGPR_Count"], "LDS", d["dispatch"]["LDS_Block_Size"])
