"""usage: regions.py WORKLOAD SPP — where the counting variant's waves spend their cycles (rt_stats.region_cycles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtiow_loader
rt = rtiow_loader.load()
wl, spp = sys.argv[1], int(sys.argv[2])
over = {"image_width": int(sys.argv[3])} if len(sys.argv) > 3 else {}
scene, params, camera, sampler = rt.configs.build(wl, samples_per_pixel=spp, **over)
gpu = rt.backend.HipBackend(0); gpu.upload_scene(scene.flatten()); gpu.upload_sampler(sampler)
_, st = gpu.render(params, camera, stats=True)
# the timers come from the product kernels of a -DRT_REGION_TIMERS=1 build, printed on stderr by the library:
# render once more with fd 2 pointed at a temporary file (same process, no child)
import tempfile
sys.stderr.flush()
with tempfile.TemporaryFile() as f:
    saved = os.dup(2)
    os.dup2(f.fileno(), 2)
    try:
        gpu.render(params, camera)
    finally:
        os.dup2(saved, 2)
        os.close(saved)
    f.seek(0)
    err = f.read().decode()
print("".join(l + "\n" for l in err.splitlines() if l.startswith("[rt-turns]")), end="")
line_ = [l for l in err.splitlines() if l.startswith("[rt-regions]")]
if not line_: sys.exit("library was not built with -DRT_REGION_TIMERS=1:\n" + err[-2000:])
st["region_cycles"] = [int(x) for x in line_[-1].split()[1:]]
names = "LOAD FINISH RECORD TEXTURE EMIT METAL DIEL ISO LAMBERT L_LIGHTS L_COSINE L_PDF B_MEDIA B_RECTS B_ALWAYS B_REST STORE NODE LEAF FETCH CAMERA SHADE_ALL BEGIN_ALL TRAV_ALL".split()
rc = dict(zip(names, st["region_cycles"]))
rays = st["rays"]
print(wl, "rays", rays, "paths", st["paths"], "node visits/ray %.2f" % (st["node_visits"] / rays), "prim tests/ray %.2f" % (sum(st["prim_tests"]) / rays),
      "occupancy node %.3f leaf %.3f" % (st["phase_lanes"][0] / max(1, st["phase_slots"][0]), st["phase_lanes"][1] / max(1, st["phase_slots"][1])),
      "node steps/ray %.2f leaf steps/ray %.2f (wave-level, x64 lanes)" % (st["phase_slots"][0] / rays, st["phase_slots"][1] / rays))
top_s = ["LOAD", "SHADE_ALL", "BEGIN_ALL", "STORE"]; top_t = ["TRAV_ALL", "FETCH", "NODE", "LEAF"]
tot = sum(rc[k] for k in top_s + ["TRAV_ALL", "CAMERA"])
def line(k, indent=""): print("%s%-10s %6.2f%%  %8.1f wave-cycles/ray" % (indent, k, 100.0 * rc[k] / tot, rc[k] / rays))
print("-- shade kernel"); [line(k) for k in top_s]
print("   inside shade_hit:"); [line(k, "   ") for k in ("RECORD", "EMIT", "METAL", "DIEL", "ISO", "LAMBERT")]
print("      inside lambertian:"); [line(k, "      ") for k in ("TEXTURE", "L_LIGHTS", "L_COSINE", "L_PDF")]
print("   inside begin_trace:"); [line(k, "   ") for k in ("B_MEDIA", "B_RECTS", "B_ALWAYS", "B_REST")]
line("FINISH")
print("-- traverse kernel"); [line(k) for k in top_t]
print("-- generate: camera"); line("CAMERA")

# the tail kernel's own timers (sums over its waves), if it ran
tl = [l for l in err.splitlines() if l.startswith("[rt-tail-regions]")]
if tl:
    tr = dict(zip(names, [int(x) for x in tl[-1].split()[1:]]))
    slow = [int(l.split()[1]) for l in err.splitlines() if l.startswith("[rt-tail-slowest-wave-cycles]")][-1]
    print("-- tail kernel: slowest wave %d cycles; sums over its waves (wave-cycles):" % slow)
    for k in ("TRAV_ALL", "NODE", "LEAF", "SHADE_ALL", "RECORD", "METAL", "DIEL", "LAMBERT", "TEXTURE", "L_LIGHTS", "L_COSINE", "L_PDF", "BEGIN_ALL"):
        print("   %-10s %12d" % (k, tr[k]))
