"""usage: stats.py WORKLOAD SPP — work counters of the counting variant, per ray."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtiow_loader
rt = rtiow_loader.load()
wl, spp = sys.argv[1], int(sys.argv[2])
scene, params, camera, sampler = rt.configs.build(wl, samples_per_pixel=spp)
gpu = rt.backend.HipBackend(0); gpu.upload_scene(scene.flatten()); gpu.upload_sampler(sampler)
_, st = gpu.render(params, camera, stats=True)
r = st["rays"]
print(wl, "rays", r, "paths", st["paths"], "rays/path %.2f" % (r / st["paths"]))
for k in ("node_visits", "prim_tests", "medium_tests", "light_pdf_evals", "shades", "tex_evals", "accepted_hits"):
    v = st[k]
    print("  %-16s" % k, [round(x / r, 4) for x in v] if isinstance(v, list) else round(v / r, 4))
