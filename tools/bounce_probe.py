"""usage: bounce_probe.py WORKLOAD SPP DEPTH_A DEPTH_B — work done by the bounces between two max_ray_depths (counting variant):
rays, node visits, box tests and primitive tests per ray of bounces A+1..B."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtiow_loader
rt = rtiow_loader.load()
wl, spp, da, db = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
scene, params, camera, sampler = rt.configs.build(wl, samples_per_pixel=spp)
gpu = rt.backend.HipBackend(0); gpu.upload_scene(scene.flatten()); gpu.upload_sampler(sampler)
res = {}
for d in (da, db):
    _, st = gpu.render(params.replace(max_ray_depth=d), camera, stats=True)
    res[d] = st
a, b = res[da], res[db]
rays = b["rays"] - a["rays"]
print(wl, "bounces %d..%d: rays %d (%.2f%% of paths per bounce)" % (da + 1, db, rays, 100.0 * rays / (db - da) / b["paths"]),
      "node visits/ray %.2f" % ((b["node_visits"] - a["node_visits"]) / max(1, rays)),
      "box tests/ray %.2f" % ((b["box_tests"] - a["box_tests"]) / max(1, rays)),
      "prim tests/ray %.2f" % ((sum(b["prim_tests"]) - sum(a["prim_tests"])) / max(1, rays)),
      "hits/ray %.3f" % ((b["accepted_hits"] - a["accepted_hits"]) / max(1, rays)),
      "shades", [y - x for x, y in zip(a["shades"], b["shades"])])
print("   all bounces: node visits/ray %.2f prim tests/ray %.2f" % (b["node_visits"] / b["rays"], sum(b["prim_tests"]) / b["rays"]))
