"""usage: depth_probe.py WORKLOAD SPP D1 D2 — work counters of bounces D1+1..D2 (what the long-lived paths cost per ray)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtiow_loader
rt = rtiow_loader.load()
wl, spp, d1, d2 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
scene, params, camera, sampler = rt.configs.build(wl, samples_per_pixel=spp)
gpu = rt.backend.HipBackend(0); gpu.upload_scene(scene.flatten()); gpu.upload_sampler(sampler)
def run(d):
    _, st = gpu.render(params.replace(max_ray_depth=d), camera, stats=True)
    return st
a, b = run(d1), run(d2)
rays = b["rays"] - a["rays"]
print("rays in bounces %d..%d: %d (%.1f %% of all %d)" % (d1 + 1, d2, rays, 100.0 * rays / b["rays"], b["rays"]))
print("  node visits / ray %.2f (all rays: %.2f)   box tests / ray %.2f (%.2f)" % (
    (b["node_visits"] - a["node_visits"]) / rays, b["node_visits"] / b["rays"], (b["box_tests"] - a["box_tests"]) / rays, b["box_tests"] / b["rays"]))
pt = [y - x for x, y in zip(a["prim_tests"], b["prim_tests"])]
print("  prim tests / ray by type", [round(v / rays, 2) for v in pt], "all rays:", [round(v / b["rays"], 2) for v in b["prim_tests"]])
print("  shades by material / ray", [round((y - x) / rays, 3) for x, y in zip(a["shades"], b["shades"])], " accepted hits / ray %.3f" % ((b["accepted_hits"] - a["accepted_hits"]) / rays))
print("  tex evals / ray", [round((y - x) / rays, 3) for x, y in zip(a["tex_evals"], b["tex_evals"])])
