"""How much of a wave's time is lost to uneven per-pixel work?  (dev tool, runs on the GPU box)"""
import sys
import numpy as np
import rtiow_loader
rt = rtiow_loader.load()
gpu = rt.backend.HipBackend(0)
for name, spp in (("C1", 0), ("C2", 64), ("C3", 32), ("C4", 8)):
    kw = {"samples_per_pixel": spp} if spp else {}
    scene, params, camera, sampler = rt.configs.build(name, **kw)
    gpu.upload_scene(scene.flatten()); gpu.upload_sampler(sampler)
    img, st, pr = gpu.render(params, camera, stats=True, want_pixel_rays=True)
    H, W = pr.shape
    Hp, Wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
    p = np.zeros((Hp, Wp), np.float64); p[:H, :W] = pr
    t = p.reshape(Hp // 8, 8, Wp // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    eff = t.sum() / (64 * t.max(1).sum())
    print(name, "rays/path %.2f" % (st["rays"] / st["paths"]), "tile balance (sum / 64*max) %.3f" % eff,
          "nodes/ray %.1f" % (st["node_visits"] / st["rays"]), "prim tests/ray %.1f" % (sum(st["prim_tests"]) / st["rays"]),
          "shades", [round(x / st["rays"], 3) for x in st["shades"]])
