"""Where a pixel's paths part ways between the GPU and the oracle: the sample whose ray count differs, and the first depth limit
at which it does (test infrastructure, like tests/fullframe_parity.py).  usage: python3 tests/pixel_probe.py C3 100 x,y [x,y ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # (this file lives in tests/: the oracle is loaded by tests only)
import numpy as np
import rtiow_loader
import oracle_ffi as O

rt = rtiow_loader.load()
name, spp = sys.argv[1], int(sys.argv[2])
scene, params, camera, sampler = rt.configs.build(name, samples_per_pixel=spp)
gpu = rt.backend.HipBackend(0)
gpu.upload_scene(scene.flatten())
gpu.upload_sampler(sampler)
oracle = O.OracleScene(scene)
for arg in sys.argv[3:]:
    x, y = (int(v) for v in arg.split(","))
    tile = (x, y, x + 1, y + 1)
    def both(p, k):
        g, st, gr = gpu.render(p, camera, tile=tile, spp_begin=k, spp_count=1, stats=True, want_pixel_rays=True)
        c, total, cr = oracle.render(p, camera, sampler, tile=tile, spp_begin=k, spp_count=1, want_pixel_rays=True, threads=1)
        return int(gr[y, x]), int(cr[y, x]), g[y, x, :3], c[y, x, :3]
    for k in range(spp):
        gr, cr, gs, cs = both(params, k)
        if gr == cr:
            continue
        print("pixel %d,%d sample %d: gpu %d oracle %d rays; sample value gpu %s oracle %s" % (x, y, k, gr, cr, gs, cs), flush=True)
        for d in range(1, params.max_ray_depth + 1):
            gr, cr, gs, cs = both(params.replace(max_ray_depth=d), k)
            if gr != cr or not np.array_equal(np.nan_to_num(gs), np.nan_to_num(cs)):
                print("   first difference at depth limit %d: rays gpu %d oracle %d, value gpu %s oracle %s" % (d, gr, cr, gs, cs))
                gp, cp, gps, cps = both(params.replace(max_ray_depth=d - 1), k)
                print("   at depth limit %d: rays %d / %d, value %s / %s" % (d - 1, gp, cp, gps, cps), flush=True)
                break
gpu.close()
