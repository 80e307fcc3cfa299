"""Development aid: find where a GPU path and the oracle's part ways.  Renders a scene on both, picks a pixel whose ray
counts differ, narrows it to one sample, prints the oracle's bounces (orc_set_trace) and asks the GPU's first-hit hook
about each of those rays.  One process, no children.  It lives under tests/ because it uses the oracle (test infrastructure).
Usage (on a GPU box):  python tests/diverge_tool.py"""
import os, re, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle_ffi as O
import test_gpu_parity as T

rt = O.rt
scene, params = T.quadric_playground(rt)
camera = params.camera()
sampler = rt.host.MultiJitteredSampler(params.samples_per_pixel)
gpu = rt.backend.HipBackend(0)
gpu.upload_scene(scene.flatten()); gpu.upload_sampler(sampler)
oracle = O.OracleScene(scene)
_, st, g_rays = gpu.render(params, camera, stats=True, want_pixel_rays=True)
_, c_total, c_rays = oracle.render(params, camera, sampler, want_pixel_rays=True)
ys, xs = np.nonzero(g_rays != c_rays)
print("pixels that differ:", len(ys), "gpu rays", st["rays"], "oracle", c_total)


def traced(x, y, k):
    """The oracle's stderr lines for one pixel sample."""
    sys.stderr.flush()
    with tempfile.TemporaryFile() as f:
        saved = os.dup(2)
        os.dup2(f.fileno(), 2)
        try:
            O.lib().orc_set_trace(1)
            oracle.render(params, camera, sampler, spp_begin=k, spp_count=1, tile=(x, y, x + 1, y + 1), threads=1)
        finally:
            O.lib().orc_set_trace(0)
            os.dup2(saved, 2)
            os.close(saved)
        f.seek(0)
        return f.read().decode().splitlines()


for y, x in list(zip(ys.tolist(), xs.tolist()))[:3]:
    for k in range(params.samples_per_pixel):
        tile = (x, y, x + 1, y + 1)
        _, s1, gr = gpu.render(params, camera, stats=True, want_pixel_rays=True, spp_begin=k, spp_count=1, tile=tile)
        _, ct, cr = oracle.render(params, camera, sampler, want_pixel_rays=True, spp_begin=k, spp_count=1, tile=tile)
        if gr[y, x] == cr[y, x]:
            continue
        print("pixel", x, y, "sample", k, "gpu rays", gr[y, x], "oracle rays", cr[y, x])
        for line in traced(x, y, k):
            m = re.match(r"\[orc\] depth (\d+) ray (.*?) \| (.*?) \| (\S+) (.*)", line)
            if not m:
                continue
            o = [float(v) for v in m.group(2).split()]; d = [float(v) for v in m.group(3).split()]; tm = float(m.group(4))
            ray = np.array([[o[0]], [o[1]], [o[2]], [d[0]], [d[1]], [d[2]], [tm]], np.float32)
            pid, t, n, ff, uv = gpu.trace_first_hit(ray)
            print(" ", line[6:])
            print("      gpu first hit: prim", int(pid[0]) if pid[0] != 0xFFFFFFFF else "miss", "t %.9g" % t[0], "front", int(ff[0]), "n", n[0])
        break
