"""The C-ABI driven by a compiled C program (integration/c_host.c, built by `make -C integration` / __graft_entry__.build()):
config load, scene build, flatten, upload, two progressive rt_render calls, rt_resolve and the parity hook — none of it through
ctypes.  Its image must equal, bit for bit, what the Python binding renders for the same scene, seed and sampler (the C program
also carries the _Static_asserts on struct layout that the Rust mirrors of integration/gpu_ffi.rs are written against)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "integration", "c_host")
CONFIG = os.path.join(ROOT, "data", "config", "raytracer.config.ron")


@pytest.mark.parametrize("scene,workload,width,spp,extra", [("CornellBox", "C2", 40, 4, []), ("RandomWorld", "C1", 64, 4, ["noalloc"])])
def test_c_host_renders_what_the_python_binding_renders(rt, gpu, tmp_path, scene, workload, width, spp, extra):
    assert os.path.exists(EXE), "integration/c_host is not built: run __graft_entry__.build()"
    out = tmp_path / "img.f32"
    # a child process with its own context on the same GPU (nothing of this process is inherited: plain fork + exec of a C program)
    r = subprocess.run([EXE, CONFIG, scene, str(width), str(spp), str(out)] + extra, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    cfg = rt.host.RaytracerConfig.load()
    params = cfg.scene_params(scene).replace(image_width=width, samples_per_pixel=spp)
    W, H = params.image_width, params.image_height
    assert (info["width"], info["height"]) == (W, H)
    data = np.fromfile(out, np.float32)
    assert data.size == 2 * W * H * 4
    sums, rgba = data[:W * H * 4].reshape(H, W, 4), data[W * H * 4:].reshape(H, W, 4)
    sc = rt.host.Scene.build(scene, rt.configs.SEED, None)
    sampler = rt.host.MultiJitteredSampler(spp, seed=rt.configs.SEED)
    gpu.upload_scene(sc.flatten())
    gpu.upload_sampler(sampler)
    camera = params.camera()
    ref, _ = gpu.render(params, camera)
    assert np.array_equal(sums.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(rgba.view(np.uint32), gpu.resolve(ref, spp).view(np.uint32))
    _, st = gpu.render(params, camera, spp_begin=spp // 2, spp_count=spp - spp // 2, stats=True)
    assert info["rays_second_half"] == st["rays"]
    o = np.array(list(camera.origin), np.float32)
    d = (np.array(list(camera.lower_left_corner), np.float32) + np.float32(0.5) * np.array(list(camera.horizontal), np.float32)
         + np.float32(0.5) * np.array(list(camera.vertical), np.float32) - o).astype(np.float32)
    ray = np.concatenate([o, d, [0.0]]).astype(np.float32).reshape(7, 1)
    pid, t, _, ff, _ = gpu.trace_first_hit(ray)
    assert info["centre_prim"] == int(pid[0]) and info["front_face"] == int(ff[0])
    assert np.float32(info["centre_t"]) == t[0]
