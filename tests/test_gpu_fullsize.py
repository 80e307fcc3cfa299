"""GPU tests at BASELINE.json's full sizes, through the C-ABI.

The oracle cannot render these frames in seconds, so full-size checks are (a) size-independent
properties — determinism, tile == frame, sample ranges add up, ray counters consistent — and
(b) oracle parity on a window / sample range of the full-size frame (same pixel indices, same
sample indices, hence the same random streams as in the full frame)."""
import json
import os

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def setup(rt, gpu, name, **kw):
    scene, params, camera, sampler = rt.configs.build(name, **kw)
    gpu.upload_scene(scene.flatten())
    gpu.upload_sampler(sampler)
    return scene, params, camera, sampler


def oracle_window(rt, gpu, scene, params, camera, sampler, tile, spp_begin=0, spp_count=None, oracle=None):
    oracle = oracle or O.OracleScene(scene)
    g, st, gr = gpu.render(params, camera, tile=tile, spp_begin=spp_begin, spp_count=spp_count, stats=True, want_pixel_rays=True)
    c, total, cr = oracle.render(params, camera, sampler, tile=tile, spp_begin=spp_begin, spp_count=spp_count, want_pixel_rays=True, threads=16)
    assert st["rays"] == total and np.array_equal(gr, cr)
    both_nan = np.isnan(g) & np.isnan(c)
    np.testing.assert_allclose(np.where(both_nan, 0, g), np.where(both_nan, 0, c), rtol=3e-4, atol=1e-5)
    return g


def test_golden_vectors_through_the_c_abi(rt, gpu):
    """tests/golden/kat.json sphere + rect vectors via rt_trace_first_hit: bit-exact."""
    def run(build, case):
        s = rt.host.Scene()
        m = s.lambertian(s.solid(0.5, 0.5, 0.5))
        objs = build(s, m)
        s.set_roots(s.hittable_list(objs), s.hittable_list([s.xz_rect(0, 1, 0, 1, 100, s.diffuse_light(s.solid(0, 0, 0)))]))
        gpu.upload_scene(s.flatten())
        r = np.array([[x] for x in case["o"] + case["d"] + [0.0]], np.float32)
        t_max = float("inf") if case["t_max"] == "inf" else case["t_max"]
        pid, t, n, ff, uv = gpu.trace_first_hit(r, case["t_min"], t_max)
        want = case["hit"]
        if want is None:
            assert pid[0] == 0xFFFFFFFF, case
            return
        def same(got, bits):      # bit-exact, except that a NaN only has to be a NaN (x86 and gfx950 differ in its sign bit)
            w = np.array(bits, np.uint32).view(np.float32)
            return all((np.isnan(a) and np.isnan(b)) or a.view(np.uint32) == b.view(np.uint32) for a, b in zip(np.ravel(got), np.ravel(w)))
        assert pid[0] == 0 and same(t[:1], [want["t"]]), case
        assert same(n[0], want["normal"]) and ff[0] == want["front_face"], case
        if "uv" in want:
            assert same(uv[0], want["uv"]), case
    for c in KAT["sphere"]:
        run(lambda s, m: [s.sphere(c["center"], c["radius"], m)], c)
    for c in KAT["rect"]:
        mk = {"xy": "xy_rect", "xz": "xz_rect", "yz": "yz_rect"}[c["kind"]]
        run(lambda s, m: [getattr(s, mk)(c["a0"], c["a1"], c["b0"], c["b1"], c["k"], m)], c)


def test_round2_golden_vectors_through_the_c_abi(rt, gpu):
    """The first-hit families of the widened numpy restatement (tests/golden/make_golden.py "round 2") via rt_trace_first_hit,
    bit-exact: triangle test + cull + uv, MovingSphere at given times, Translate(RotateY(..)) records.  A mesh is compared with
    the triangle test alone: the kernels' own padded boxes stand in for the mesh's Aabb::hit, which rejects rays whose box test
    runs into 0 * inf ("BVH conservativeness") — one vector here (the ray through a vertex with d.x = d.y = 0)."""
    from test_oracle_kat import check_first_hit, tri_scene, moving_scene, wrapped_scene
    differ = 0
    for c in KAT["triangle"]:
        gpu.upload_scene(tri_scene(c["verts"]).flatten())
        check_first_hit(gpu.trace_first_hit, dict(c, hit=c["triangle_alone"]), uv=True)
        differ += c["hit"] != c["triangle_alone"]
    assert differ == 1
    for c in KAT["moving_sphere"]:
        gpu.upload_scene(moving_scene(c).flatten())
        check_first_hit(gpu.trace_first_hit, c, time=c["time"])
    for c in KAT["wrapped"]:
        gpu.upload_scene(wrapped_scene(c).flatten())
        check_first_hit(gpu.trace_first_hit, c)


def test_quadric_golden_vectors_through_the_c_abi(rt, gpu):
    """tests/golden/kat.json quadric + Transform vectors via rt_trace_first_hit: bit-exact."""
    from test_oracle_kat import check_quadric, quadric_scene
    for case in KAT["quadric"]:
        gpu.upload_scene(quadric_scene(case).flatten())
        check_quadric(gpu.trace_first_hit, case)


def test_c1_book1_full_frame(rt, gpu):
    scene, params, camera, sampler = setup(rt, gpu, "C1")
    assert (params.image_width, params.image_height, sampler.spp) == (1200, 675, 10)
    full, st, rays = gpu.render(params, camera, stats=True, want_pixel_rays=True)
    assert st["paths"] == 1200 * 675 * 10 and st["rays"] == int(rays.sum(dtype=np.uint64))
    assert (full[..., 3] == 10).all() and np.isfinite(full[..., :3]).mean() > 0.9999
    again, _ = gpu.render(params, camera)
    assert np.array_equal(again.view(np.uint32), full.view(np.uint32))                       # deterministic
    tile, _ = gpu.render(params, camera, tile=(400, 200, 531, 333))
    assert np.array_equal(tile[200:333, 400:531].view(np.uint32), full[200:333, 400:531].view(np.uint32))
    assert not tile[:200].any() and not tile[:, :400].any()
    a, _ = gpu.render(params, camera, spp_begin=0, spp_count=4)
    b, _ = gpu.render(params, camera, spp_begin=4, spp_count=6, out=a)                        # accumulates into `a`
    np.testing.assert_allclose(np.nan_to_num(b), np.nan_to_num(full), rtol=1e-5, atol=1e-6)
    oracle_window(rt, gpu, scene, params, camera, sampler, (560, 300, 688, 396))             # 128x96 window, all 10 spp


def test_c1_headline_frame_every_pixel_against_the_oracle(rt, gpu):
    """BASELINE's headline configuration whole: 1200 x 675 x 10 spp, 21 M rays — ray total, every pixel's ray count and every sum
    against the oracle (3 s on the box's 16 host threads; the larger frames are compared whole by tests/fullframe_parity.py, by hand:
    profiles/r3/fullframe_parity.txt)."""
    scene, params, camera, sampler = setup(rt, gpu, "C1")
    oracle_window(rt, gpu, scene, params, camera, sampler, (0, 0, params.image_width, params.image_height))


def test_c2_cornell_full_frame(rt, gpu):
    scene, params, camera, sampler = setup(rt, gpu, "C2")
    assert (params.image_width, params.image_height, sampler.spp) == (600, 600, 1000)
    full, st = gpu.render(params, camera, stats=True)
    assert st["paths"] == 600 * 600 * 1000
    mean = np.nan_to_num(full[..., :3]).sum() / (600 * 600 * 3 * 1000)
    assert 0.05 < mean < 3.0
    win = oracle_window(rt, gpu, scene, params, camera, sampler, (280, 380, 312, 412), spp_begin=500, spp_count=100)
    # the same window of the full frame restricted to that sample range is not separable; check the tile path instead
    tile, _ = gpu.render(params, camera, tile=(280, 380, 312, 412))
    assert np.array_equal(tile[380:412, 280:312].view(np.uint32), full[380:412, 280:312].view(np.uint32))
    assert win[380:412, 280:312, 3].min() == 100


def test_c2_cornell_whole_frame_at_full_spp_against_the_oracle(rt, gpu):
    """BASELINE configs[1] as configured, whole: 600 x 600 x 1000 spp, 1.99 G rays — ray total, every pixel's ray count, the NaN
    pixels and every sum against the oracle (16 s on the box's 16 host threads: a Cornell ray costs the oracle 13 objects)."""
    scene, params, camera, sampler = setup(rt, gpu, "C2")
    assert (params.image_width, params.image_height, sampler.spp) == (600, 600, 1000)
    oracle_window(rt, gpu, scene, params, camera, sampler, (0, 0, 600, 600))


def test_c3_book2_windows_of_the_full_frame(rt, gpu):
    scene, params, camera, sampler = setup(rt, gpu, "C3")
    assert (params.image_width, params.image_height, sampler.spp) == (800, 800, 10000)
    oracle = O.OracleScene(scene)
    # three 24x24 windows (ground + fog, glass/medium sphere, the 1000-sphere cube), 64 of the 10 000 samples each
    for tile, s0 in (((100, 650, 124, 674), 0), ((300, 420, 324, 444), 5000), ((560, 130, 584, 154), 9936)):
        oracle_window(rt, gpu, scene, params, camera, sampler, tile, spp_begin=s0, spp_count=64, oracle=oracle)
    a, st = gpu.render(params, camera, spp_begin=0, spp_count=8, stats=True)
    assert st["paths"] == 800 * 800 * 8 and st["medium_tests"] == 2 * st["rays"]            # both media are probed by every ray
    b, _ = gpu.render(params, camera, spp_begin=0, spp_count=8)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_c5_book2_4k_one_rank_of_eight(rt, gpu):
    """BASELINE configs[4]: 3840x2160, 16 384 spp over 8 GPUs.  One GPU here plays rank 5: the head of its sample range on the
    whole 4K frame, oracle parity on windows of it, and the sum over ranks (all eight ranges, a few samples each, accumulated
    in rank order as the reduce does) equal to one render of the same samples."""
    scene, params, camera, sampler = setup(rt, gpu, "C5")
    assert (params.image_width, params.image_height, sampler.spp) == (3840, 2160, 16384)
    ranges = [rt.sharding.sample_range(r, 8, sampler.spp) for r in range(8)]
    assert ranges[5] == (10240, 2048) and sum(n for _, n in ranges) == 16384
    b5 = ranges[5][0]
    part, st = gpu.render(params, camera, spp_begin=b5, spp_count=2, stats=True)
    assert st["paths"] == 3840 * 2160 * 2 and st["medium_tests"] == 2 * st["rays"]
    assert (part[..., 3] == 2).all() and np.isfinite(part[..., :3]).mean() > 0.9999
    again, _ = gpu.render(params, camera, spp_begin=b5, spp_count=2)
    assert np.array_equal(again.view(np.uint32), part.view(np.uint32))
    tile, _ = gpu.render(params, camera, tile=(3001, 1203, 3130, 1300), spp_begin=b5, spp_count=2)      # a window of it, not on 8-pixel bounds
    assert np.array_equal(tile[1203:1300, 3001:3130].view(np.uint32), part[1203:1300, 3001:3130].view(np.uint32))
    oracle = O.OracleScene(scene)
    for win, s0 in (((1900, 1700, 1924, 1724), b5), ((3816, 2136, 3840, 2160), 16384 - 8), ((0, 0, 24, 24), 0)):
        oracle_window(rt, gpu, scene, params, camera, sampler, win, spp_begin=s0, spp_count=8, oracle=oracle)
    # eight ranks, one sample each, summed in rank order == the reduce's result; the RNG is keyed by the absolute sample index,
    # so each rank's image is a function of its range alone
    win = (1800, 900, 2056, 1028)
    total = np.zeros((2160, 3840, 4), np.float32)
    for b, _ in ranges:
        one, _ = gpu.render(params, camera, tile=win, spp_begin=b, spp_count=1)
        total += one
    acc = np.zeros_like(total)
    for b, _ in ranges:
        gpu.render(params, camera, tile=win, spp_begin=b, spp_count=1, out=acc)
    y0, y1, x0, x1 = win[1], win[3], win[0], win[2]
    np.testing.assert_allclose(np.nan_to_num(total[y0:y1, x0:x1]), np.nan_to_num(acc[y0:y1, x0:x1]), rtol=1e-5, atol=1e-6)
    assert (acc[y0:y1, x0:x1, 3] == 8).all() and not acc[:y0].any()


def test_c4_mesh_2m_triangles(rt, gpu):
    scene, params, camera, sampler = setup(rt, gpu, "C4")
    f = scene.flatten()
    assert f.desc.n_prims == 100 * 22646 + 2 and (params.image_width, params.image_height, sampler.spp) == (1920, 1080, 256)
    # first-hit ids against the oracle's brute force over all 2.26 M triangles (nearest mode), 384 primary rays
    rng = np.random.default_rng(3)
    n = 384
    o = np.array(list(camera.origin), np.float32)
    d = (np.array(list(camera.lower_left_corner), np.float32)[None] + rng.random((n, 1), np.float32) * np.array(list(camera.horizontal), np.float32)[None]
         + rng.random((n, 1), np.float32) * np.array(list(camera.vertical), np.float32)[None] - o[None]).astype(np.float32)
    rays = np.zeros((7, n), np.float32)
    rays[0:3] = o[:, None]; rays[3:6] = d.T
    g = gpu.trace_first_hit(rays)
    c = O.OracleScene(scene, exhaustive=True).first_hit(rays)
    assert np.array_equal(g[0], c[0]) and (g[0] >= 2).mean() > 0.05                          # some rays hit teapots
    for k in (1, 2, 4):
        assert np.array_equal(g[k].view(np.uint32), c[k].view(np.uint32))
    a, st = gpu.render(params, camera, spp_begin=0, spp_count=4, stats=True)
    assert st["paths"] == 1920 * 1080 * 4 and st["prim_tests"][5] > 0
    tile, _ = gpu.render(params, camera, spp_begin=0, spp_count=4, tile=(900, 500, 1000, 580))
    assert np.array_equal(tile[500:580, 900:1000].view(np.uint32), a[500:580, 900:1000].view(np.uint32))


@pytest.mark.parametrize("name,windows", [
    # GeometricPrimitives as shipped (raytracer.config.ron:88-107): 800x600, 256 spp, SphericalPanoramic over 360 x 180 degrees
    ("GP", (((250, 300, 282, 332), 0), ((520, 260, 552, 292), 128), ((384, 380, 416, 412), 192))),
    # Blocks as shipped (:130-149): 800x600, 256 spp, aperture 0.5
    ("BLOCKS", (((300, 200, 332, 232), 0), ((470, 330, 502, 362), 200))),
])
def test_quadric_scenes_at_their_shipped_configs(rt, gpu, name, windows):
    scene, params, camera, sampler = setup(rt, gpu, name)
    assert (params.image_width, params.image_height, sampler.spp) == (800, 600, 256)
    oracle = O.OracleScene(scene)
    for tile, s0 in windows:                                      # 32x32 windows, 48 of the 256 samples each
        oracle_window(rt, gpu, scene, params, camera, sampler, tile, spp_begin=s0, spp_count=48, oracle=oracle)
    a, st = gpu.render(params, camera, spp_begin=0, spp_count=8, stats=True)
    assert st["paths"] == 800 * 600 * 8 and st["node_visits"] == 0       # about 20 leaves, all in the always-list
    if name == "GP":
        assert all(st["prim_tests"][k] > 0 for k in (6, 7, 8))           # cylinders, cones, paraboloids
    tile, _ = gpu.render(params, camera, spp_begin=0, spp_count=8, tile=(96, 40, 300, 177))
    assert np.array_equal(tile[40:177, 96:300].view(np.uint32), a[40:177, 96:300].view(np.uint32))
