"""Round 3 of the independent numpy restatement (tests/golden/make_golden.py "round 3"): Hyperboloid with its `+ 2`, the sampler
table (points, shuffles, index table, concentric map), the checker, Camera::new and the four projections, Cylinder's pdf_value /
random.  Host library and oracle here (CPU); the same vectors through the C-ABI on the GPU in the tests marked `gpu`."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_ffi as O

rt = O.rt
ffi = rt._ffi
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
INF = float("inf")
PROJ = {"perspective": 0, "orthographic": 1, "fisheye": 2, "spherical_panoramic": 3}
ODD, EVEN, BG = (0.9, 0.1, 0.2), (0.2, 0.8, 0.4), (0.1, 0.2, 0.3)


def f32(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def ray(o, d, time=0.0):
    return np.array([[o[0]], [o[1]], [o[2]], [d[0]], [d[1]], [d[2]], [time]], np.float32)


# ------------------------------------------------------------------ sampler table (sampling.rs:26-149, 230-253)

@pytest.mark.parametrize("tab", KAT["sampler"], ids=lambda t: "spp%d" % t["spp"])
def test_sampler_table_against_the_numpy_restatement(tab):
    s = rt.host.MultiJitteredSampler(tab["spp"], sets=tab["sets"], seed=tab["seed"])
    samples, disk, shuffled = s.arrays()
    assert samples.view(np.uint32).tolist() == tab["samples"]           # points after the x and the y shuffle: bit for bit
    assert shuffled.tolist() == tab["shuffled"]
    rp = f32(tab["disk_r_phi"])                                         # (r, phi * 0.25): the map's branch and quotient are exact,
    want = np.stack([rp[:, 0] * np.cos(rp[:, 1]), rp[:, 0] * np.sin(rp[:, 1])], 1)   # sin / cos belong to the math library in use
    np.testing.assert_allclose(disk, want, rtol=0, atol=3e-7)
    # every set is a permutation of the sub-cell columns and of the rows (what multi-jittering means), shuffles or not
    n = tab["spp"]
    for p in range(tab["sets"]):
        blk = samples[p * n:(p + 1) * n]
        assert sorted(np.floor(blk[:, 0] * n).astype(int)) == list(range(n)) and sorted(np.floor(blk[:, 1] * n).astype(int)) == list(range(n))
        assert sorted(shuffled[p * n:(p + 1) * n]) == list(range(n))


# ------------------------------------------------------------------ checker (checker_texture.rs:36-46)

def test_checker_against_the_numpy_restatement():
    for c in KAT["checker"]:
        s = rt.host.Scene()
        tex = s.checker(s.solid(*ODD), s.solid(*EVEN), c["freq"])
        m = s.lambertian(tex)
        s.set_roots(s.hittable_list([s.sphere((0, 0, 0), 1, m)]), s.hittable_list([s.xz_rect(0, 1, 0, 1, 9, m)]))
        got = O.OracleScene(s).texture_value(tex, 0.0, 0.0, c["p"])
        assert np.allclose(got, ODD if c["odd"] else EVEN, atol=0), c


# ------------------------------------------------------------------ Camera::new and the projections (camera.rs:39-184)

def golden_camera(frame):
    cam = ffi.rt_camera()
    for k, v in frame["camera"].items():
        if k == "lens_radius":
            cam.lens_radius = float(f32(v))
        else:
            setattr(cam, k, (C.c_float * 3)(*[float(x) for x in f32(v)]))
    cam.time0, cam.time1 = 0.0, 1.0
    return cam


def frame_params(frame):
    lk = frame["look"]
    return rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=frame["width"], aspect_ratio=lk["aspect"], samples_per_pixel=frame["spp"], max_ray_depth=1,
        look_from=lk["from"], look_at=lk["at"], world_up=lk["up"], vertical_fov=lk["vfov"], aperture=lk["aperture"], focus_dist=lk["focus"],
        background=BG, projection=PROJ[frame["projection"]], psi_max=frame["psi_max_degrees"], lambda_max=frame["lambda_max_degrees"])


def frame_scene(frame):
    r = frame["rect"]
    s = rt.host.Scene()
    lamp = s.diffuse_light(s.checker(s.solid(*ODD), s.solid(*EVEN), frame["checker_freq"]))
    s.set_roots(s.hittable_list([s.xy_rect(r["a0"], r["a1"], r["b0"], r["b1"], r["k"], lamp)]),
                s.hittable_list([s.xz_rect(0, 1, 0, 1, 100, s.diffuse_light(s.solid(0, 0, 0)))]))
    return s


def expected_frame(frame):
    table = np.array([BG, ODD, EVEN, (0.0, 0.0, 0.0)], np.float32)
    cls = np.frombuffer(frame["classes"].encode(), np.uint8) - ord("0")
    return table[cls].reshape(frame["height"], frame["width"], 3)


def check_frame(img, frame):
    """img: H x W x 4 sums of ONE sample at depth 1.  Perspective / Orthographic: every pixel; FishEye / SphericalPanoramic go
    through sin / cos (the math library in use), so a pixel on an edge of the rectangle or of a checker cell may fall either way."""
    want = expected_frame(frame)
    got = img[..., :3]
    if frame["projection"] == "fisheye":
        outside = (got == 0).all(-1) & (want == np.float32(BG)).all(-1)      # outside the disk: nothing was added (camera.rs:231-233)
        want = np.where(outside[..., None], 0.0, want).astype(np.float32)
        assert outside[0, 0] and outside[-1, -1] and not outside[frame["height"] // 2, frame["width"] // 2]
    bad = int((got != want).any(-1).sum())
    assert bad <= (0 if frame["projection"] in ("perspective", "orthographic") else 2), "%s: %d pixels differ" % (frame["projection"], bad)


@pytest.mark.parametrize("frame", KAT["projection_frames"], ids=lambda f: f["projection"])
def test_camera_new_and_projections_oracle(frame):
    params = frame_params(frame)
    assert params.image_height == frame["height"]
    host_cam = params.camera()                          # rth_camera_new against the numpy Camera::new: tan() is the library's
    cam = golden_camera(frame)
    for k in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w"):
        np.testing.assert_allclose(np.array(list(getattr(host_cam, k))), np.array(list(getattr(cam, k))), rtol=2e-6, atol=1e-6, err_msg=k)
    assert host_cam.lens_radius == cam.lens_radius
    sampler = rt.host.MultiJitteredSampler(frame["spp"], sets=frame["sets"], seed=frame["seed"])
    # the four sample rays of the frame through the oracle's projection: Perspective / Orthographic bit for bit
    p = O.orc_render_params()
    p.image_width, p.image_height, p.camera = frame["width"], frame["height"], cam
    p.projection, p.psi_max, p.lambda_max = params.projection, params.psi_max_radians, params.lambda_max_radians
    for r in frame["rays"]:
        out = (C.c_float * 6)()
        dk = f32(r["disk"])
        assert O.lib().orc_project(C.byref(p), float(f32(r["u"])), float(f32(r["v"])), float(dk[0]), float(dk[1]), C.byref(out)) == 1
        got, want = np.array(list(out), np.float32), np.concatenate([f32(r["origin"]), f32(r["direction"])])
        if frame["projection"] in ("perspective", "orthographic"):
            assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), r
        else:
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    img, _ = O.OracleScene(frame_scene(frame)).render(params, cam, sampler, seed=frame["seed"], spp_begin=0, spp_count=1, threads=2)
    check_frame(img, frame)


# ------------------------------------------------------------------ Hyperboloid (hyperboloid.rs:26-170)

def hyperboloid_scene(case):
    s = rt.host.Scene()
    m = s.lambertian(s.solid(0.5, 0.5, 0.5))
    s.set_roots(s.hittable_list([s.hyperboloid(case["p1"], case["p2"], case["phi_max"], m)]),
                s.hittable_list([s.xz_rect(0, 1, 0, 1, 100, s.diffuse_light(s.solid(0, 0, 0)))]))
    return s


def check_hyperboloid(first_hit, case):
    t_max = INF if case["t_max"] == "inf" else case["t_max"]
    pid, t, n, ff, uv = first_hit(ray(case["o"], case["d"]), 0.001, t_max)
    want = case["hit"]
    if want is None:
        assert pid[0] == 0xFFFFFFFF, case
        return
    assert pid[0] == 0 and int(t.view(np.uint32)[0]) == want["t"], case          # which root, which attempt: exact
    assert ff[0] == want["front_face"] and int(uv.view(np.uint32)[0, 1]) == want["v"], case
    np.testing.assert_allclose(n[0], want["normal"], rtol=0, atol=3e-6)          # through atan2, sin and cos of the library in use
    assert uv[0, 0] == pytest.approx(want["phi"] / case["phi_max"], abs=2e-6)    # u = phi / phi_max with the `+ 2` of :116 in it


@pytest.mark.parametrize("case", KAT["hyperboloid"], ids=lambda c: "attempt%s" % (c["hit"] and c["hit"]["attempt"]))
def test_hyperboloid_oracle(case):
    check_hyperboloid(O.OracleScene(hyperboloid_scene(case)).first_hit, case)


def test_hyperboloid_coefficients_in_the_flattened_slot():
    """ah, ch of Hyperboloid::new (hyperboloid.rs:36-54) as the flattener writes them into the slot (words 12, 13)."""
    for case in KAT["hyperboloid"]:
        flat = hyperboloid_scene(case).flatten()
        slots = np.ctypeslib.as_array(flat.desc.slots, (flat.desc.n_slots * 16,)).view(np.uint32).reshape(-1, 16)
        assert slots[0, 12:14].tolist() == case["coefficients"], case


# ------------------------------------------------------------------ Cylinder::pdf_value / random (cylinder.rs:134-161)

def cylinder_light_scene(c, origin=(0.0, 0.0, 0.0)):
    s = rt.host.Scene()
    grey = s.lambertian(s.solid(0.5, 0.5, 0.5))
    dark = s.diffuse_light(s.solid(0, 0, 0))
    floor = s.xz_rect(-50, 50, -50, 50, float(origin[1]), grey)
    s.set_roots(s.hittable_list([floor]), s.hittable_list([s.cylinder(c["radius"], c["zmin"], c["zmax"], c.get("phi_max", float(np.float32(np.pi) * np.float32(2))), dark)]))
    return s


@pytest.mark.parametrize("c", KAT["cylinder_pdf"], ids=lambda c: "r%g" % c["radius"])
def test_cylinder_pdf_value_oracle(c):
    got = O.OracleScene(cylinder_light_scene(c)).lights_pdf_value(np.array(c["origin"], np.float32), np.array(c["v"], np.float32))
    assert np.float32(got).view(np.uint32) == c["pdf"], (got, float(f32(c["pdf"])))


@pytest.mark.parametrize("c", KAT["cylinder_random"], ids=lambda c: "r1=%g" % c["r1"])
def test_cylinder_random_oracle(c):
    """One Lambertian bounce on a floor through the point `origin` with the light branch forced (xi0 < 0.5), one light, so the
    numbers drawn are: mixture choice, random_int (one member: any), r1, r2 (pdf.rs:72-78, hittable_list.rs:71-75, types.rs:153-163)."""
    o = c["origin"]
    oracle = O.OracleScene(cylinder_light_scene(c, o))
    b = oracle.bounce((o[0], o[1] + 1.0, o[2]), (0.0, -1.0, 0.0), [0.25, 0.0, c["r1"], c["r2"]])
    assert b["kind"] == 3 and b["used"] == 4 and np.allclose(b["p"], o, atol=1e-6)
    np.testing.assert_allclose(b["dir"], c["direction"], rtol=0, atol=3e-6)


# ================================================================== the same vectors through the C-ABI, on the GPU

@pytest.mark.gpu
@pytest.mark.parametrize("frame", KAT["projection_frames"], ids=lambda f: f["projection"])
def test_projections_sampler_and_checker_on_the_gpu(gpu, frame):
    """One sample per pixel at depth 1 of a rectangle that emits a checker: every pixel's sum is the colour the numpy restatement
    of the sampler look-up, the projection, the rectangle test and the checker predicts."""
    params, cam = frame_params(frame), golden_camera(frame)
    sampler = rt.host.MultiJitteredSampler(frame["spp"], sets=frame["sets"], seed=frame["seed"])
    gpu.upload_scene(frame_scene(frame).flatten())
    gpu.upload_sampler(sampler)
    img = gpu.render(params, cam, seed=frame["seed"], spp_begin=0, spp_count=1)[0]
    check_frame(img, frame)


@pytest.mark.gpu
def test_hyperboloid_on_the_gpu(gpu):
    for case in KAT["hyperboloid"]:
        gpu.upload_scene(hyperboloid_scene(case).flatten())
        t_max = INF if case["t_max"] == "inf" else case["t_max"]
        check_hyperboloid(lambda r, t_min, t_max_: gpu.trace_first_hit(r, t_min, t_max_), case)


@pytest.mark.gpu
def test_cylinder_pdf_and_random_on_the_gpu(gpu):
    """Cylinder in `lights`: the GPU's bounce off the floor below `origin` is the oracle's (same draws: equal per-pixel ray counts
    and sums), and the oracle's pdf_value / random are the numpy restatement's (tests above)."""
    c = KAT["cylinder_random"][0]
    s = cylinder_light_scene(c, c["origin"])
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=32, aspect_ratio=1.0, samples_per_pixel=4, max_ray_depth=3, look_from=(0.0, 8.0, 6.0), look_at=(0.0, 3.0, 0.0),
        vertical_fov=40.0, background=(0.3, 0.4, 0.5))
    cam, sampler = params.camera(), rt.host.MultiJitteredSampler(4)
    gpu.upload_scene(s.flatten())
    gpu.upload_sampler(sampler)
    got, st, rays = gpu.render(params, cam, stats=True, want_pixel_rays=True)
    want, n_rays, want_rays = O.OracleScene(s).render(params, cam, sampler, want_pixel_rays=True)
    assert np.array_equal(rays, want_rays) and st["rays"] == n_rays and st["light_pdf_evals"] > 0
    np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=2e-4, atol=1e-6)
