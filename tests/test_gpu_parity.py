"""GPU parity: the HIP path (through the C-ABI of include/rt_api.h) against the CPU oracle.

Bars (north_star):
  * first hit: primitive ids bit-exact, t / normals within 1e-5 relative  — here they are
    bit-identical because both sides do the same single f32 operations;
  * images: same scene + seed schedule.  Path geometry is bit-identical (checked through the
    per-pixel ray counters, which must be EQUAL), radiance sums differ only by the association
    of the throughput products: rtol 2e-4 on un-normalised sums, and the resolved 8-bit image
    may differ by at most 1 LSB in at most 0.1 % of the channels; per-pixel RMSE of the
    resolved [0,1] image below 1e-4.
"""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def primary_and_bounce_rays(rt, scene, params, camera, n, seed=7):
    """Seeded rays: pinhole-ish primaries through random pixels plus rays leaving their hit points."""
    rng = np.random.default_rng(seed)
    W, H = params.image_width, params.image_height
    o = np.array(list(camera.origin), np.float32)
    ll = np.array(list(camera.lower_left_corner), np.float32)
    hor = np.array(list(camera.horizontal), np.float32)
    ver = np.array(list(camera.vertical), np.float32)
    s = rng.random(n, dtype=np.float32)
    t = rng.random(n, dtype=np.float32)
    d = (ll[None, :] + s[:, None] * hor[None, :] + t[:, None] * ver[None, :] - o[None, :]).astype(np.float32)
    rays = np.zeros((7, n), np.float32)
    rays[0:3] = o[:, None]
    rays[3:6] = d.T
    rays[6] = rng.random(n, dtype=np.float32)
    return rays


def bounce_rays(rays, t, normal, hit, seed=11):
    rng = np.random.default_rng(seed)
    n = rays.shape[1]
    p = rays[0:3] + t[None, :] * rays[3:6]
    v = rng.normal(size=(3, n)).astype(np.float32)
    flip = (v * normal.T).sum(0) < 0
    v[:, flip] *= -1
    out = rays.copy()
    out[0:3] = p
    out[3:6] = v
    out[6] = rng.random(n, dtype=np.float32)
    return np.ascontiguousarray(out[:, hit])


def check_first_hit(gpu, oracle, rays):
    g = gpu.trace_first_hit(rays)
    c = oracle.first_hit(rays)
    names = ["prim_id", "t", "normal", "front_face", "uv"]
    assert np.array_equal(g[0], c[0]), "primitive ids differ on %d of %d rays" % ((g[0] != c[0]).sum(), len(g[0]))
    for k in (1, 2, 4):
        np.testing.assert_allclose(g[k], c[k], rtol=1e-5, atol=0, err_msg=names[k])
    assert np.array_equal(g[3], c[3])
    bit_exact = all(np.array_equal(g[k].view(np.uint32), c[k].view(np.uint32)) for k in (1, 2, 4))
    return g, bit_exact


def compare_images(rt, gpu, oracle, params, camera, sampler, **kw):
    g_img, stats, g_rays = gpu.render(params, camera, stats=True, want_pixel_rays=True, **kw)
    c_img, c_total, c_rays = oracle.render(params, camera, sampler, want_pixel_rays=True, **kw)
    assert stats["rays"] == c_total, "ray totals differ: gpu %d oracle %d" % (stats["rays"], c_total)
    assert np.array_equal(g_rays, c_rays), "per-pixel ray counts differ on %d pixels" % (g_rays != c_rays).sum()
    both_nan = np.isnan(g_img) & np.isnan(c_img)
    np.testing.assert_allclose(np.where(both_nan, 0, g_img), np.where(both_nan, 0, c_img), rtol=2e-4, atol=1e-6)
    spp = sampler.spp
    g8 = (np.clip(gpu.resolve(g_img, spp)[..., :3], 0, 1) * 255).astype(np.int32)
    c8 = (np.clip(O.resolve(c_img, spp)[..., :3], 0, 1) * 255).astype(np.int32)
    diff = np.abs(g8 - c8)
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    # north_star's second bar, stated: per-pixel RMSE of the resolved [0,1] image (same scene and seed schedule) < 1e-4
    g_res, c_res = np.nan_to_num(gpu.resolve(g_img, spp)[..., :3]), np.nan_to_num(O.resolve(c_img, spp)[..., :3])
    rmse = float(np.sqrt(np.mean((g_res.astype(np.float64) - c_res) ** 2)))
    assert rmse < 1e-4, rmse
    # the plain (no counters) kernel must give the same sums as the counting variant
    g_plain, _ = gpu.render(params, camera, **kw)
    assert np.array_equal(g_plain.view(np.uint32), g_img.view(np.uint32))
    return g_img, c_img, stats


@pytest.mark.parametrize("name,width,spp,n_rays", [("C1", 240, 9, 1 << 16), ("C2", 96, 16, 1 << 16), ("C3", 96, 16, 1 << 16),
                                                   ("C4", 64, 4, 1 << 13), ("GP", 96, 16, 1 << 16), ("BLOCKS", 96, 16, 1 << 16),
                                                   ("TWO", 64, 9, 1 << 12), ("PERLIN", 96, 9, 1 << 14), ("LIGHT", 64, 16, 1 << 14), ("EARTH", 64, 4, 1 << 12),
                                                   ("SMOKE", 96, 16, 1 << 14)])   # ConstantMedium over Translate(RotateY(Block)) boundaries
def test_image_and_first_hit_small(rt, gpu, name, width, spp, n_rays):
    extra = {"mesh_grid": 1} if name == "C4" else {}
    scene, params, camera, sampler = rt.configs.build(name, image_width=width, samples_per_pixel=spp, **extra)
    flat = scene.flatten()
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    oracle = O.OracleScene(scene)
    rays = primary_and_bounce_rays(rt, scene, params, camera, n_rays)
    g, exact = check_first_hit(gpu, oracle, rays)
    assert exact, "t / normal / uv are expected to be bit-identical"
    hit = g[0] != 0xFFFFFFFF
    assert hit.mean() > 0.25
    rays2 = bounce_rays(rays, g[1], g[2], hit)
    check_first_hit(gpu, oracle, rays2)
    compare_images(rt, gpu, oracle, params, camera, sampler)


def test_degenerate_rays_follow_the_reference_fold(rt, gpu):
    """Zero direction components, rays lying in a rectangle's plane (NaN distance, accepted by
    rectangles.rs:25-32), NaN / zero directions: the GPU must return what the list fold returns."""
    def build(order):
        s = rt.host.Scene()
        m = s.lambertian(s.solid(0.5, 0.5, 0.5))
        objs = {"rect": lambda: s.xy_rect(0, 1, 0, 1, 0, m), "sphere": lambda: s.sphere((3, 0.5, 0), 0.5, m),
                "far_rect": lambda: s.yz_rect(-5, 5, -5, 5, 9, m)}
        world = [objs[k]() for k in order]
        s.set_roots(s.hittable_list(world), s.hittable_list([s.xz_rect(0, 1, 0, 1, 100, s.diffuse_light(s.solid(0, 0, 0)))]))
        return s
    rays = np.array([
        [0.5, 0.5, 0, 1, 0, 0, 0],          # in the plane z = 0 of `rect`: t = NaN there; sphere and far_rect are real hits
        [0.5, 0.5, 0, 1, 0.01, 0, 0],       # one zero component, nothing degenerate
        [-1, 0.5, 0, 1, 0, 0, 0],           # same line, starting outside the rectangle: still 0/0
        [0.5, 0.5, 1, 1, 0, 0, 0],          # parallel above the plane: -inf / +inf, rejected
        [0.5, 0.5, -1, 0, 0, 1, 0],         # straight at the rectangle: two zero components
        [0.5, 0.5, -1, 0, 0, 0, 0],         # null direction
        [0.5, 0.5, -1, np.nan, 0, 1, 0],    # NaN direction: every leaf "hits"
        [np.inf, 0.5, -1, 0, 0, 1, 0],
    ], np.float32).T.copy()
    for order in (("rect", "sphere", "far_rect"), ("sphere", "far_rect", "rect"), ("far_rect", "rect", "sphere")):
        scene = build(order)
        gpu.upload_scene(scene.flatten())
        g = gpu.trace_first_hit(rays)
        c = O.OracleScene(scene).first_hit(rays)
        assert np.array_equal(g[0], c[0]), (order, g[0], c[0])
        assert np.array_equal(np.isnan(g[1]), np.isnan(c[1])), order
        ok = ~np.isnan(c[1])
        assert np.array_equal(g[1][ok], c[1][ok]) and np.array_equal(g[3], c[3])
    # a BVH scene: zero out one direction component of ordinary rays
    scene, params, camera, sampler = rt.configs.build("C3", image_width=64, samples_per_pixel=4)
    gpu.upload_scene(scene.flatten())
    r = primary_and_bounce_rays(rt, scene, params, camera, 6000)
    r[3, :2000] = 0; r[4, 2000:4000] = 0; r[5, 4000:] = 0
    r[1, :100] = 0.0          # origins ON the y = 0 plane of the ground boxes (d.y != 0 there: t = 0 < t_min)
    # Not tested against the BVH oracle: rays lying exactly IN a face plane of leaves that sit inside a
    # reference BvhNode (camera z = -600 is such a plane).  Their NaN candidates are folded in the
    # BvhNode's leaf order, which follows the reference's RANDOM split axes (bvh.rs:17) — the reference
    # itself does not reproduce that case from run to run.  Move those rays off the plane.
    r[2, 4000:] = -612.3
    g = gpu.trace_first_hit(r)
    c = O.OracleScene(scene).first_hit(r)
    assert np.array_equal(g[0], c[0]) and np.array_equal(g[1], c[1]) and np.array_equal(g[3], c[3])
    assert (g[0] != 0xFFFFFFFF).mean() > 0.2


@pytest.mark.parametrize("projection,name", [(1, "Orthographic"), (2, "FishEye"), (3, "SphericalPanoramic")])
def test_other_projections_match_the_oracle(rt, gpu, projection, name):
    """camera.rs:98-184: same paths as the oracle (equal per-pixel ray counts), sums within 2e-4."""
    scene, params, camera, sampler = rt.configs.build("C1", image_width=96, samples_per_pixel=4)
    # look_from raised so that the Orthographic view plane stays above the ground sphere: ray origins INSIDE the
    # r=1000 sphere make paths of 1000-unit segments, whose f32 sphere tests accept far-away small spheres by
    # cancellation (tests/test_oracle_kat.py::test_cancellation_false_positive) -- hits a BVH culls
    params = params.replace(projection=projection, psi_max=150.0, lambda_max=170.0, look_from=(13.0, 4.0, 3.0))
    assert params.projection_name == name
    camera = params.camera()
    gpu.upload_scene(scene.flatten())
    gpu.upload_sampler(sampler)
    got, st, rays = gpu.render(params, camera, stats=True, want_pixel_rays=True)
    want, n_rays, want_rays = O.OracleScene(scene).render(params, camera, sampler, want_pixel_rays=True)
    assert np.array_equal(rays, want_rays) and st["rays"] == n_rays
    np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=2e-4, atol=1e-6)
    assert np.all(got[..., 3] == sampler.spp)
    if projection == 2:      # FishEye: the corners of the frame are outside the unit disk and trace nothing
        assert rays[0, 0] == 0 and rays[-1, -1] == 0 and rays[rays.shape[0] // 2, rays.shape[1] // 2] > 0


def test_rays_from_far_away_see_what_the_flat_list_sees(rt, gpu):
    """sphere.rs:46-66 in f32 accepts small spheres thousands of radii away from a ray's origin (cancellation: DESIGN.md section 3,
    tests/test_oracle_kat.py::test_cancellation_false_positive); the reference's flat `world` takes those hits, a walk that culls
    by the spheres' boxes would not.  Rays that start beyond rt_scene_desc::far_limit walk the tree with every box grown by the
    distance at which that can happen (rt_kernels.h, ray_fat): the known ray gets the oracle's hit from rt_trace_first_hit, and
    the Orthographic view whose plane dips into the r = 1000 ground sphere (the default look_from) traces the oracle's paths,
    pixel for pixel."""
    scene, params, camera, sampler = rt.configs.build("C1", image_width=96, samples_per_pixel=4)
    flat = scene.flatten()
    assert flat.desc.far_limit == pytest.approx(0.2 * 1024)          # RandomWorld is a flat list; its smallest spheres have r = 0.2
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    ray = np.array([[-647.327393, -1432.85925, -627.37561, 909.112427, 1986.85925, 878.007812, 0.662694097]], np.float32).T.copy()
    oracle = O.OracleScene(scene)
    want = oracle.first_hit(ray)
    got = gpu.trace_first_hit(ray)
    assert want[0][0] != 0xFFFFFFFF and got[0][0] == want[0][0] and got[1][0] == want[1][0]
    near = ray.copy(); near[0:6] /= 2048.0                            # the same line at a human scale: the walk, and no hit
    assert gpu.trace_first_hit(near)[0][0] == oracle.first_hit(near)[0][0]
    params = params.replace(projection=1)
    assert params.projection_name == "Orthographic"
    camera = params.camera()
    got, st, rays = gpu.render(params, camera, stats=True, want_pixel_rays=True)
    want, n_rays, want_rays = oracle.render(params, camera, sampler, want_pixel_rays=True)
    assert np.array_equal(rays, want_rays) and st["rays"] == n_rays
    np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=2e-4, atol=1e-6)


def quadric_playground(rt):
    """Every quadric kind, partial sweeps, FlipFace on either side of a Transform, metal and glass quadrics (so that
    paths leave the surfaces at grazing angles) next to 31 spheres that go through the BVH."""
    two_pi = float(np.float32(np.pi) * np.float32(2))
    rng = np.random.default_rng(5)
    s = rt.host.Scene()
    grey = s.lambertian(s.checker(s.solid(0.2, 0.3, 0.1), s.solid(0.9, 0.9, 0.9), 4.0))
    metal = s.metal(0.8, 0.8, 0.9, 0.1)
    glass = s.dielectric(1.5)
    # the floor is a rectangle, as in the reference's scenes with these primitives (main.rs:839-848): paths that get under
    # a r = 1000 ground SPHERE run into the f32 cancellation false-positives of DESIGN.md §3
    objs = [s.xz_rect(-1000, 1000, -1000, 1000, 0, grey)]
    for i in range(30):
        c = rng.uniform(-6, 6, 3).astype(np.float32)
        objs.append(s.sphere((float(c[0]), 0.3, float(c[2])), 0.3, s.lambertian(s.solid(*rng.uniform(0.2, 0.9, 3).tolist()))))

    def trs(t, angle, axis, scale):
        a = np.radians(angle)
        c, si = np.cos(a), np.sin(a)
        x, y, z = axis
        r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * si, x * z * (1 - c) + y * si, 0],
                      [y * x * (1 - c) + z * si, c + y * y * (1 - c), y * z * (1 - c) - x * si, 0],
                      [z * x * (1 - c) - y * si, z * y * (1 - c) + x * si, c + z * z * (1 - c), 0], [0, 0, 0, 1]])
        m = np.eye(4); m[:3, 3] = t
        return (m @ r @ np.diag(list(scale) + [1.0])).astype(np.float32)

    objs.append(s.transform(s.cylinder(1, -0.5, 0.5, two_pi, metal), trs((-3, 1.5, 0), 90, (1, 0, 0), (0.8, 0.8, 3))))
    objs.append(s.transform(s.cylinder(1, -0.5, 0.5, 4.0, grey), trs((3, 1.0, 1), 70, (0, 1, 0), (1, 1, 2))))
    objs.append(s.transform(s.cone(1, 1, two_pi, grey), trs((-1.5, 0, -4), -90, (1, 0, 0), (1.5, 1.5, 2.5))))
    objs.append(s.transform(s.cone(1, 1, 5.0, glass), trs((2, 0, -3), -90, (1, 0, 0), (1, 1, 2))))
    objs.append(s.transform(s.paraboloid(0.5, -0.5, 1.0, two_pi, metal), trs((-2, 2.0, -3), 90, (1, 0, 0), (2, 2, 2))))
    objs.append(s.transform(s.hyperboloid((1, 0, -1), (0.6, 0.5, 1), 5.5, grey), trs((5, 1.2, -1), 90, (1, 0, 0), (1, 1, 1.2))))
    objs.append(s.paraboloid(1.0, 0.0, 1.0, two_pi, grey))                                     # a quadric with no Transform at all
    objs.append(s.transform(s.block((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), grey), trs((-5, 0.8, 2), 30, (0, 1, 0), (1, 1.6, 1))))
    objs.append(s.flip_face(s.transform(s.xz_rect(-1, 1, -1, 1, 0, s.diffuse_light(s.solid(6, 6, 6))), trs((0, 6, 0), 0, (0, 1, 0), (3, 1, 3)))))
    objs.append(s.transform(s.flip_face(s.sphere((0, 0, 0), 1, glass)), trs((1, 0.7, 5), 0, (0, 1, 0), (0.7, 0.7, 0.7))))
    light = s.xz_rect(-3, 3, -3, 3, 6, s.diffuse_light(s.solid(0, 0, 0)))
    s.set_roots(s.hittable_list(objs), s.hittable_list([light]))
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=128, aspect_ratio=1.6, samples_per_pixel=16, look_from=(0.0, 4.0, 13.0), look_at=(0.0, 1.0, 0.0),
        vertical_fov=40.0, background=(0.3, 0.4, 0.6))
    return s, params


def test_quadrics_and_transforms_next_to_a_bvh(rt, gpu):
    """Quadrics live in the always-list (they may accept t < t_min, cone.rs:61-65) while everything else, slots under a
    general Transform included, goes through the BVH."""
    s, params = quadric_playground(rt)
    flat = s.flatten()
    assert flat.desc.n_bvh_nodes > 0 and flat.desc.always_count == 7        # the quadrics; block, light and glass ball under their Transforms are BVH leaves
    camera = params.camera()
    sampler = rt.host.MultiJitteredSampler(16)
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    oracle = O.OracleScene(s)
    rays = primary_and_bounce_rays(rt, s, params, camera, 1 << 15)
    g, exact = check_first_hit(gpu, oracle, rays)
    assert exact
    hit = g[0] != 0xFFFFFFFF
    quadric_ids = set(range(31, 38))
    assert len(quadric_ids & set(g[0][hit].tolist())) == 7, "every quadric must be seen by some ray"
    check_first_hit(gpu, oracle, bounce_rays(rays, g[1], g[2], hit))
    compare_images(rt, gpu, oracle, params, camera, sampler)


def test_many_transformed_blocks_go_through_the_bvh(rt, gpu):
    """48 blocks and 12 spheres, each under its own general Transform (rotation about a random axis, non-uniform scale):
    368 leaves in the BVH, none in the always-list; the second instance of the traversal kernel (with the Transform code)."""
    rng = np.random.default_rng(9)
    s = rt.host.Scene()
    grey = s.lambertian(s.checker(s.solid(0.2, 0.3, 0.1), s.solid(0.9, 0.9, 0.9), 2.0))
    objs = [s.xz_rect(-1000, 1000, -1000, 1000, 0, grey)]

    def trs(t, angle, axis, scale):
        a = np.radians(angle)
        c, si = np.cos(a), np.sin(a)
        x, y, z = np.asarray(axis) / np.linalg.norm(axis)
        r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * si, x * z * (1 - c) + y * si, 0],
                      [y * x * (1 - c) + z * si, c + y * y * (1 - c), y * z * (1 - c) - x * si, 0],
                      [z * x * (1 - c) - y * si, z * y * (1 - c) + x * si, c + z * z * (1 - c), 0], [0, 0, 0, 1]])
        m = np.eye(4); m[:3, 3] = t
        return (m @ r @ np.diag(list(scale) + [1.0])).astype(np.float32)

    for i in range(60):
        pos = (float(rng.uniform(-8, 8)), float(rng.uniform(0.8, 3.0)), float(rng.uniform(-8, 8)))
        m = trs(pos, float(rng.uniform(0, 360)), rng.normal(size=3), rng.uniform(0.5, 1.6, 3).tolist())
        mtl = [s.lambertian(s.solid(*rng.uniform(0.2, 0.9, 3).tolist())), s.metal(0.8, 0.8, 0.7, 0.05), s.dielectric(1.5)][i % 3]
        inner = s.sphere((0, 0, 0), 0.5, mtl) if i % 5 == 4 else s.block((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), mtl)
        objs.append(s.transform(inner, m))
    objs.append(s.flip_face(s.xz_rect(-4, 4, -4, 4, 9, s.diffuse_light(s.solid(5, 5, 5)))))
    s.set_roots(s.hittable_list(objs), s.hittable_list([s.xz_rect(-4, 4, -4, 4, 9, s.diffuse_light(s.solid(0, 0, 0)))]))
    flat = s.flatten()
    assert flat.desc.always_count == 0 and flat.desc.n_slots == 1 + 48 * 6 + 12 + 1 and flat.desc.n_matrices == 60
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=128, aspect_ratio=1.6, samples_per_pixel=16, look_from=(0.0, 6.0, 16.0), look_at=(0.0, 1.5, 0.0),
        vertical_fov=40.0, background=(0.3, 0.4, 0.6))
    camera = params.camera()
    sampler = rt.host.MultiJitteredSampler(16)
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    oracle = O.OracleScene(s)
    rays = primary_and_bounce_rays(rt, s, params, camera, 1 << 15)
    g, exact = check_first_hit(gpu, oracle, rays)
    assert exact
    hit = g[0] != 0xFFFFFFFF
    check_first_hit(gpu, oracle, bounce_rays(rays, g[1], g[2], hit))
    img, _, st = compare_images(rt, gpu, oracle, params, camera, sampler)
    assert st["node_visits"] > 0
    # and on a tree built on the device
    gpu.upload_scene(s.flatten(device_bvh=True))
    img2, _ = gpu.render(params, camera)
    assert np.array_equal(img2.view(np.uint32), img.view(np.uint32))


def nested_wrappers_scene(rt, many):
    """Translate / RotateY / Transform in the orders the one-record form cannot hold (transform.rs:14-213), around every
    kind of leaf: rectangles and spheres (record inline in the slot), triangles and a moving sphere (record fetched from
    sc.xforms), a quadric (always-list), with FlipFace at different depths."""
    rng = np.random.default_rng(21)
    s = rt.host.Scene()
    # not 0.9 for green: floor x sky = 0.9 * 0.4 resolves to 153.0, a knife edge for the 8-bit comparison
    grey = s.lambertian(s.checker(s.solid(0.2, 0.3, 0.1), s.solid(0.8, 0.85, 0.9), 3.0))
    red = s.lambertian(s.solid(0.7, 0.2, 0.2))
    metal = s.metal(0.8, 0.8, 0.9, 0.05)
    glass = s.dielectric(1.5)
    lamp = s.diffuse_light(s.solid(5, 5, 5))
    shear = np.eye(4, dtype=np.float32); shear[0, 1] = 0.3; shear[:3, 3] = (0.5, 0.2, -0.5); shear[2, 2] = 1.4
    scale = np.diag([0.8, 1.3, 0.8, 1.0]).astype(np.float32); scale[:3, 3] = (-2.0, 0.0, 1.0)
    block = lambda m: s.block((-0.5, 0.0, -0.5), (0.5, 1.0, 0.5), m)
    objs = [s.xz_rect(-1000, 1000, -1000, 1000, 0, grey)]
    objs.append(s.rotate_y(s.translate(block(red), (3.0, 0.0, 0.0)), 35.0))                                   # R(T(.))
    objs.append(s.translate(s.translate(s.sphere((0, 0.6, 0), 0.6, glass), (-1.0, 0.0, 2.0)), (0.5, 0.3, 0.0)))   # T(T(.))
    objs.append(s.rotate_y(s.rotate_y(block(metal), 20.0), -50.0))                                            # R(R(.))
    objs.append(s.transform(s.translate(s.rotate_y(block(grey), 25.0), (1.0, 0.0, -3.0)), scale))             # M(T(R(.)))
    objs.append(s.translate(s.rotate_y(s.transform(s.sphere((0, 1, 0), 0.7, metal), shear), 60.0), (-4.0, 0.0, -1.0)))  # T(R(M(.)))
    objs.append(s.transform(s.transform(block(red), shear), scale))                                           # M(M(.))
    objs.append(s.translate(s.flip_face(s.rotate_y(s.flip_face(s.translate(s.sphere((0, 0.5, 0), 0.5, glass), (0, 0, 4.0))), 15.0)), (2.0, 0.0, 0.0)))
    objs.append(s.flip_face(s.rotate_y(s.flip_face(s.transform(s.flip_face(s.translate(s.xz_rect(-2, 2, -2, 2, 7, lamp), (0.5, 0, 0))), np.eye(4, dtype=np.float32))), 10.0)))
    objs.append(s.rotate_y(s.translate(s.moving_sphere((0, 0.4, 0), (0, 0.8, 0), 0.0, 1.0, 0.4, red), (-3.0, 0.0, 3.0)), -20.0))
    verts = np.array([[0, 0, 0, 0, 0, 1, 0, 0], [1.5, 0, 0, 0, 0, 1, 1, 0], [0, 1.5, 0, 0, 0, 1, 0, 1], [1.5, 1.5, 0.3, 0, 0, 1, 1, 1]], np.float32)
    idx, node = np.array([0, 1, 2, 2, 1, 3], np.uint32), np.array([[0, 6]], np.uint32)
    objs.append(s.rotate_y(s.translate(s.triangle_mesh(verts, idx, node, None, grey), (4.0, 0.2, 1.0)), 20.0))
    quarter = np.array([[1, 0, 0, 1.0], [0, 0, -1, 1.0], [0, 1, 0, -4.0], [0, 0, 0, 1]], np.float32)
    objs.append(s.translate(s.transform(s.cylinder(0.5, -0.8, 0.8, 2.0 * np.pi, metal), quarter), (0.0, 0.3, 0.0)))     # T(M(quadric))
    deep = s.sphere((0, 0.5, 0), 0.5, metal)  # twelve wrappers around one sphere: T R T M R T T R M T R T (a chain of eleven records)
    for k, kind in enumerate("TRTMRTTRMTRT"):
        if kind == "T":
            deep = s.translate(deep, (-0.6 + 0.05 * k, 0.02 * k, 0.35))
        elif kind == "R":
            deep = s.rotate_y(deep, 7.0 + 3.0 * k)
        else:
            deep = s.transform(deep, scale if k == 3 else shear)
    objs.append(deep)
    for i in range(40 if many else 0):       # enough leaves for a real tree, each under a chain of its own
        pos = (float(rng.uniform(-9, 9)), 0.0, float(rng.uniform(-9, 9)))
        inner = s.sphere((0, 0.3, 0), 0.3, s.lambertian(s.solid(*rng.uniform(0.2, 0.9, 3).tolist()))) if i % 2 else block(grey)
        objs.append(s.rotate_y(s.translate(s.rotate_y(inner, float(rng.uniform(0, 90))), pos), float(rng.uniform(-30, 30))))
    s.set_roots(s.hittable_list(objs), s.hittable_list([s.xz_rect(-2, 2, -2, 2, 7, s.diffuse_light(s.solid(0, 0, 0)))]))
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=128, aspect_ratio=1.6, samples_per_pixel=16, look_from=(0.0, 5.0, 14.0), look_at=(0.0, 1.0, 0.0),
        vertical_fov=42.0, background=(0.3, 0.4, 0.6))
    return s, params


@pytest.mark.parametrize("many", [False, True])
def test_wrappers_nested_in_any_order(rt, gpu, many):
    """Chains of xform records: first hits (ids, t, normal, front_face, u v) and images against the oracle, whose wrapper
    classes nest by construction.  Without the 40 extra objects the leaves sit in a small tree; with them the chains
    are met by the traversal kernel's Transform instance too, and on a device-built tree."""
    s, params = nested_wrappers_scene(rt, many)
    flat = s.flatten()
    x = flat.xforms()
    assert (x[:, 0].view(np.int32) == 8).sum() >= 10
    camera = params.camera()
    sampler = rt.host.MultiJitteredSampler(16)
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    oracle = O.OracleScene(s)
    rays = primary_and_bounce_rays(rt, s, params, camera, 1 << 15)
    g, exact = check_first_hit(gpu, oracle, rays)
    assert exact
    hit = g[0] != 0xFFFFFFFF
    seen = set(g[0][hit].tolist())
    assert {7, 20, 27, 29, 32, 33} <= seen and seen & {30, 31}, "both spheres' kinds, the moving sphere, a triangle, the cylinder and the sphere under twelve wrappers must be seen"
    _, exact = check_first_hit(gpu, oracle, bounce_rays(rays, g[1], g[2], hit))
    assert exact
    img, _, st = compare_images(rt, gpu, oracle, params, camera, sampler)
    if many:
        assert st["node_visits"] > 0
        gpu.upload_scene(s.flatten(device_bvh=True))
        img2, _ = gpu.render(params, camera)
        assert np.array_equal(img2.view(np.uint32), img.view(np.uint32))


def test_paths_deeper_than_255_bounces(rt, gpu):
    """max_ray_depth is an unbounded i32 in the reference (camera.rs:253-262).  Inside a closed diffuse sphere with a tiny
    emitter and a zero-emission `lights` rectangle that is not in `world` (Book-1's arrangement, main.rs:187-196; here it
    lies inside the sphere, so a light-sampled direction still ends on the wall) almost every path runs to the depth limit: at depth 300 every such sample must issue 300 rays, as the oracle's does (a path's remaining depth once
    travelled in 8 bits and wrapped at 256)."""
    s = rt.host.Scene()
    wall = s.lambertian(s.solid(0.9, 0.9, 0.9))
    lamp = s.diffuse_light(s.solid(40.0, 30.0, 20.0))
    world = [s.sphere((0, 0, 0), 100.0, wall), s.sphere((30, 0, 0), 2.0, lamp)]
    for k in range(30):      # enough leaves for a BVH walk (more than the 24 a flat scan takes)
        world.append(s.sphere((-40 + 3.0 * k, -50 + 2.0 * k, 40 - 2.5 * k), 1.5, wall))
    s.set_roots(s.hittable_list(world), s.hittable_list([s.xz_rect(-10, 10, -10, 10, 50, s.diffuse_light(s.solid(0, 0, 0)))]))
    _, params, _, _ = rt.configs.build("C2", image_width=24, samples_per_pixel=4)
    params = params.replace(max_ray_depth=300, look_from=(0.0, 0.0, -60.0), look_at=(0.0, 0.0, 0.0), vertical_fov=60.0)
    camera = params.camera()
    sampler = rt.host.MultiJitteredSampler(4, seed=rt.configs.SEED)
    gpu.upload_scene(s.flatten())
    gpu.upload_sampler(sampler)
    g_img, c_img, stats = compare_images(rt, gpu, O.OracleScene(s), params, camera, sampler)
    assert stats["rays"] > 200 * stats["paths"], "most paths should reach the depth limit"
    assert stats["node_visits"] > 0 and stats["box_tests"] >= stats["node_visits"]


@pytest.mark.gpu
def test_wrapped_lights_forward_their_pdf_untransformed(rt, gpu):
    """`lights` may hold wrapped objects: Translate / RotateY / Transform forward pdf_value and random to the object they wrap
    WITHOUT transforming (transform.rs:48-54, 146-152, 206-212), so the direction that is sampled aims at where the inner object
    would be; FlipFace and Block keep the trait defaults (pdf 0, random = unit_x, hittable.rs:47-53).  All of it against the
    oracle, with per-pixel ray counts equal."""
    def build(with_defaults):
        s = rt.host.Scene()
        grey, red = s.lambertian(s.solid(0.7, 0.7, 0.7)), s.lambertian(s.solid(0.7, 0.2, 0.2))
        lamp, dark = s.diffuse_light(s.solid(9, 9, 9)), s.diffuse_light(s.solid(0, 0, 0))
        world = [s.xz_rect(-20, 20, -20, 20, 0, grey), s.sphere((-2, 1, 0), 1, red), s.block((1, 0, -1), (2.5, 1.5, 0.5), grey),
                 s.flip_face(s.translate(s.xz_rect(-1, 1, -1, 1, 5, lamp), (1.5, 0, 0.5))),          # the emitters, where the wrappers put them
                 s.rotate_y(s.sphere((4, 3, 0), 0.5, lamp), 40.0)]
        m = np.eye(4, dtype=np.float32); m[:3, 3] = (0, 1, -3); m[0, 0] = 2
        lights = [s.translate(s.xz_rect(-1, 1, -1, 1, 5, dark), (1.5, 0, 0.5)),       # sampled as the bare rectangle at x, z in [-1, 1]
                  s.rotate_y(s.sphere((4, 3, 0), 0.5, dark), 40.0),                   # ... as the sphere before the rotation
                  s.transform(s.yz_rect(0, 1, 0, 1, 2, dark), m),
                  s.translate(s.rotate_y(s.xy_rect(-1, 1, 2, 3, -4, dark), 15.0), (0, 0, 1))]   # two wrappers deep
        if with_defaults:
            lights += [s.flip_face(s.xz_rect(-1, 1, -1, 1, 5, dark)),                 # trait defaults: pdf 0, random = (1, 0, 0)
                       s.block((0, 0, 0), (1, 1, 1), dark)]
        s.set_roots(s.hittable_list(world), s.hittable_list(lights))
        return s
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=96, aspect_ratio=1.5, samples_per_pixel=16, look_from=(0.0, 3.0, 11.0), look_at=(0.5, 1.5, 0.0),
        vertical_fov=40.0, background=(0.02, 0.02, 0.03))
    camera, sampler = params.camera(), rt.host.MultiJitteredSampler(16)
    gpu.upload_sampler(sampler)
    s = build(False)
    flat = s.flatten()
    assert flat.desc.n_lights == 4
    gpu.upload_scene(flat)
    oracle = O.OracleScene(s)
    g_img, c_img, stats = compare_images(rt, gpu, oracle, params, camera, sampler)
    assert stats["light_pdf_evals"] > 0 and np.isfinite(g_img[..., :3]).all() and g_img[..., :3].max() > 1.0
    # forwarding means: the list's pdf is that of the bare inner objects
    bare = rt.host.Scene()
    d2 = bare.diffuse_light(bare.solid(0, 0, 0))
    bare.set_roots(bare.hittable_list([bare.sphere((0, 0, 0), 1, d2)]),
                   bare.hittable_list([bare.xz_rect(-1, 1, -1, 1, 5, d2), bare.sphere((4, 3, 0), 0.5, d2), bare.yz_rect(0, 1, 0, 1, 2, d2),
                                       bare.xy_rect(-1, 1, 2, 3, -4, d2)]))
    o_bare = O.OracleScene(bare)
    rng = np.random.default_rng(5)
    o = rng.uniform(-3, 3, (256, 3)).astype(np.float32); o[:, 1] = np.abs(o[:, 1])
    v = rng.normal(size=(256, 3)).astype(np.float32); v[:, 1] = np.abs(v[:, 1])
    vals = [oracle.lights_pdf_value(o[k], v[k]) for k in range(256)]
    assert vals == [o_bare.lights_pdf_value(o[k], v[k]) for k in range(256)] and max(vals) > 0
    # With the two trait-default entries many samples are NaN in the reference's arithmetic (random = unit_x lies in the floor's
    # plane and runs into +-inf / NaN distances and pdfs; a path that then ends on black is 0 * NaN on the way up the recursion).
    # The kernels' forward form must give NaN in exactly those pixels and the same numbers elsewhere (compare_images).
    s = build(True)
    flat = s.flatten()
    assert flat.desc.n_lights == 6
    gpu.upload_scene(flat)
    g_img, c_img, _ = compare_images(rt, gpu, O.OracleScene(s), params, camera, sampler)
    nan = np.isnan(g_img[..., :3]).any(-1)
    assert np.array_equal(nan, np.isnan(c_img[..., :3]).any(-1)) and 0 < nan.mean() < 0.9


@pytest.mark.gpu
def test_the_rare_folds_run_by_the_wave_give_the_list_fold(rt, gpu):
    """Rays that are not finite take the literal fold over every slot, rays with a zero direction component look at every
    rectangle for a NaN candidate (hittable_list.rs:40-57 with rectangles.rs:25-32 and sphere.rs:61-65 accepting NaN); the wave
    runs those loops together (rt_kernels.h coop_exhaustive, coop_rect_floor, finish_trace).  A scene of several hundred slots in
    which such rays are common: a FlipFace entry in `lights` makes every second lambertian bounce aim along (1, 0, 0), which lies
    in the floor's plane — a NaN distance, a NaN hit point, and from there a path of NaN rays — and a medium hands on rays with a
    time outside the shutter.  Image, ray total and per-pixel ray counts against the oracle.  (The clutter is a plain list: NaN
    candidates inside a reference BvhNode are folded in that node's random leaf order)"""
    s = rt.host.Scene()
    rng = np.random.default_rng(11)
    grey, red = s.lambertian(s.solid(0.7, 0.7, 0.7)), s.lambertian(s.solid(0.7, 0.2, 0.2))
    lamp, dark, glass = s.diffuse_light(s.solid(9, 9, 9)), s.diffuse_light(s.solid(0, 0, 0)), s.dielectric(1.5)
    clutter = []
    for k in range(150):
        c = (float(rng.uniform(-6, 6)), float(rng.uniform(0.2, 3)), float(rng.uniform(-6, 3)))
        clutter.append(s.sphere(c, float(rng.uniform(0.1, 0.3)), red if k % 3 else glass))
    for k in range(30):     # 180 rectangles.  They absorb, and none lies in the floor's plane: a ray along (1, 0, 0) that starts ON a
        # Block's face meets a NaN candidate inside the Block's own list of sides, which the kernels fold flat
        x, z = float(rng.uniform(-6, 5)), float(rng.uniform(-6, 2))
        clutter.append(s.block((x, 0.25 if k % 2 else 0.5, z), (x + 0.4, 0.9, z + 0.4), dark))
    fog = s.constant_medium(s.sphere((0, 1, 0), 2.5, glass), s.solid(0.8, 0.8, 0.9), 0.3)
    world = [s.xz_rect(-20, 20, -20, 20, 0, grey), s.xz_rect(-1, 1, -1, 1, 5, lamp), fog, s.moving_sphere((2, 1, 1), (2, 1.5, 1), 0, 1, 0.4, red)]
    world += clutter
    lights = [s.xz_rect(-1, 1, -1, 1, 5, dark), s.flip_face(s.xz_rect(-1, 1, -1, 1, 5, dark))]     # the second: pdf 0, random = (1, 0, 0)
    s.set_roots(s.hittable_list(world), s.hittable_list(lights))
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=96, aspect_ratio=1.5, samples_per_pixel=16, look_from=(0.0, 3.0, 11.0), look_at=(0.5, 1.5, 0.0),
        vertical_fov=40.0, background=(0.02, 0.02, 0.03))
    camera, sampler = params.camera(), rt.host.MultiJitteredSampler(16)
    flat = s.flatten()
    assert flat.desc.n_slots > 330 and flat.desc.n_rect_slots > 180 and flat.desc.n_media == 1
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    g_img, c_img, _ = compare_images(rt, gpu, O.OracleScene(s), params, camera, sampler)
    nan = np.isnan(g_img[..., :3]).any(-1)
    assert np.array_equal(nan, np.isnan(c_img[..., :3]).any(-1)) and 0.02 < nan.mean() < 0.98, nan.mean()


@pytest.mark.gpu
def test_moving_sphere_in_lights(rt, gpu):
    """MovingSphere::pdf_value tests the ray against the sphere at time 0 and ::random aims at center(time1) - center(time0), the
    difference of the two centres (objects/sphere.rs:204-226, sic).  Against the oracle, ray counts equal; time0 != 0 so that
    center(0) is an extrapolation."""
    s = rt.host.Scene()
    grey, red = s.lambertian(s.solid(0.7, 0.7, 0.7)), s.lambertian(s.solid(0.7, 0.2, 0.2))
    lamp, dark = s.diffuse_light(s.solid(12, 11, 9)), s.diffuse_light(s.solid(0, 0, 0))
    c0, c1, t0, t1, r = (1.0, 4.0, 0.5), (2.5, 4.5, -0.5), 0.25, 1.0, 0.6
    world = [s.xz_rect(-20, 20, -20, 20, 0, grey), s.sphere((-2, 1, 0), 1, red), s.block((1, 0, -1), (2.5, 1.5, 0.5), grey),
             s.moving_sphere(c0, c1, t0, t1, r, lamp)]
    lights = [s.moving_sphere(c0, c1, t0, t1, r, dark), s.xz_rect(-1, 1, -1, 1, 6, dark)]
    s.set_roots(s.hittable_list(world), s.hittable_list(lights))
    flat = s.flatten()
    assert flat.desc.n_lights == 2
    rec = np.ctypeslib.as_array(flat.desc.lights, (2 * 8,))[:8].copy()
    f = np.float32
    at = lambda t: np.array(c0, f) + ((f(t) - f(t0)) / (f(t1) - f(t0))) * (np.array(c1, f) - np.array(c0, f))
    assert rec[:1].view(np.int32)[0] == 1 and np.array_equal(rec[1:4], at(0.0)) and rec[4] == f(r)       # RT_PRIM_MOVING_SPHERE, center(0)
    assert np.array_equal(rec[5:8], at(t1) - at(t0))
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=96, aspect_ratio=1.5, samples_per_pixel=16, look_from=(0.0, 3.0, 11.0), look_at=(0.5, 1.5, 0.0),
        vertical_fov=40.0, background=(0.02, 0.02, 0.03))
    camera, sampler = params.camera(), rt.host.MultiJitteredSampler(16)
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    oracle = O.OracleScene(s)
    g_img, c_img, stats = compare_images(rt, gpu, oracle, params, camera, sampler)
    assert stats["light_pdf_evals"] > 0 and np.nanmax(g_img[..., :3]) > 1.0
    assert np.array_equal(np.isnan(g_img), np.isnan(c_img))


def cylinder_light_scene(rt):
    s = rt.host.Scene()
    grey, red = s.lambertian(s.solid(0.7, 0.7, 0.7)), s.lambertian(s.solid(0.7, 0.2, 0.2))
    lamp, dark = s.diffuse_light(s.solid(7, 6, 5)), s.diffuse_light(s.solid(0, 0, 0))
    two_pi = float(np.float32(np.pi) * np.float32(2.0))
    m = np.eye(4, dtype=np.float32); m[:3, 3] = (3.0, 1.0, -2.0)
    world = [s.xz_rect(-20, 20, -20, 20, -1.5, grey), s.sphere((-2.5, -0.5, 0), 1, red), s.block((1.5, -1.5, -1), (3.0, 0.0, 0.5), grey),
             s.cylinder(0.5, -1.0, 1.0, two_pi, lamp),                              # a tube light along z through the origin
             s.transform(s.cylinder(0.3, -0.5, 0.5, 4.0, lamp), m)]                 # a partial one, moved
    lights = [s.cylinder(0.5, -1.0, 1.0, two_pi, dark),
              s.transform(s.cylinder(0.3, -0.5, 0.5, 4.0, dark), m),                # forwarded untransformed (transform.rs:206-212)
              s.xz_rect(-1, 1, -1, 1, 6, dark)]
    s.set_roots(s.hittable_list(world), s.hittable_list(lights))
    return s


def test_cylinder_in_lights(rt, gpu):
    """Cylinder is the one quadric with its own pdf_value / random (cylinder.rs:134-161): the pdf from its hit() and area(),
    the direction like a sphere of its radius around the centre of its box.  Pinned by hand for one ray, then images and
    per-pixel ray counts against the oracle."""
    s = cylinder_light_scene(rt)
    flat = s.flatten()
    assert flat.desc.n_lights == 3
    rec = np.ctypeslib.as_array(flat.desc.lights, (3 * 8,)).copy().reshape(3, 8)
    two_pi = np.float32(np.pi) * np.float32(2.0)
    assert rec[0, :1].view(np.int32)[0] == 6                                        # RT_PRIM_CYLINDER
    assert np.array_equal(rec[0, 1:], np.array([-1.0, 1.0, two_pi, 0.5, 0.0, 0.0, 0.0], np.float32))
    assert np.array_equal(rec[1, 1:], np.array([-0.5, 0.5, 4.0, 0.3, 0.0, 0.0, 0.0], np.float32))
    oracle = O.OracleScene(s)
    # straight down from (0, 3, 0) onto the tube: p = (0, 0.5, 0), |n . -v| = 1, d^2 = 6.25, area = 2 * 0.5 * 2 pi; and onto the
    # partial one where it is WITHOUT its Transform: p = (0, 0.3, 0) at phi = pi / 2 < 4, d^2 = 7.29, area = 1 * 0.3 * 4
    got = oracle.lights_pdf_value(np.array([0, 3, 0], np.float32), np.array([0, -1, 0], np.float32))
    assert got == pytest.approx((6.25 / (2.0 * np.pi) + 7.29 / 1.2) / 3.0, rel=1e-5)
    assert oracle.lights_pdf_value(np.array([0, 3, 5], np.float32), np.array([0, -1, 0], np.float32)) == 0.0
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=96, aspect_ratio=1.5, samples_per_pixel=16, look_from=(0.0, 3.0, 11.0), look_at=(0.5, 0.0, 0.0),
        vertical_fov=40.0, background=(0.02, 0.02, 0.03))
    camera, sampler = params.camera(), rt.host.MultiJitteredSampler(16)
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    g_img, c_img, stats = compare_images(rt, gpu, oracle, params, camera, sampler)
    assert stats["light_pdf_evals"] > 0 and np.nanmax(g_img[..., :3]) > 1.0
    assert np.array_equal(np.isnan(g_img), np.isnan(c_img))


def nested_light_lists_scene(rt):
    s = rt.host.Scene()
    grey, red = s.lambertian(s.solid(0.7, 0.7, 0.7)), s.lambertian(s.solid(0.7, 0.2, 0.2))
    lamp, dark = s.diffuse_light(s.solid(7, 6, 5)), s.diffuse_light(s.solid(0, 0, 0))
    two_pi = float(np.float32(np.pi) * np.float32(2.0))
    world = [s.xz_rect(-20, 20, -20, 20, -1.5, grey), s.sphere((-2.5, -0.5, 0), 1, red), s.block((1.5, -1.5, -1), (3.0, 0.0, 0.5), grey),
             s.cylinder(0.5, -1.0, 1.0, two_pi, lamp), s.flip_face(s.xz_rect(-1, 1, -1, 1, 6, lamp)),
             s.sphere((4, 2, -1), 0.5, lamp), s.sphere((-4, 3, 1), 0.4, lamp)]
    inner = s.hittable_list([s.cylinder(0.5, -1.0, 1.0, two_pi, dark), s.xz_rect(-1, 1, -1, 1, 6, dark)])
    lights = [s.xz_rect(-1, 1, -1, 1, 6, dark),
              s.hittable_list([s.sphere((4, 2, -1), 0.5, dark), inner]),                       # a list in a list in `lights`
              s.translate(s.hittable_list([s.sphere((-4, 3, 1), 0.4, dark)]), (9, 9, 9))]       # a wrapped one: forwarded
    s.set_roots(s.hittable_list(world), s.hittable_list(lights))
    return s


def test_lists_inside_lights(rt, gpu):
    """HittableList::pdf_value is the mean of its members' pdfs and ::random draws a member (hittable_list.rs:59-75); a list
    that is itself a member does the same one level down, with a draw of its own.  Layout pinned, then images and per-pixel
    ray counts against the oracle's nested lists."""
    s = nested_light_lists_scene(rt)
    flat = s.flatten()
    n = flat.desc.n_lights
    rec = np.ctypeslib.as_array(flat.desc.lights, (n * 8,)).copy().reshape(n, 8)
    kinds = rec[:, 0].view(np.int32).tolist()
    runs = rec[:, 1:3].view(np.int32)
    # root | rect, list, list | sphere, list | sphere | cylinder, rect
    assert kinds == [253, 3, 254, 254, 0, 254, 0, 6, 3]
    assert runs[0].tolist() == [1, 3] and runs[2].tolist() == [4, 2] and runs[3].tolist() == [6, 1] and runs[5].tolist() == [7, 2]
    oracle = O.OracleScene(s)
    # from (0, 3, 0) straight up: only the two rectangles at y = 6 are hit (d^2 = 9, cosine 1, area 4): 9/4 as a member of
    # `lights` and as a member of the innermost list: (1/3) (9/4 + (1/2) (0 + (1/2) (0 + 9/4)) + 0)
    got = oracle.lights_pdf_value(np.array([0, 3, 0], np.float32), np.array([0, 1, 0], np.float32))
    assert got == pytest.approx((2.25 + 0.25 * 2.25) / 3.0, rel=1e-6)
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=96, aspect_ratio=1.5, samples_per_pixel=16, look_from=(0.0, 3.0, 11.0), look_at=(0.5, 0.0, 0.0),
        vertical_fov=40.0, background=(0.02, 0.02, 0.03))
    camera, sampler = params.camera(), rt.host.MultiJitteredSampler(16)
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    g_img, c_img, stats = compare_images(rt, gpu, oracle, params, camera, sampler)
    assert stats["light_pdf_evals"] > 0 and np.nanmax(g_img[..., :3]) > 1.0
    assert np.array_equal(np.isnan(g_img), np.isnan(c_img))


def test_checkers_of_every_shape(rt, gpu):
    """rt_scene_upload folds a checker over two solid colours into one device record; every other checker (a branch that is
    a checker, a noise texture or an image) keeps the two-level form, and one of those may point at a folded one.  Tex-eval
    counters, ray counts and images against the oracle (checker_texture.rs:37-45)."""
    s = rt.host.Scene()
    plain = s.checker(s.solid(0.2, 0.3, 0.1), s.solid(0.9, 0.85, 0.8), 10.0)                  # folded
    nested = s.checker(plain, s.solid(0.7, 0.2, 0.2), 1.5)                                    # a folded checker inside a checker
    twice = s.checker(nested, s.checker(s.solid(0.1, 0.1, 0.6), s.solid(0.6, 0.6, 0.1), 6.0), 0.7)
    marbled = s.checker(s.noise(4.0), s.solid(0.8, 0.8, 0.8), 3.0)                            # a branch that is Perlin noise
    rgba = np.zeros((8, 8, 4), np.uint8); rgba[..., 0] = np.arange(8)[None, :] * 30; rgba[..., 1] = np.arange(8)[:, None] * 30; rgba[..., 3] = 255
    pictured = s.checker(s.image(rgba), s.solid(0.3, 0.3, 0.3), 5.0)                          # a branch that reads u, v
    world = [s.xz_rect(-30, 30, -30, 30, 0, s.lambertian(twice)), s.sphere((-3.2, 1, 0), 1, s.lambertian(plain)),
             s.sphere((-1, 1, 0.5), 1, s.lambertian(nested)), s.sphere((1.2, 1, 0), 1, s.lambertian(marbled)),
             s.sphere((3.4, 1, 0.5), 1, s.lambertian(pictured)), s.sphere((0, 1, -3), 1, s.diffuse_light(plain)),
             s.flip_face(s.xz_rect(-2, 2, -2, 2, 6, s.diffuse_light(s.solid(6, 6, 6))))]
    s.set_roots(s.hittable_list(world), s.hittable_list([s.xz_rect(-2, 2, -2, 2, 6, s.diffuse_light(s.solid(0, 0, 0)))]))
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=128, aspect_ratio=1.6, samples_per_pixel=16, look_from=(0.0, 4.0, 11.0), look_at=(0.0, 1.0, 0.0),
        vertical_fov=40.0, background=(0.25, 0.35, 0.55))
    camera, sampler = params.camera(), rt.host.MultiJitteredSampler(16)
    gpu.upload_scene(s.flatten())
    gpu.upload_sampler(sampler)
    oracle = O.OracleScene(s)
    g_img, c_img, stats = compare_images(rt, gpu, oracle, params, camera, sampler)
    assert np.isfinite(g_img).all()
    ev = stats["tex_evals"]
    assert ev[1] > ev[0] * 0.5 and ev[2] > 0 and ev[3] > 0      # checkers (several levels per hit), noise and image branches were all taken


def wrapped_media_scene(rt):
    """constant_medium.rs:13-39 takes any boundary and is itself a Hittable that any wrapper takes: media under wrappers of their
    own, over boundaries under Transforms and under more than one record, in a lit room."""
    s = rt.host.Scene()
    white = s.lambertian(s.solid(0.73, 0.73, 0.73))
    red = s.lambertian(s.solid(0.65, 0.05, 0.05))
    green = s.lambertian(s.solid(0.12, 0.45, 0.15))
    lamp = s.diffuse_light(s.solid(7, 7, 7))
    light = s.xz_rect(113, 443, 127, 432, 554, lamp)
    objs = [s.yz_rect(0, 555, 0, 555, 555, green), s.yz_rect(0, 555, 0, 555, 0, red), s.flip_face(light),
            s.xz_rect(0, 555, 0, 555, 0, white), s.xz_rect(0, 555, 0, 555, 555, white), s.xy_rect(0, 555, 0, 555, 555, white)]
    block = lambda: s.block((0, 0, 0), (165, 165, 165), white)
    # a medium under RotateY(Translate(.)) over a bare block: T and R are the MEDIUM's wrappers
    objs.append(s.rotate_y(s.translate(s.constant_medium(block(), s.solid(0.1, 0.1, 0.1), 0.012), (60, 0, 330)), 12.0))
    # ... under a Translate, over a boundary that has a RotateY of its own and a Translate above that (two records)
    objs.append(s.translate(s.constant_medium(s.translate(s.rotate_y(block(), -20.0), (0, 10, 0)), s.solid(0.9, 0.9, 0.9), 0.015), (300, 0, 60)))
    # ... under a general Transform (a sheared sphere of smoke), and a plain medium over a block under a general Transform
    shear = np.eye(4, dtype=np.float32); shear[0, 1] = 0.35; shear[1, 1] = 1.3; shear[:3, 3] = (130, 260, 150)
    objs.append(s.transform(s.constant_medium(s.sphere((0, 0, 0), 70, white), s.solid(0.2, 0.4, 0.9), 0.02), shear))
    squash = np.diag([1.2, 0.5, 1.0, 1.0]).astype(np.float32); squash[:3, 3] = (330, 300, 300)
    objs.append(s.constant_medium(s.transform(s.rotate_y(block(), 30.0), squash), s.solid(0.9, 0.6, 0.2), 0.01))
    s.set_roots(s.hittable_list(objs), s.hittable_list([light]))
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(image_width=96, samples_per_pixel=16)
    return s, params


def test_media_under_wrappers_and_over_wrapped_boundaries(rt, gpu):
    """Where a medium's hit is made — in the frame of the medium's own wrappers (the ray's length, the hit point and its way out,
    constant_medium.rs:76-93 inside transform.rs' hit()s) — decides the bits of everything a path does afterwards: per-pixel ray
    counts against the oracle's nested objects, images to the bar of DESIGN.md section 5."""
    s, params = wrapped_media_scene(rt)
    flat = s.flatten()
    med = flat.media().view(np.int32)
    assert med.shape[0] == 4 and (med[:, 7] == 1).all() and (med[:3, 2] > 0).all() and med[3, 2] == 0
    camera = params.camera()
    sampler = rt.host.MultiJitteredSampler(16)
    gpu.upload_scene(flat)
    gpu.upload_sampler(sampler)
    _, _, st = compare_images(rt, gpu, O.OracleScene(s), params, camera, sampler)
    assert st["medium_tests"] > 0 and st["shades"][4] > 1000          # isotropic scatters happened
