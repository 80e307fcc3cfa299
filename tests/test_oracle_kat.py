"""Oracle vs the hand-derived known answers of tests/golden/kat.json (bit-exact), plus the
tie-break / wrapper quirks of the reference that have a closed-form expectation."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_ffi as O

rt = O.rt
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
INF = float("inf")


def f32(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def ray(o, d, time=0.0):
    return np.array([[o[0]], [o[1]], [o[2]], [d[0]], [d[1]], [d[2]], [time]], np.float32)


def tmax(c):
    return INF if c["t_max"] == "inf" else c["t_max"]


def scene_with(build):
    s = rt.host.Scene()
    white = s.lambertian(s.solid(0.5, 0.5, 0.5))
    objs = build(s, white)
    light = s.xz_rect(0, 1, 0, 1, 100, s.diffuse_light(s.solid(0, 0, 0)))
    s.set_roots(s.hittable_list(objs), s.hittable_list([light]))
    return s


def check_hit(scene, case, expect_uv=False):
    o = O.OracleScene(scene)
    pid, t, n, ff, uv = o.first_hit(ray(case["o"], case["d"]), case["t_min"], tmax(case))
    want = case["hit"]
    if want is None:
        assert pid[0] == 0xFFFFFFFF
        return
    assert pid[0] == 0
    assert t.view(np.uint32)[0] == want["t"]
    assert list(n.view(np.uint32)[0]) == want["normal"]
    assert ff[0] == want["front_face"]
    if expect_uv:
        assert list(uv.view(np.uint32)[0]) == want["uv"]


@pytest.mark.parametrize("case", KAT["sphere"])
def test_sphere(case):
    check_hit(scene_with(lambda s, m: [s.sphere(case["center"], case["radius"], m)]), case)


@pytest.mark.parametrize("case", KAT["rect"])
def test_rect(case):
    mk = {"xy": "xy_rect", "xz": "xz_rect", "yz": "yz_rect"}[case["kind"]]
    check_hit(scene_with(lambda s, m: [getattr(s, mk)(case["a0"], case["a1"], case["b0"], case["b1"], case["k"], m)]), case, True)


@pytest.mark.parametrize("case", KAT["aabb"])
def test_aabb(case):
    r = (C.c_float * 7)(*case["o"], *case["d"], 0.0)
    got = O.lib().orc_aabb_hit((C.c_float * 3)(*case["min"]), (C.c_float * 3)(*case["max"]), r, case["t_min"], tmax(case))
    assert got == case["hit"]


def test_onb_schlick_refract():
    for c in KAT["onb"]:
        out = (C.c_float * 9)()
        O.lib().orc_onb((C.c_float * 3)(*c["n"]), out)
        assert list(np.array(list(out), np.float32).view(np.uint32)) == c["uvw"]
    for c in KAT["schlick"]:
        v = np.float32(O.lib().orc_schlick(c["cosine"], c["ior"]))
        assert int(v.view(np.uint32)) == c["value"]
    for c in KAT["refract"]:
        out = (C.c_float * 3)()
        O.lib().orc_refract((C.c_float * 3)(*c["uv"]), (C.c_float * 3)(*c["n"]), c["eta"], out)
        got = np.array(list(out), np.float32).view(np.uint32)
        want = np.array(c["value"], np.uint32)
        nan = np.isnan(want.view(np.float32))
        assert np.array_equal(got[~nan], want[~nan]) and np.isnan(got.view(np.float32)[nan]).all()


def perlin_table():
    p = KAT["perlin"]
    tab = np.zeros(1536, np.uint32)
    tab[:768] = np.array(p["ranvec"], np.uint32).reshape(768)
    tab[768:1024], tab[1024:1280], tab[1280:] = p["perm_x"], p["perm_y"], p["perm_z"]
    return tab


def test_perlin_double_smoothing():
    tab = perlin_table()
    for c in KAT["perlin"]["noise"]:
        v = np.float32(O.lib().orc_perlin(tab.ctypes.data, (C.c_float * 3)(*c["p"]), 0))
        assert int(v.view(np.uint32)) == c["value"], c
    for c in KAT["perlin"]["turbulence"]:
        v = np.float32(O.lib().orc_perlin(tab.ctypes.data, (C.c_float * 3)(*c["p"]), 7))
        assert int(v.view(np.uint32)) == c["value"], c


def test_image_texel_addressing_and_blank():
    img = np.zeros((2, 4, 4), np.uint8)
    for j in range(2):
        for i in range(4):
            img[j, i] = (10 * i + j, 100 + i, 200 + j, 255)
    s = rt.host.Scene()
    tex = s.image(img)
    blank = s.image(None)
    m = s.lambertian(tex)
    s.set_roots(s.hittable_list([s.sphere((0, 0, 0), 1, m)]), s.hittable_list([s.sphere((0, 5, 0), 1, m)]))
    o = O.OracleScene(s)
    for c in KAT["image"]:
        i, j = c["ij"]
        got = o.texture_value(tex, c["u"], c["v"], (0, 0, 0))
        want = np.array([np.float32(1 / 255) * np.float32(x) for x in img[j, i, :3]], np.float32)
        assert np.array_equal(got, want), c
    assert list(o.texture_value(blank, 0.3, 0.3, (0, 0, 0))) == [0.0, 1.0, 1.0]   # image_texture.rs:57-59


def test_list_tie_later_object_wins():
    """hittable_list.rs:43-49: the comparator never returns Equal, so on equal t the LATER object is kept."""
    s = scene_with(lambda s, m: [s.xy_rect(0, 1, 0, 1, 0, m), s.xy_rect(0, 1, 0, 1, 0, m), s.xy_rect(0, 1, 0, 1, 2, m)])
    pid, t, *_ = O.OracleScene(s).first_hit(ray((0.5, 0.5, -1), (0, 0, 1)))
    assert pid[0] == 1 and t[0] == 1.0


def test_list_fold_keeps_what_comes_after_the_last_nan_candidate():
    """hittable_list.rs:40-50 is `min_by` with a comparator that says Less only if a.t < b.t: the held candidate is replaced unless
    its t is smaller — so a NaN candidate (a rectangle whose plane contains the ray: (k - o.z) / d.z = 0 / 0, and every range test
    of rectangles.rs:25-32 is false for NaN) replaces whatever was held, and is itself replaced by whatever comes next.  By hand,
    for the ray (0.5, 0.5, 0) + t (1, 0, 0): A = xy_rect in z = 0 (NaN), B = sphere at (3, 0.5, 0) r 0.5 (t = 2), C = yz_rect at
    x = 9 (t = 8.5)."""
    objs = {"A": lambda s, m: s.xy_rect(0, 1, 0, 1, 0, m), "B": lambda s, m: s.sphere((3, 0.5, 0), 0.5, m), "C": lambda s, m: s.yz_rect(-5, 5, -5, 5, 9, m)}
    want = {"ABC": ("B", 2.0),      # A; B replaces the NaN; C: 2 < 8.5, B stays
            "BCA": ("A", None),     # B; C: B stays; A: 2 < NaN is false -> A, with its NaN distance
            "CAB": ("B", 2.0),      # C; A replaces; B replaces
            "BAC": ("C", 8.5),      # B; A replaces; C replaces the NaN: the closer B is forgotten
            "ACB": ("B", 2.0),      # A; C replaces; B: 8.5 < 2 is false -> B
            "CBA": ("A", None)}
    for order, (winner, t_want) in want.items():
        sc = scene_with(lambda s, m: [objs[k](s, m) for k in order])
        pid, t, *_ = O.OracleScene(sc).first_hit(ray((0.5, 0.5, 0), (1, 0, 0)))
        assert pid[0] == order.index(winner), (order, pid[0])
        assert (np.isnan(t[0]) if t_want is None else t[0] == t_want), (order, t[0])


def test_a_block_folds_its_sides_before_the_list_sees_it():
    """block.rs:111-113: a Block's hit() is its own list's hit(), so the enclosing list sees ONE candidate.  The ray
    (-2, 1, 0.5) + t (1, 0, 0) runs along the top face (y = 1) of the block [2, 3] x [0, 1] x [0, 1]: sides in the order of
    block.rs:27-79 — XY@z1 and XY@z0 miss (-inf), XZ@y1 is 0 / 0 = NaN, XZ@y0 misses (-inf), YZ@x1 t = 5 replaces the NaN, YZ@x0
    t = 4 replaces that (5 < 4 is false): the Block answers t = 4.  The sphere at (0, 1, 0.5) r 0.5 is hit at t = 1.5 (half_b = -2,
    c = 3.75, root = 2 - sqrt(0.25)): in either order the list keeps the sphere.  (Folding the six sides into the enclosing list
    would let the NaN side wipe out the sphere: Block side at t = 4 — what the GPU kernels do)  Off to the side
    (z = 5: no YZ side is hit, but the NaN side ignores its extents) the Block answers with the NaN side itself, and the list,
    which holds the sphere at t = 1.5, takes it: 1.5 < NaN is false."""
    for order in ("SB", "BS"):
        build = lambda s, m: [s.sphere((0, 1, 0.5), 0.5, m) if k == "S" else s.block((2, 0, 0), (3, 1, 1), m) for k in order]
        pid, t, *_ = O.OracleScene(scene_with(build)).first_hit(ray((-2, 1, 0.5), (1, 0, 0)))
        sphere_id = 0 if order == "SB" else 6          # a Block's sides carry six consecutive ids
        assert pid[0] == sphere_id and t[0] == 1.5, (order, pid[0], t[0])
    build = lambda s, m: [s.sphere((0, 1, 5), 0.5, m), s.block((2, 0, 0), (3, 1, 1), m)]
    pid, t, *_ = O.OracleScene(scene_with(build)).first_hit(ray((-2, 1, 5), (1, 0, 0)))
    assert pid[0] == 3 and np.isnan(t[0]), (pid[0], t[0])        # sphere 0; XY@z1 1, XY@z0 2, XZ@y1 3
    build = lambda s, m: [s.block((2, 0, 0), (3, 1, 1), m), s.sphere((0, 1, 5), 0.5, m)]
    pid, t, *_ = O.OracleScene(scene_with(build)).first_hit(ray((-2, 1, 5), (1, 0, 0)))
    assert pid[0] == 6 and t[0] == 1.5, (pid[0], t[0])           # the sphere comes after the NaN and replaces it


def test_bvh_tie_right_child_wins():
    """bvh.rs:83: `if a.t < b.t {a} else {b}` -> right."""
    s = scene_with(lambda s, m: [s.bvh([s.xy_rect(0, 1, 0, 1, 0, m), s.xy_rect(0, 1, 0, 1, 0, m)])])
    got = set()
    for seed in range(8):   # either build order: the winner is whichever leaf ends up on the right
        pid, *_ = O.OracleScene(s, bvh_seed=seed).first_hit(ray((0.5, 0.5, -1), (0, 0, 1)))
        got.add(int(pid[0]))
    assert got <= {0, 1} and len(got) >= 1


def test_flip_face_and_translate_front_face_quirk():
    """flip_face.rs flips only the flag; Translate re-runs HitRecord::new on the already
    face-forwarded normal (transform.rs:36-44), which makes front_face true whatever it was."""
    def build(s, m):
        r = s.xz_rect(0, 1, 0, 1, 1, m)          # seen from below: outward n = +y, d = +y -> back face
        return [s.flip_face(r)]
    pid, t, n, ff, _ = O.OracleScene(scene_with(build)).first_hit(ray((0.5, 0, 0.5), (0, 1, 0)))
    assert ff[0] == 1 and list(n[0]) == [0, -1, 0]       # flag flipped, normal untouched
    def build2(s, m):
        return [s.translate(s.flip_face(s.xz_rect(0, 1, 0, 1, 1, m)), (10, 0, 0))]
    pid, t, n, ff, _ = O.OracleScene(scene_with(build2)).first_hit(ray((10.5, 0, 0.5), (0, 1, 0)))
    assert ff[0] == 1 and list(n[0]) == [0, -1, 0]       # Translate recomputed it: true, flip lost
    def build3(s, m):
        return [s.flip_face(s.translate(s.xz_rect(0, 1, 0, 1, 1, m), (10, 0, 0)))]
    pid, t, n, ff, _ = O.OracleScene(scene_with(build3)).first_hit(ray((10.5, 0, 0.5), (0, 1, 0)))
    assert ff[0] == 0                                     # true from Translate, then flipped


def test_rotate_y_block_hit_point():
    """transform.rs:116-144: 90 degrees about Y maps the block [0,1]^3 to x in [0,1], z in [-1,0]."""
    def build(s, m):
        return [s.rotate_y(s.block((0, 0, 0), (1, 1, 1), m), 90.0)]
    pid, t, n, ff, _ = O.OracleScene(scene_with(build)).first_hit(ray((0.5, 0.5, -3), (0, 0, 1)))
    assert pid[0] != 0xFFFFFFFF and abs(t[0] - 2.0) < 1e-5 and ff[0] == 1
    np.testing.assert_allclose(n[0], [0, 0, -1], atol=1e-6)


def test_portable_math_close_to_libm():
    """include/rt_portable_math.h vs glibc on the ranges the path uses: within a few ulp."""
    rng = np.random.default_rng(1)
    L = O.lib()
    worst = 0.0
    for fn, lo, hi in ((0, -700, 700), (1, -700, 700), (2, -1, 1), (4, 1e-7, 1.0)):
        for x in rng.uniform(lo, hi, 4000).astype(np.float32):
            a, b = np.float32(L.orc_math(fn, float(x), 0.0, 0)), np.float32(L.orc_math(fn, float(x), 0.0, 1))
            ulp = np.spacing(np.float32(max(abs(b), 1e-3 if fn < 2 else abs(b))))
            worst = max(worst, abs(float(a) - float(b)) / float(ulp))
    for y, x in rng.uniform(-5, 5, (4000, 2)).astype(np.float32):
        a, b = np.float32(L.orc_math(3, float(y), float(x), 0)), np.float32(L.orc_math(3, float(y), float(x), 1))
        worst = max(worst, abs(float(a) - float(b)) / float(np.spacing(np.float32(max(abs(b), 1e-3)))))
    assert worst <= 8.0, worst
    assert L.orc_math(4, 0.0, 0.0, 0) == -INF        # ln(0) = -inf (constant_medium.rs:78 relies on it)


def test_mesh_first_hit_vs_nearest_modes():
    """the reference returns the FIRST triangle in index order that hits, not the nearest."""
    verts = np.array([[0, 0, 5, 0, 0, -1, 0, 0], [1, 0, 5, 0, 0, -1, 1, 0], [0, 1, 5, 0, 0, -1, 0, 1],      # far triangle first
                      [0, 0, 2, 0, 0, -1, 0, 0], [1, 0, 2, 0, 0, -1, 1, 0], [0, 1, 2, 0, 0, -1, 0, 1]], np.float32)
    idx = np.array([0, 1, 2, 3, 4, 5], np.uint32)
    nodes = np.array([[0, 6]], np.uint32)
    def build(s, m):
        return [s.triangle_mesh(verts, idx, nodes, np.eye(4), m)]
    sc = scene_with(build)
    r = ray((0.2, 0.2, 0), (0, 0, 1))
    pid, t, *_ = O.OracleScene(sc, mesh_nearest=False).first_hit(r)
    assert pid[0] == 0 and t[0] == 5.0
    pid, t, *_ = O.OracleScene(sc, mesh_nearest=True).first_hit(r)
    assert pid[0] == 1 and t[0] == 2.0


def test_camera_projections_follow_camera_rs():
    """get_ray_perspective / _ortho / _fisheye / _spherical_panoramic (camera.rs:82-184), recomputed in float64."""
    import ctypes as C
    lib = O.lib()
    lib.orc_project.argtypes = [C.POINTER(O.orc_render_params), C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    lib.orc_project.restype = C.c_int
    scene, params, camera, sampler = rt.configs.build("C1", image_width=64, samples_per_pixel=4)
    cam = {k: np.array(list(getattr(camera, k)), np.float64) for k in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")}
    lens = float(camera.lens_radius)
    psi_max, lam_max = np.radians(120.0), np.radians(170.0)

    def expected(proj, u, v, dx, dy):
        if proj in (0, 1):
            off = cam["u"] * (lens * dx) + cam["v"] * (lens * dy)
            point = cam["lower_left_corner"] + u * cam["horizontal"] + v * cam["vertical"]
            return (cam["origin"] + off, point - cam["origin"] - off) if proj == 0 else (point + off, -cam["w"])
        px, py = 2 * u - 1, 2 * v - 1
        if proj == 2:
            r2 = px * px + py * py
            if r2 > 1:
                return None
            r = np.sqrt(r2)
            psi = r * psi_max
            return cam["origin"], np.sin(psi) * (px / r) * cam["u"] + np.sin(psi) * (py / r) * cam["v"] - np.cos(psi) * cam["w"]
        phi, theta = np.pi - px * lam_max, 0.5 * np.pi - py * psi_max
        return cam["origin"], np.sin(theta) * np.sin(phi) * cam["u"] + np.cos(theta) * cam["v"] + np.sin(theta) * np.cos(phi) * cam["w"]

    rng = np.random.default_rng(3)
    p = O.orc_render_params()
    p.camera = camera
    p.psi_max, p.lambda_max = psi_max, lam_max
    none_seen = 0
    for proj in range(4):
        p.projection = proj
        for _ in range(200):
            u, v, dx, dy = (float(np.float32(x)) for x in (rng.random(), rng.random(), rng.uniform(-0.7, 0.7), rng.uniform(-0.7, 0.7)))
            out = (C.c_float * 6)()
            ok = lib.orc_project(C.byref(p), u, v, dx, dy, out)
            want = expected(proj, u, v, dx, dy)
            if want is None:
                assert ok == 0
                none_seen += 1
                continue
            assert ok == 1
            np.testing.assert_allclose(np.array(out[:3]), want[0], rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(np.array(out[3:]), want[1], rtol=1e-5, atol=2e-6 * np.abs(want[1]).max())
    assert none_seen > 10          # FishEye corners lie outside the unit disk: Option::None (camera.rs:146)


def test_cancellation_false_positive():
    """A documented deviation class.  sphere.rs:46-66 in f32 accepts this ray for a r=0.2
    sphere it passes 0.66 away from: origin and direction are ~2000 long (a path segment INSIDE the r=1000 ground
    sphere), half_b^2 - a*c cancels, and the reported hit point is not on the sphere.  The reference's flat list
    takes the hit; any traversal that culls by bounding box (the GPU's) does not.  Found with an Orthographic view
    of RandomWorld whose view plane dips into the ground; C1-C4 as configured have no such rays."""
    scene, params, camera, sampler = rt.configs.build("C1", image_width=96, samples_per_pixel=4)
    ray = np.array([[-647.327393, -1432.85925, -627.37561, 909.112427, 1986.85925, 878.007812, 0.662694097]], np.float32).T.copy()
    pid, t, *_ = O.OracleScene(scene).first_hit(ray)
    slots = scene.flatten().slots()
    s = slots[slots[:, 6].view(np.int32) == int(pid[0])][0]
    centre, radius = s[0:3].astype(np.float64), float(s[3])
    assert radius == pytest.approx(0.2)
    p = ray[0:3, 0].astype(np.float64) + float(t[0]) * ray[3:6, 0].astype(np.float64)
    assert np.linalg.norm(p - centre) > 3 * radius          # the "hit point" is nowhere near the sphere


# ---- quadrics and the general Transform.  Every expectation below is worked out by hand
# from cylinder.rs / cone.rs / paraboloid.rs / hyperboloid.rs / transform.rs; the roots are exact in f32.

def one_hit(scene, o, d, t_min=0.001, t_max=INF):
    pid, t, n, ff, uv = O.OracleScene(scene).first_hit(ray(o, d), t_min, t_max)
    return int(pid[0]), float(t[0]), n[0].astype(np.float64), bool(ff[0]), uv[0].astype(np.float64)


def test_cylinder_known_answers():
    two_pi = float(np.float32(np.pi) * np.float32(2))
    sc = scene_with(lambda s, m: [s.cylinder(1, -0.5, 0.5, two_pi, m)])
    # a = 1, b = -6, c = 8: q = 4, roots 2 and 4.  The normal is cross(dpdv, dpdu) (cylinder.rs:121), which
    # points INWARDS, so a ray from outside sees front_face = false and the face-forwarded normal +x.
    pid, t, n, ff, uv = one_hit(sc, (3, 0, 0), (-1, 0, 0))
    assert (pid, t, ff) == (0, 2.0, False) and np.allclose(n, [1, 0, 0]) and np.allclose(uv, [0, 0.5])
    assert one_hit(sc, (3, 0, 0.6), (-1, 0, 0))[0] == 0xFFFFFFFF            # above zmax at both roots
    assert one_hit(sc, (0, 0, 3), (0, 0, -1))[0] == 0xFFFFFFFF              # along the axis: a = 0, b = 0 -> no roots
    # half cylinder (phi_max = pi): the near root has phi = 3 pi / 2 > phi_max, the far one phi = pi / 2
    half = scene_with(lambda s, m: [s.cylinder(1, -0.5, 0.5, float(np.float32(np.pi)), m)])
    pid, t, n, ff, uv = one_hit(half, (0, -3, 0), (0, 1, 0))
    assert (pid, t, ff) == (0, 4.0, True) and np.allclose(n, [0, -1, 0], atol=1e-6) and np.allclose(uv, [0.5, 0.5], atol=1e-6)
    assert one_hit(half, (0, -3, 0), (0, 1, 0), t_max=3.9)[0] == 0xFFFFFFFF  # cylinder.rs:95
    # t_min is honoured by the cylinder (:68-72): from inside, the root behind the origin is skipped
    pid, t, n, ff, uv = one_hit(sc, (0, 0, 0), (1, 0, 0))
    assert (pid, t, ff) == (0, 1.0, True) and np.allclose(n, [-1, 0, 0])


def test_cone_known_answers_and_ignored_t_min():
    two_pi = float(np.float32(np.pi) * np.float32(2))
    sc = scene_with(lambda s, m: [s.cone(1, 1, two_pi, m)])
    # k = 1, a = 1, b = -6, c = 9 - 0.25: delta = 1, q = 3.5, roots 2.5 and 3.5; p = (0, 0.5, 0.5), phi = pi/2
    # dpdu = (-pi, pi, 0) (sic: p.y twice, cone.rs:109), dpdv = (-0, -1, 1): normal = (1, 1, 1) / sqrt 3
    pid, t, n, ff, uv = one_hit(sc, (0, 3, 0.5), (0, -1, 0))
    assert (pid, t, ff) == (0, 2.5, True) and np.allclose(n, np.ones(3) / np.sqrt(3), atol=1e-6) and np.allclose(uv, [0.25, 0.5], atol=1e-6)
    # Cone::hit never looks at t_min (cone.rs:38, 61-65): the hit at t = 2.5 is returned for t_min = 3 too
    assert one_hit(sc, (0, 3, 0.5), (0, -1, 0), t_min=3.0)[1] == 2.5
    # with p.y = 0 the sic dpdu is the null vector and so is the normal (normalize of 0, vec3.rs:690-700)
    pid, t, n, ff, uv = one_hit(sc, (3, 0, 0.5), (-1, 0, 0))
    assert (pid, t) == (0, 2.5) and np.all(n == 0) and ff is False


def test_paraboloid_and_hyperboloid_known_answers():
    two_pi = float(np.float32(np.pi) * np.float32(2))
    # Paraboloid::new(0.5, -0.5, 1, 2 pi) of main.rs:932-938: z = 4 (x^2 + y^2)
    sc = scene_with(lambda s, m: [s.paraboloid(0.5, -0.5, 1.0, two_pi, m)])
    # a = 4, b = -24, c = 35.75: delta = 4, q = 13, roots 2.75 and 3.25; p = (0, 0.25, 0.25)
    pid, t, n, ff, uv = one_hit(sc, (0, 3, 0.25), (0, -1, 0))
    assert (pid, t, ff) == (0, 2.75, True)
    assert np.allclose(n, np.array([0, 0.75, -0.375]) / np.hypot(0.75, 0.375), atol=1e-6) and np.allclose(uv, [0.25, 0.5], atol=1e-6)
    # p1 = (1,0,-1), p2 = (1,0,1): ah = 1, ch = 0, the unit cylinder; the normal of a hyperboloid points outwards
    hy = scene_with(lambda s, m: [s.hyperboloid((1, 0, -1), (1, 0, 1), two_pi, m)])
    pid, t, n, ff, uv = one_hit(hy, (3, 0, 0), (-1, 0, 0))
    assert (pid, t, ff) == (0, 2.0, True) and np.allclose(n, [1, 0, 0]) and np.allclose(uv, [0, 0.5])
    # sic (hyperboloid.rs:116): a negative phi gets + 2, not + 2 pi, on the first attempt.  At p = (0,-1,0)
    # phi = -pi/2 + 2 = 0.429 <= phi_max = 1, so the near root is ACCEPTED although its true angle is 3 pi / 2
    hy1 = scene_with(lambda s, m: [s.hyperboloid((1, 0, -1), (1, 0, 1), 1.0, m)])
    pid, t, n, ff, uv = one_hit(hy1, (0, -3, 0), (0, 1, 0))
    assert (pid, t) == (0, 2.0) and np.isclose(uv[0], 2.0 - np.pi / 2, atol=1e-6)


def test_transform_known_answers():
    # Transform(translate(10,0,0) * scale(2), unit cube): object-space ray o = (5, .15, .1), d = (-.5, 0, 0) meets
    # the YZ side at x = 0.5 for t = 9 (t is kept, transform.rs:198-202); p goes back to world space
    m2w = np.array([[2, 0, 0, 10], [0, 2, 0, 0], [0, 0, 2, 0], [0, 0, 0, 1]], np.float32)
    sc = scene_with(lambda s, m: [s.transform(s.block((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), m), m2w)])
    pid, t, n, ff, uv = one_hit(sc, (20, 0.3, 0.2), (-1, 0, 0))
    assert (pid, t, ff) == (4, 9.0, True) and np.allclose(n, [1, 0, 0]) and np.allclose(uv, [0.65, 0.6], atol=1e-6)
    # non-uniform scale: the normal goes through adjoint(world2object)^T and is re-normalised (:171, 196-200).
    # adjoint(W) = det(W) W^-1, so that matrix is proportional to obj2world^T, NOT to the inverse transpose a
    # normal needs: reference behaviour, kept.  Sphere scaled by (1, 2, 1) at the point (s, 2s, 0), s = 1/sqrt 2:
    # the object normal (s, s, 0) becomes (1, 2, 0) / sqrt 5 (the ellipsoid's true normal there is (2, 1, 0) / sqrt 5)
    el = scene_with(lambda s, m: [s.transform(s.sphere((0, 0, 0), 1, m), np.diag([1.0, 2.0, 1.0, 1.0]))])
    s_ = 1 / np.sqrt(2)
    pid, t, n, ff, uv = one_hit(el, (s_ + 2, 2 * s_ + 1, 0), (-2, -1, 0))
    assert pid == 0 and np.isclose(t, 1.0, atol=1e-6) and np.allclose(n, np.array([1, 2, 0]) / np.sqrt(5), atol=1e-6)
    # FlipFace inside or outside a Transform flips the flag only
    for wrap in (lambda s, o: s.flip_face(s.transform(o, m2w)), lambda s, o: s.transform(s.flip_face(o), m2w)):
        fl = scene_with(lambda s, m: [wrap(s, s.block((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), m))])
        pid, t, n, ff, uv = one_hit(fl, (20, 0.3, 0.2), (-1, 0, 0))
        assert (pid, t, ff) == (4, 9.0, False) and np.allclose(n, [1, 0, 0])


def quadric_scene(case):
    """One quadric (and its Transform) as the only object of a scene: primitive id 0."""
    def build(s, m):
        kind, prm = case["kind"], case["params"]
        q = {"cylinder": lambda: s.cylinder(prm[0], prm[1], prm[2], prm[3], m), "cone": lambda: s.cone(prm[0], prm[1], prm[2], m),
             "paraboloid": lambda: s.paraboloid(prm[0], prm[1], prm[2], prm[3], m)}[kind]()
        return [q if case["matrix"] is None else s.transform(q, np.array(case["matrix"], np.float32))]
    return scene_with(build)


def check_quadric(first_hit, case):
    pid, t, n, ff, uv = first_hit(ray(case["o"], case["d"]), case["t_min"], tmax(case))
    want = case["hit"]
    if want is None:
        assert pid[0] == 0xFFFFFFFF, case
        return
    assert pid[0] == 0 and int(t.view(np.uint32)[0]) == want["t"], case
    assert list(n.view(np.uint32)[0]) == want["normal"] and ff[0] == want["front_face"], case
    assert int(uv.view(np.uint32)[0, 1]) == want["v"], case          # u = phi / phi_max depends on the atan2 in use: not pinned here


@pytest.mark.parametrize("case", KAT["quadric"], ids=lambda c: "%s%s" % (c["kind"], "" if c["matrix"] is None else "-xf"))
def test_quadrics_and_transform_against_the_numpy_restatement(case):
    """tests/golden/make_golden.py restates cylinder.rs / cone.rs / paraboloid.rs / polynomial.rs / transform.rs / mat4.rs
    independently (numpy f32, one operation at a time): t, normal, front_face and v must agree bit for bit."""
    check_quadric(O.OracleScene(quadric_scene(case)).first_hit, case)


# ---- round 2: the independent numpy restatement widened (tests/golden/make_golden.py "round 2") --------------------------
# Triangle test and cull, MovingSphere, Translate(RotateY(..)) records: through first_hit, here for the oracle and in
# tests/test_gpu_fullsize.py for the kernels.  One bounce of ray_color (metal, dielectric, Lambertian + mixture pdf + clamp +
# rectangle / sphere pdf_value, ConstantMedium::hit) with GIVEN random numbers: through orc_bounce, oracle only — the kernels are
# tied to the oracle by the bit-identical paths of the image tests (equal per-pixel ray counts, tests/test_gpu_parity.py).

def tri_scene(verts):
    v = np.array(verts, np.float32)
    return scene_with(lambda s, m: [s.triangle_mesh(v, np.arange(3, dtype=np.uint32), np.array([[0, 3]], np.uint32), None, m)])


def check_first_hit(first_hit, case, time=0.0, uv=False):
    pid, t, n, ff, tuv = first_hit(ray(case["o"], case["d"], time), case.get("t_min", 0.001), tmax(case) if "t_max" in case else INF)
    want = case["hit"]
    if want is None:
        assert pid[0] == 0xFFFFFFFF
        return
    assert pid[0] != 0xFFFFFFFF
    assert t.view(np.uint32)[0] == want["t"]
    assert list(n.view(np.uint32)[0]) == want["normal"]
    assert ff[0] == want["front_face"]
    if uv:
        assert list(tuv.view(np.uint32)[0]) == want["uv"]


@pytest.mark.parametrize("case", KAT["triangle"], ids=lambda c: "hit" if c["hit"] else "none")
def test_triangle_and_cull(case):
    """triangle_mesh.rs:185-268: barycentrics, interpolated normal, back-face cull, uv; the mesh and node boxes in front."""
    check_first_hit(O.OracleScene(tri_scene(case["verts"]), mesh_nearest=False).first_hit, case, uv=True)


def test_a_mesh_in_an_axis_plane_is_never_hit():
    """Its Aabb has zero thickness and Aabb::hit (aabb3.rs:19-45) needs tmax > tmin: the triangle alone would be hit."""
    c = KAT["triangle_flat_mesh"]
    assert c["hit"] is None and c["triangle_alone"] is not None
    check_first_hit(O.OracleScene(tri_scene(c["verts"]), mesh_nearest=False).first_hit, c)


def moving_scene(c):
    return scene_with(lambda s, m: [s.moving_sphere(c["c0"], c["c1"], c["time0"], c["time1"], c["radius"], m)])


@pytest.mark.parametrize("case", KAT["moving_sphere"], ids=lambda c: "t%g" % c["time"])
def test_moving_sphere(case):
    check_first_hit(O.OracleScene(moving_scene(case)).first_hit, case, time=case["time"])


def wrapped_scene(c):
    def build(s, m):
        i = c["inner"]
        if i[0] == "sphere":
            obj = s.sphere(i[1], i[2], m)
        else:
            obj = getattr(s, i[1] + "_rect")(i[2], i[3], i[4], i[5], i[6], m)
        return [s.translate(s.rotate_y_sincos(obj, c["sin"], c["cos"]), c["offset"])]
    return scene_with(build)


@pytest.mark.parametrize("case", KAT["wrapped"], ids=lambda c: c["inner"][0] + ("" if c["hit"] else "-miss"))
def test_translate_rotate_y_records(case):
    """transform.rs:28-45, 116-144: the ray goes in through Translate then RotateY, the record comes back through both."""
    check_first_hit(O.OracleScene(wrapped_scene(case)).first_hit, case)


def bounce_scene(c):
    s = rt.host.Scene()
    if c["material"] == "lambertian":          # the Cornell floor with the scene's two lights
        white = s.lambertian(s.solid(0.73, 0.73, 0.73))
        objs = [s.xz_rect(0, 555, 0, 555, 0, white)]
        lm = s.diffuse_light(s.solid(0, 0, 0))
        lights = [s.xz_rect(213, 343, 227, 332, 554, lm), s.sphere((190, 90, 190), 90, lm)]
    else:
        m = s.metal(*c["albedo"], c["fuzz"]) if c["material"] == "metal" else s.dielectric(c["ior"])
        objs = [s.sphere((0, 0, 0), 1.0, m)]
        lights = [s.xz_rect(0, 1, 0, 1, 100, s.diffuse_light(s.solid(0, 0, 0)))]
    s.set_roots(s.hittable_list(objs), s.hittable_list(lights))
    return s


@pytest.mark.parametrize("case", KAT["bounce"], ids=lambda c: c["material"])
def test_one_bounce_with_given_random_numbers(case):
    """metal.rs:20-37, dielectric.rs:16-68, lambertian.rs:32-52 + camera.rs:268-300 + pdf.rs + rectangles.rs:54-77 + sphere.rs:91-105."""
    b = O.OracleScene(bounce_scene(case)).bounce(case["o"], case["d"], case["xi"])
    e = case["expect"]
    assert b["t"].view(np.uint32) == case["t"]
    assert b["kind"] == e["kind"] and b["used"] == e["used"]
    if e["kind"] >= 2:
        assert list(b["dir"].view(np.uint32)) == e["dir"]
    if e["kind"] == 2:
        assert list(b["attenuation"].view(np.uint32)) == e["attenuation"]
    if e["kind"] == 3:
        assert b["pdf"].view(np.uint32) == e["pdf"] and b["scattering_pdf"].view(np.uint32) == e["scattering_pdf"]


def test_mixture_pdf_clamp_is_reached():
    """camera.rs:279-288: a grazing cosine sample away from both lights has a pdf below 1e-5 and gets +1e-4."""
    vals = [f32(c["expect"]["pdf"]) for c in KAT["bounce"] if c["material"] == "lambertian"]
    assert any(v == np.float32(1.0e-4) for v in vals) and any(v > np.float32(1.0e-3) for v in vals)


@pytest.mark.parametrize("case", KAT["medium"], ids=lambda c: "xi%g" % c["xi"])
def test_constant_medium_hit_with_a_given_number(case):
    """constant_medium.rs:46-94 over a sphere boundary; ln is this backend's (rt_portable_math.h), restated in numpy too."""
    s = rt.host.Scene()
    fog = s.constant_medium(s.sphere(case["center"], case["radius"], s.lambertian(s.solid(1, 1, 1))), s.solid(1, 1, 1), case["density"])
    s.set_roots(s.hittable_list([fog]), s.hittable_list([s.xz_rect(0, 1, 0, 1, 100, s.diffuse_light(s.solid(0, 0, 0)))]))
    b = O.OracleScene(s).bounce(case["o"], case["d"], [case["xi"], 0.5, 0.5, 0.5, 0.5, 0.5, 0.5])
    if case["t"] is None:
        assert b["kind"] == 0
    else:
        assert b["kind"] != 0 and b["t"].view(np.uint32) == case["t"]
        assert list(b["normal"]) == [1.0, 0.0, 0.0] and b["front_face"] == 1        # constant_medium.rs:86-88


def test_reference_known_answers_for_mat4_and_the_oracles_matrix_code():
    """math/src/mat4.rs:657-667 (transpose) against the oracle's own mat4_transpose, and its det / adjoint / invert
    (mat4.rs:536-617, index typos included) against the numpy restatement, bit for bit."""
    k = KAT["mat4_transpose"]
    assert list(O.mat4(0, k["m"])) == k["result"]
    for c in KAT["mat4_invert"]:
        assert list(O.mat4(1, c["m"]).view(np.uint32)) == c["inverse"]
