"""Host logic (librtiow_host.so): config, params, camera, sampler, scene builders, GLB import, flatten/BVH."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def test_config_ron_matches_the_shipped_file(rt):
    cfg = rt.host.RaytracerConfig.load()
    assert cfg.active_scene == "Blocks"                          # raytracer.config.ron:2
    p = cfg.scene_params("RandomWorld")                          # :25-44
    assert (p.workers, p.image_width, p.samples_per_pixel, p.max_ray_depth) == (12, 800, 144, 50)
    assert p.look_from == [13.0, 2.0, 3.0] and p.aperture == pytest.approx(0.1) and p.projection_name == "Perspective"
    assert p.background == pytest.approx([0.5, 0.8, 0.99])
    d = cfg.scene_params("CornellBox")                           # (CornellBox, None) -> default_params :3-21
    assert d.look_from == [278.0, 278.0, -800.0] and d.samples_per_pixel == 256 and d.image_width == 800
    assert d.image_height == int(np.float32(800) / np.float32(1.77777))   # main.rs:1180
    g = cfg.scene_params("GeometricPrimitives")
    assert g.projection_name == "SphericalPanoramic" and g.psi_max == 360.0
    assert cfg.scene_params().worker_block_pixels == 8           # active scene
    with pytest.raises(rt.RtError):
        cfg.scene_params("TexturedSpheres")                      # not in defined_scenes -> "Specified scene not found"


def test_config_parser_errors_and_syntax(rt):
    text = open(rt.host.DEFAULT_CONFIG).read()
    cfg = rt.host.RaytracerConfig.parse("// c\n" + text.replace("active_scene: Blocks", "active_scene: /* x */ MeshTest"))
    assert cfg.active_scene == "MeshTest"
    for bad in ("(", "(active_scene: Blocks)", text.replace("workers: 8,", "", 1), text.replace("Perspective", "Cubist")):
        with pytest.raises(rt.RtError):
            rt.host.RaytracerConfig.parse(bad)
    with pytest.raises(rt.RtError):
        rt.host.RaytracerConfig.load("/nonexistent/raytracer.config.ron")


def test_work_blocks_follow_main_rs(rt):
    b = rt.host.work_blocks(1200, 675, 8)                        # 151 x 85 = 12 835 blocks
    assert b.shape == (151 * 85, 4)
    area = ((b[:, 1] - b[:, 0]) * (b[:, 3] - b[:, 2])).sum()
    assert area == 1200 * 675
    assert (b[:, 1] == b[:, 0]).sum() == 85 + 0                  # 1200/8 is exact: the extra column is empty
    assert b[:, 1].max() == 1200 and b[:, 3].max() == 675


def test_camera_new_known_answer(rt):
    cam = rt.host.Camera.new((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 2.0, 2.0, 1.0)
    np.testing.assert_allclose(list(cam.w), [0, 0, 1], atol=0)
    np.testing.assert_allclose(list(cam.u), [1, 0, 0], atol=0)
    np.testing.assert_allclose(list(cam.v), [0, 1, 0], atol=0)
    np.testing.assert_allclose(list(cam.lower_left_corner), [-2, -1, -1], rtol=1e-6)   # half_h = tan(45deg) = 1, half_w = 2
    np.testing.assert_allclose(list(cam.horizontal), [4, 0, 0], rtol=1e-6)
    np.testing.assert_allclose(list(cam.vertical), [0, 2, 0], rtol=1e-6)
    assert cam.lens_radius == 1.0 and (cam.time0, cam.time1) == (0.0, 1.0)


@pytest.mark.parametrize("spp", [9, 16, 10, 1000, 1])
def test_sampler_table_properties(rt, spp):
    s = rt.host.MultiJitteredSampler(spp, seed=3)
    sq, disk, idx = s.arrays()
    assert s.desc.sets == 83 and sq.shape == (83 * spp, 2)
    for p in range(83):
        blk = sq[p * spp:(p + 1) * spp]
        assert (blk >= 0).all() and (blk < 1.0 + 1e-6).all()
        # n-rooks: exactly one point per sub-cell column and per row, also after the coordinate shuffles
        for c in range(2):
            x = np.sort(blk[:, c].astype(np.float64)) * spp      # k-th smallest lies in sub-cell k (f32 rounding slack)
            k = np.arange(spp)
            assert (x >= k - 1e-3).all() and (x <= k + 1 + 1e-3).all(), (p, c)
        assert np.array_equal(np.sort(idx[p * spp:(p + 1) * spp]), np.arange(spp))
    assert (np.hypot(disk[:, 0], disk[:, 1]) <= 1.0 + 1e-5).all()
    s2 = rt.host.MultiJitteredSampler(spp, seed=3)
    assert all(np.array_equal(a, b) for a, b in zip(s.arrays(), s2.arrays()))
    if spp > 1:
        s3 = rt.host.MultiJitteredSampler(spp, seed=4)
        assert not np.array_equal(s.arrays()[0], s3.arrays()[0])


def test_multijittered_square_spp_keeps_the_reference_grid(rt):
    """sampling.rs:242-245 before the shuffles: x in sub-cell i*n+j, y in sub-cell j*n+i.  After the
    x/y shuffles only the multiset of sub-cells survives, which the n-rooks test covers; here: spp=1 and 4."""
    s = rt.host.MultiJitteredSampler(4, sets=2, seed=1)
    sq, _, _ = s.arrays()
    assert sq.shape == (8, 2)


def test_mat4_reference_known_answer(rt):
    k = KAT["mat4_mul"]                                           # math/src/mat4.rs:641-655
    out = (C.c_float * 16)()
    rt._ffi.host_lib().rth_mat4_mul((C.c_float * 16)(*k["a"]), (C.c_float * 16)(*k["b"]), out)
    assert list(out) == k["result"]
    m = np.array([[2, 0, 0, 1], [0, 4, 0, 2], [0, 0, 8, 3], [0, 0, 0, 1]], np.float32)
    rt._ffi.host_lib().rth_mat4_invert((C.c_float * 16)(*m.reshape(16)), out)
    np.testing.assert_allclose(np.array(list(out)).reshape(4, 4) @ m, np.eye(4), atol=1e-6)
    assert rt._ffi.host_lib().rth_mat4_invert((C.c_float * 16)(*([0.0] * 16)), out) < 0


def graph_counts(scene):
    g = scene.graph().contents
    types = [g.nodes[i].type for i in range(g.n_nodes)]
    return g, types


def test_reference_known_answers_for_transpose_and_colours(rt):
    """The reference's own tests next to the path: math/src/mat4.rs:657-667 and math/src/colors.rs:589-629."""
    L = rt._ffi.host_lib()
    k = KAT["mat4_transpose"]
    out = (C.c_float * 16)()
    L.rth_mat4_transpose((C.c_float * 16)(*k["m"]), out)
    assert list(out) == k["result"]
    L.rth_rgba8_from_html.argtypes = [C.c_char_p, C.c_uint8 * 4]
    L.rth_rgba8_to_u32.argtypes = [C.c_uint8] * 4; L.rth_rgba8_to_u32.restype = C.c_uint32
    L.rth_color_from_u32.argtypes = [C.c_uint32, C.c_float * 4]
    for text, want in KAT["colors"]["html"]:
        got = (C.c_uint8 * 4)()
        L.rth_rgba8_from_html(text.encode(), got)
        assert list(got) == want, text
    for rgba, want in KAT["colors"]["to_u32"]:
        assert L.rth_rgba8_to_u32(*rgba) == want
    f = (C.c_float * 4)()
    L.rth_color_from_u32(0xff34eb9e, f)                             # colors.rs:34-41
    assert list(f) == [np.float32(52) / np.float32(255), np.float32(235) / np.float32(255), np.float32(158) / np.float32(255), 1.0]
    for bad in (b"", b"   ", b"12345", b"#1234567", b"+12345"):
        got = (C.c_uint8 * 4)(9, 9, 9, 9)
        L.rth_rgba8_from_html(bad, got)
        assert list(got) == [0, 0, 0, 0]


def test_scene_builders_object_counts(rt):
    s = rt.host.Scene.build("CornellBox")
    g, types = graph_counts(s)
    world = g.nodes[g.world_root]
    assert world.child_count == 8                                 # main.rs:387-502
    assert g.nodes[g.lights_root].child_count == 2
    f = s.flatten()
    assert f.desc.n_prims == 13 and f.desc.always_count == 13 and f.desc.n_lights == 2 and f.desc.n_xforms == 1

    s = rt.host.Scene.build("RandomWorld", seed=1)
    g, types = graph_counts(s)
    n_world = g.nodes[g.world_root].child_count
    assert 400 < n_world <= 488                                   # 1 + 22*22 (minus rejected) + 3
    moving = types.count(1)
    assert 0.7 < moving / (n_world - 4) < 0.9                     # ~80 % moving Lambertian, main.rs:131
    f = s.flatten()
    assert f.desc.n_prims == n_world and f.desc.n_moving_slots == moving and f.desc.n_lights == 1
    s2 = rt.host.Scene.build("RandomWorld", seed=1)
    assert np.array_equal(s2.flatten().slots(), f.slots())        # seeded: reproducible
    assert not np.array_equal(rt.host.Scene.build("RandomWorld", seed=2).flatten().slots()[:50], f.slots()[:50])

    assets = rt.host.Assets(need=("earthmap",))
    s = rt.host.Scene.build("Chapter2Final", assets=assets)
    g, _ = graph_counts(s)
    assert g.nodes[g.world_root].child_count == 11                # main.rs:715-814
    f = s.flatten()
    assert f.desc.n_prims == 400 * 6 + 1 + 5 + 2 + 1000 + 1       # SURVEY: 3 409 leaf prims (2 media among them)
    assert f.desc.n_slots == f.desc.n_prims - 2 and f.desc.n_media == 2 and f.desc.n_perlin == 1
    assert f.desc.n_texel_bytes == 1024 * 512 * 4
    with pytest.raises(rt.RtError):
        rt.host.Scene.build("NoSuchScene")
    for name, n_world in (("TwoSpheres", 2), ("PerlinSpheres", 4), ("SimpleLight", 4), ("TexturedSpheres", 1)):   # main.rs:201-360
        scene, params, camera, sampler = rt.configs.build({"TwoSpheres": "TWO", "PerlinSpheres": "PERLIN", "SimpleLight": "LIGHT",
                                                           "TexturedSpheres": "EARTH"}[name], samples_per_pixel=4)
        g, _ = graph_counts(scene)
        assert g.nodes[g.world_root].child_count == n_world and g.nodes[g.lights_root].child_count == 1
        assert scene.flatten().desc.n_prims == n_world

    s = rt.host.Scene.build("Blocks", assets=rt.host.Assets(need=("uvgrid01",)))
    g, types = graph_counts(s)
    assert g.nodes[g.world_root].child_count == 5 and types.count(13) == 3      # main.rs:1049-1120: floor, light, 3 Transform(Block)
    f = s.flatten()
    assert f.desc.n_prims == 2 + 3 * 6 and f.desc.n_matrices == 3 and f.desc.always_count == f.desc.n_slots
    m = np.ctypeslib.as_array(f.desc.matrices, (3, 3, 3, 4)).astype(np.float64)  # [transform][w2o, o2w, n2w][row][col]
    for w2o, o2w, n2w in m:
        full = lambda a: np.vstack([a, [0, 0, 0, 1]])
        assert np.allclose(full(w2o) @ full(o2w), np.eye(4), atol=1e-5)          # transform.rs:166
        assert np.allclose(n2w[:, :3], np.linalg.inv(w2o[:, :3]).T * np.linalg.det(w2o[:, :3]), rtol=1e-4, atol=1e-7)  # adjoint^T (:171)
    # block1 = translate(0,5,0) * scale(5,10,5), main.rs:1093-1095
    assert np.allclose(m[1, 1], [[5, 0, 0, 0], [0, 10, 0, 5], [0, 0, 5, 0]])

    s = rt.host.Scene.build("GeometricPrimitives", assets=rt.host.Assets(need=("uvgrid01", "uvgrid03", "uvgrid08", "uvgrid10")))
    g, types = graph_counts(s)
    assert g.nodes[g.world_root].child_count == 10                # main.rs:848-954
    assert [types.count(t) for t in (13, 14, 15, 16, 5)] == [8, 2, 2, 2, 2]
    f = s.flatten()
    assert f.desc.n_prims == 2 + 2 * 6 + 6 and f.desc.n_matrices == 8
    types4 = f.slots()[:, 4].view(np.int32) & 15
    assert sorted(types4[types4 >= 6].tolist()) == [6, 6, 7, 7, 8, 8]


def check_bvh(flat, expect_boxes=False):
    nodes, slots, d = flat.nodes(), flat.slots(), flat.desc
    seen = np.zeros(d.n_slots, np.int32)
    max_depth = [0]

    def leaf_slots(ref):
        code = ~ref & 0xFFFFFFFF
        return range(code >> 3, (code >> 3) + (code & 7) + 1)

    def slot_box(s):
        """World box of a sphere / moving-sphere slot without transform (enough for RandomWorld); None otherwise."""
        w4 = int(slots[s, 4].view(np.int32))
        kind, xf = w4 & 15, (w4 >> 8) - 1
        if xf >= 0 or kind > 1:
            return None
        c0, r = slots[s, 0:3].astype(np.float64), float(slots[s, 3])
        lo, hi = c0 - r, c0 + r
        if kind == 1:
            c1 = slots[s, 8:11].astype(np.float64)
            lo, hi = np.minimum(lo, c1 - r), np.maximum(hi, c1 + r)
        return lo, hi

    def child_box(n, c):
        """The quantised child box of include/rt_api.h, decoded in exact arithmetic."""
        origin, scale = n[0:3].astype(np.float64), n[3:6].astype(np.float64)
        q = n[6:12].view(np.uint32)
        lo = np.array([origin[a] + scale[a] * ((int(q[2 * a]) >> (8 * c)) & 255) for a in range(3)])
        hi = np.array([origin[a] + scale[a] * ((int(q[2 * a + 1]) >> (8 * c)) & 255) for a in range(3)])
        return lo, hi

    def walk(ref, depth):
        """Returns the exact box of everything below `ref` (None if it holds a primitive this test cannot box)."""
        max_depth[0] = max(max_depth[0], depth)
        if ref < 0:
            boxes = []
            for s in leaf_slots(ref):
                seen[s] += 1
                boxes.append(slot_box(s))
            if any(b is None for b in boxes):
                return None
            return np.min([b[0] for b in boxes], 0), np.max([b[1] for b in boxes], 0)
        n = nodes[ref]
        refs = n[12:16].view(np.int32)
        assert (refs != 0x7FFFFFFF).sum() >= 2                     # a node has 2..4 children
        below = []
        for c in range(4):
            if refs[c] != 0x7FFFFFFF:                              # RT_BVH_EMPTY: no child
                sub = walk(int(refs[c]), depth + 1)
                below.append(sub)
                if sub is not None:                                # conservative: the 8-bit box contains what it culls for
                    lo, hi = child_box(n, c)
                    assert (lo <= sub[0]).all() and (hi >= sub[1]).all(), "quantised child box cuts into its subtree"
                    checked[0] += 1
        if any(b is None for b in below):
            return None
        return np.min([b[0] for b in below], 0), np.max([b[1] for b in below], 0)
    checked = [0]
    walk(d.bvh_root_ref, 0)
    if expect_boxes:
        assert checked[0] > 100
    seen[d.always_begin:d.always_begin + d.always_count] += 1     # the always-list is tested for every ray instead
    assert (seen == 1).all(), "every slot must sit in exactly one leaf or in the always-list"
    assert 3 * max_depth[0] + 1 <= d.bvh_stack_depth             # a visit pushes at most three children
    ids = slots[:, 6].view(np.int32)
    assert len(set(ids.tolist())) == d.n_slots


def test_flatten_bvh_invariants(rt):
    f = rt.host.Scene.build("RandomWorld").flatten()
    check_bvh(f, expect_boxes=True)
    f = rt.host.Scene.build("Chapter2Final", assets=rt.host.Assets(need=("earthmap",))).flatten()
    check_bvh(f)
    slots = f.slots()
    w4 = slots[:, 4].view(np.int32)
    assert ((w4 >> 8) > 0).sum() == 1000                          # the 1000 spheres carry the Translate(RotateY) record
    assert ((w4 & 16) != 0).sum() == 1                            # one FlipFace'd light


def test_quadrics_stay_out_of_the_bvh(rt):
    # cone / paraboloid / hyperboloid accept hits below t_min (cone.rs:61-65): they are tested for every ray;
    # slots under a general Transform honour t_min and go through the BVH like any other
    s = rt.host.Scene()
    m = s.lambertian(s.solid(1, 1, 1))
    objs = [s.sphere((3.0 * i, 0, 0), 1, m) for i in range(40)]
    objs.append(s.cone(1, 1, 6.0, m))
    objs.append(s.hyperboloid((1, 0, -1), (1, 0, 1), 6.0, m))
    objs.append(s.transform(s.flip_face(s.sphere((0, 0, 0), 1, m)), np.diag([2.0, 2.0, 2.0, 1.0])))
    s.set_roots(s.hittable_list(objs), s.hittable_list([s.sphere((0, 9, 0), 1, m)]))
    f = s.flatten()
    d = f.desc
    assert d.n_slots == 43 and d.always_begin == 41 and d.always_count == 2 and d.n_bvh_nodes > 0
    check_bvh(f)
    sl = f.slots()
    assert (sl[41:, 4].view(np.int32) & 15).tolist() == [7, 9]
    assert sl[41:, 6].view(np.int32).tolist() == [40, 41]          # primitive ids follow the world list
    ball = sl[sl[:, 6].view(np.int32) == 42][0]                    # the transformed sphere sits in a BVH leaf
    assert (ball[4:5].view(np.int32)[0] & 16) != 0                 # FlipFace inside a Transform survives (transform.rs:198-202)
    assert (ball[4:5].view(np.int32)[0] >> 8) > 0
    assert np.allclose(sl[42, 12:14], [1.0, 0.0])                  # x^2 + y^2 - 0 z^2 = 1: ah = 1, ch = 0


def test_any_wrapper_nesting_flattens_to_chain_records(rt):
    """Translate(RotateY(.)) and a lone wrapper stay one xform record; every other nesting of Translate / RotateY /
    Transform (transform.rs:14-213: each hit() hands its ray to its child) becomes a chain record over a run of them."""
    s = rt.host.Scene()
    m = s.lambertian(s.solid(1, 1, 1))
    ball = lambda: s.sphere((0, 0, 0), 1, m)
    objs = [s.translate(s.rotate_y(ball(), 10.0), (1, 0, 0)),                       # one record, flags 3
            s.rotate_y(s.translate(ball(), (1, 0, 0)), 10.0),                       # chain of 2: R, T
            s.transform(s.translate(ball(), (1, 0, 0)), np.eye(4)),                 # chain of 2: M, T
            s.translate(s.translate(s.rotate_y(ball(), 10.0), (1, 0, 0)), (0, 2, 0)),   # chain of 2: T, T+R
            s.rotate_y(s.translate(ball(), (1, 0, 0)), 10.0)]                       # the same chain again: shared
    s.set_roots(s.hittable_list(objs), s.hittable_list([s.sphere((0, 9, 0), 1, m)]))
    f = s.flatten()
    x = f.xforms()
    flags = x[:, 0].view(np.int32)
    sl = f.slots()
    by_id = {int(i): int(w >> 8) - 1 for i, w in zip(sl[:, 6].view(np.int32), sl[:, 4].view(np.int32))}
    assert flags[by_id[0]] == 3
    for k, kinds in ((1, [1, 2]), (2, [4, 2]), (3, [2, 3])):
        rec = x[by_id[k]].view(np.int32)
        assert rec[0] == 8 and rec[2] == 2
        assert flags[rec[1]:rec[1] + 2].tolist() == kinds
    assert by_id[4] == by_id[1]
    assert np.allclose(x[x[by_id[3]].view(np.int32)[1], 3:6], [0, 2, 0])            # outermost first
    # the world box follows the whole chain: R(T(ball)) sits at the rotated offset
    deep = ball()
    for _ in range(9):
        deep = s.translate(deep, (1, 0, 0))
    s.set_roots(s.hittable_list([deep]), s.hittable_list([s.sphere((0, 9, 0), 1, m)]))
    x9 = s.flatten().xforms()                                                       # nine wrappers: one chain of nine records
    assert (x9[:, 0].view(np.int32) == 8).sum() == 1 and x9[x9[:, 0].view(np.int32) == 8][0].view(np.int32)[2] == 9
    for _ in range(60):
        deep = s.translate(deep, (1, 0, 0))
    s.set_roots(s.hittable_list([deep]), s.hittable_list([s.sphere((0, 9, 0), 1, m)]))
    with pytest.raises(rt.RtError) as e:                                            # at most 64 levels of nesting
        s.flatten()
    assert e.value.status == -5
    s = rt.host.Scene()
    m = s.lambertian(s.solid(1, 1, 1))
    with pytest.raises(rt.RtError):                                # not affine
        s.transform(s.sphere((0, 0, 0), 1, m), np.arange(16.0))
    with pytest.raises(rt.RtError):                                # singular: the reference asserts (mat4.rs:610)
        s.transform(s.sphere((0, 0, 0), 1, m), np.diag([1.0, 0.0, 1.0, 1.0]))
    smoke = s.constant_medium(s.translate(s.translate(s.block((0, 0, 0), (1, 1, 1), m), (1, 0, 0)), (1, 0, 0)), s.solid(1, 1, 1), 0.1)
    s.set_roots(s.hittable_list([smoke]), s.hittable_list([s.sphere((0, 9, 0), 1, m)]))
    f2 = s.flatten()                                               # a boundary under two Translates: six boundary slots under a chain
    med = f2.media()
    assert med.shape[0] == 1 and med[0].view(np.int32)[7] == 1 and med[0].view(np.int32)[1] == 6 and med[0].view(np.int32)[2] == 0
    wrapped = s.rotate_y(s.constant_medium(s.translate(s.sphere((0, 1, 0), 1, m), (2, 0, 0)), s.solid(1, 1, 1), 0.2), 30.0)
    s.set_roots(s.hittable_list([wrapped]), s.hittable_list([s.sphere((0, 9, 0), 1, m)]))
    f3 = s.flatten()                                               # a medium under a wrapper of its own: its record names that wrapper
    med, x = f3.media(), f3.xforms()
    outer = med[0].view(np.int32)[2] - 1
    assert med[0].view(np.int32)[7] == 1 and med[0].view(np.int32)[1] == 1 and outer >= 0 and x[outer].view(np.int32)[0] == 1   # RotateY alone
    quad = s.constant_medium(s.cylinder(1, -1, 1, 6.0, m), s.solid(1, 1, 1), 0.2)
    s.set_roots(s.hittable_list([quad]), s.hittable_list([s.sphere((0, 9, 0), 1, m)]))
    with pytest.raises(rt.RtError):                                # a quadric (or a mesh, a list, another medium) as the boundary
        s.flatten()


def test_constructors_that_take_what_the_reference_objects_hold(rt):
    """integration/flatten.rs hands over RotateY's stored sin / cos (transform.rs:57-62) and TriangleMesh's baked vertices
    (triangle_mesh.rs:53-72): both forms flatten to the same slots and tree as the constructor-argument forms."""
    verts = np.array([[0, 0, 5, 0, 0, -2, 0, 0], [1, 0, 5, 0, 0, -2, 1, 0], [0, 1, 5, 0, 0, -2, 0, 1]], np.float32)
    o2w = np.eye(4, dtype=np.float32); o2w[:3, 3] = (3, 4, 5); o2w[0, 0] = 2
    tri, node = np.arange(3, dtype=np.uint32), np.array([[0, 3]], np.uint32)

    def build(sincos=None, baked=None):
        s = rt.host.Scene()
        m = s.lambertian(s.solid(1, 1, 1))
        blk = s.block((0, 0, 0), (1, 2, 1), m)
        rot = s.rotate_y_sincos(blk, *sincos) if sincos else s.rotate_y(blk, 15.0)
        mesh = s.triangle_mesh(baked, tri, node, None, m) if baked is not None else s.triangle_mesh(verts, tri, node, o2w, m)
        s.set_roots(s.hittable_list([s.translate(rot, (1, 0, 2)), mesh]), s.hittable_list([s.sphere((0, 9, 0), 1, m)]))
        return s

    first = build()
    flat = first.flatten()
    g = first.graph().contents
    baked = np.ctypeslib.as_array(C.cast(g.meshes[0].vertices, C.POINTER(C.c_float)), (3, 8)).copy()
    assert baked[0, 0] == 3.0 and baked[1, 0] == 5.0 and abs(np.linalg.norm(baked[0, 3:6]) - 1) < 1e-6      # baked + normalised
    xf = np.ctypeslib.as_array(flat.desc.xforms, (flat.desc.n_xforms, 8)).copy()
    sin_t, cos_t = float(xf[0, 1]), float(xf[0, 2])
    assert abs(sin_t - np.sin(np.deg2rad(15.0))) < 1e-6 and abs(cos_t - np.cos(np.deg2rad(15.0))) < 1e-6
    second = build((sin_t, cos_t), baked)
    other = second.flatten()
    assert np.array_equal(other.slots().view(np.uint32), flat.slots().view(np.uint32))
    assert np.array_equal(other.nodes().view(np.uint32), flat.nodes().view(np.uint32))


def test_constant_medium_over_a_block_boundary(rt):
    """scene_cornell_box_smoke (main.rs:524-683): the boundaries' 2 x 6 sides become slots behind the always-list that only the
    media records refer to; they take no primitive id; what the flattener cannot express is refused."""
    s = rt.host.Scene.build("CornellBoxSmoke")
    flat = s.flatten()                                                    # keeps the arrays behind desc alive
    d = flat.desc
    assert (d.n_slots, d.always_begin, d.always_count, d.n_media, d.n_prims, d.n_rect_slots) == (18, 0, 6, 2, 8, 6)
    med = np.ctypeslib.as_array(d.media, (2, 8)).copy()
    assert med.view(np.int32)[:, [0, 1, 6, 7]].tolist() == [[6, 6, 6, 1], [12, 6, 7, 1]] and list(med[:, 4]) == [-100.0, -100.0]
    slots = np.ctypeslib.as_array(d.slots, (18, 16)).copy()
    assert (slots[6:, 6].view(np.int32) == -1).all() and ((slots[6:, 4].view(np.int32) >> 8) > 0).all()     # no id, all under a transform
    with pytest.raises(rt.RtError) as e:                                  # the device build re-orders the head of the slot array only
        s.flatten(device_bvh=True)
    assert e.value.status == -5
    x = rt.host.Scene()
    m = x.lambertian(x.solid(1, 1, 1))
    fog = x.constant_medium(x.cylinder(1.0, 0.0, 1.0, 6.2831855, m), x.solid(1, 1, 1), 0.1)
    x.set_roots(x.hittable_list([fog]), x.hittable_list([x.sphere((0, 9, 0), 1, m)]))
    with pytest.raises(rt.RtError):
        x.flatten()


def read_glb_positions(path):
    d = open(path, "rb").read()
    jlen = struct.unpack("<I", d[12:16])[0]
    j = json.loads(d[20:20 + jlen])
    off = 20 + ((jlen + 3) & ~3) + 8
    out = []
    for mesh in j["meshes"]:
        acc = j["accessors"][mesh["primitives"][0]["attributes"]["POSITION"]]
        bv = j["bufferViews"][acc["bufferView"]]
        out.append(np.frombuffer(d, np.float32, acc["count"] * 3, off + bv.get("byteOffset", 0)).reshape(-1, 3))
    return out


def test_glb_import_counts_and_quaternion_quirk(rt):
    path = os.path.join(rt.host.DATA_DIR, "models/teapot/teapot.glb")
    geo = rt.host.Geometry(path)
    verts, idx, nodes = geo.view()
    assert verts.shape[0] == 10070 + 3763 and idx.shape[0] == 52392 + 15546
    assert nodes.tolist() == [[0, 52392], [52392, 15546]]
    assert idx[52392:].min() == 10070                                                  # indices rebased to the global vertex array
    raw = np.concatenate(read_glb_positions(path))
    # identity TRS read as quaternion (w,x,y,z) = (0,0,0,1) -> diag(-1,-1,1)
    assert np.array_equal(verts[:, 0], -raw[:, 0]) and np.array_equal(verts[:, 1], -raw[:, 1]) and np.array_equal(verts[:, 2], raw[:, 2])
    np.testing.assert_allclose(np.linalg.norm(verts[:, 3:6], axis=1), 1.0, atol=1e-5)
    with pytest.raises(rt.RtError):
        rt.host.Geometry("/nonexistent.glb")


def glb_json_and_bin(path):
    d = open(path, "rb").read()
    jlen = struct.unpack("<I", d[12:16])[0]
    return json.loads(d[20:20 + jlen]), d[20 + ((jlen + 3) & ~3) + 8:]


def test_glb_pbr_side_of_the_teapot(rt):
    # geometry_import.rs:172-304 + 385-425 against the file's own JSON, read here with the json module
    from PIL import Image
    import io
    path = os.path.join(rt.host.DATA_DIR, "models/teapot/teapot.glb")
    j, blob = glb_json_and_bin(path)
    geo = rt.host.Geometry(path)
    mats = geo.pbr_materials()
    assert len(mats) == len(j["materials"]) == 1
    m = mats[0]
    assert m["base_color_factor"].tolist() == [1.0, 1.0, 1.0] and m["metallic_factor"] == 0.0 and m["roughness_factor"] == 1.0   # glTF defaults
    assert (m["base_color_texarray_id"], m["metallic_rough_texarray_id"], m["normal_texarray_id"]) == (0, 0, 0)                   # one layer per list
    assert geo.material_index("utahTeapot") == 0 and geo.material_index("nope") == -1
    attrs = geo.vertex_attrs()
    verts, _, _ = geo.view()
    assert len(attrs) == len(verts) and np.all(attrs["pbr_buf_id"] == 0)
    assert np.all(attrs["color"] == np.array([0, 0, 0, 1], np.float32))           # no COLOR_0 in the file: GeometryVertex::default
    want = []
    for mesh in j["meshes"]:
        for prim in mesh["primitives"]:
            acc = j["accessors"][prim["attributes"]["TANGENT"]]
            bv = j["bufferViews"][acc["bufferView"]]
            assert acc["componentType"] == 5126 and acc["type"] == "VEC4" and "byteStride" not in bv
            want.append(np.frombuffer(blob, np.float32, acc["count"] * 4, bv.get("byteOffset", 0) + acc.get("byteOffset", 0)).reshape(-1, 4))
    assert np.array_equal(attrs["tangent"], np.concatenate(want))                  # copied verbatim (not transformed, :415-419)
    # the three lists: base colour = image of texture 1, metallic-roughness = texture 2, normal = texture 0
    mat = j["materials"][0]
    src = lambda t: j["textures"][t["index"]]["source"]
    for kind, tex in enumerate((mat["pbrMetallicRoughness"]["baseColorTexture"], mat["pbrMetallicRoughness"]["metallicRoughnessTexture"], mat["normalTexture"])):
        w, h, layers = geo.pbr_images(kind)
        bv = j["bufferViews"][j["images"][src(tex)]["bufferView"]]
        png = Image.open(io.BytesIO(blob[bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]])).convert("RGBA")
        assert len(layers) == 1 and np.array_equal(layers[0], np.asarray(png))
        size_of = mat["pbrMetallicRoughness"]["metallicRoughnessTexture"] if kind == 2 else tex     # sic, geometry_import.rs:135
        bv2 = j["bufferViews"][j["images"][src(size_of)]["bufferView"]]
        assert (w, h) == Image.open(io.BytesIO(blob[bv2.get("byteOffset", 0):bv2.get("byteOffset", 0) + bv2["byteLength"]])).size


def make_glb(doc, blob):
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    blob += b"\0" * (-len(blob) % 4)
    body = struct.pack("<II", len(js), 0x4e4f534a) + js + struct.pack("<II", len(blob), 0x004e4942) + blob
    return struct.pack("<III", 0x46546c67, 2, 12 + len(body)) + body


def test_glb_pbr_texture_layers_follow_sort_and_dedup(rt, tmp_path):
    # three materials that share images out of order: a layer id is the image's rank among the DISTINCT images of its kind
    # (geometry_import.rs:225-297); primitives find their material by name (:363-375); COLOR_0 as u8 RGB -> rgba f32 (:421-425)
    from PIL import Image
    import io
    blob, views, images = b"", [], []
    def add(data, **kw):
        nonlocal blob
        blob += b"\0" * (-len(blob) % 4)
        views.append(dict(buffer=0, byteOffset=len(blob), byteLength=len(data), **kw))
        blob += data
        return len(views) - 1
    pixels = []
    for k in range(5):                                   # five 2x1 PNGs, RGB and RGBA mixed
        im = Image.new("RGBA" if k % 2 else "RGB", (2 + k, 1), (10 * k, 20, 30) + ((40,) if k % 2 else ()))
        b = io.BytesIO(); im.save(b, "PNG")
        images.append(dict(bufferView=add(b.getvalue()), mimeType="image/png"))
        pixels.append(np.asarray(im.convert("RGBA")))
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    col = np.array([[255, 0, 51], [0, 255, 0], [0, 0, 255]], np.uint8)
    acc = [dict(bufferView=add(pos.tobytes()), componentType=5126, count=3, type="VEC3"),
           dict(bufferView=add(np.array([0, 1, 2], np.uint16).tobytes()), componentType=5123, count=3, type="SCALAR"),
           dict(bufferView=add(np.pad(col, ((0, 0), (0, 1))).tobytes(), byteStride=4), componentType=5121, count=3, type="VEC3", normalized=True)]
    tex = [dict(source=k) for k in range(5)]
    mk = lambda name, base, mr, nrm, **pbr: dict(name=name, normalTexture=dict(index=nrm),
                                                 pbrMetallicRoughness=dict(baseColorTexture=dict(index=base), metallicRoughnessTexture=dict(index=mr), **pbr))
    mats = [mk("c", 4, 3, 0, baseColorFactor=[0.5, 0.25, 0.125, 0.75], roughnessFactor=0.5), mk("a", 1, 3, 2), mk("b", 4, 0, 2, metallicFactor=0.25)]
    prim = lambda m: dict(attributes=dict(POSITION=0, COLOR_0=2), indices=1, material=m)
    doc = dict(asset=dict(version="2.0"), scenes=[dict(nodes=[0])], nodes=[dict(mesh=0)], meshes=[dict(primitives=[prim(2), prim(0), prim(1)])],
               materials=mats, textures=tex, images=images, accessors=acc, bufferViews=views, buffers=[dict(byteLength=len(blob))])
    path = tmp_path / "three_materials.glb"
    path.write_bytes(make_glb(doc, blob))
    geo = rt.host.Geometry(str(path))
    m = geo.pbr_materials()
    # base colour images {4, 1} -> layers 1:0, 4:1; metallic-roughness {3, 0} -> 0:0, 3:1; normal {0, 2} -> 0:0, 2:1
    assert m["base_color_texarray_id"].tolist() == [1, 0, 1]
    assert m["metallic_rough_texarray_id"].tolist() == [1, 1, 0]
    assert m["normal_texarray_id"].tolist() == [0, 1, 1]
    assert m["base_color_factor"].tolist() == [[0.5, 0.25, 0.125], [1, 1, 1], [1, 1, 1]]
    assert m["metallic_factor"].tolist() == [1.0, 1.0, 0.25] and m["roughness_factor"].tolist() == [0.5, 1.0, 1.0]
    assert [geo.material_index(n) for n in "abc"] == [1, 2, 0]
    a = geo.vertex_attrs()
    assert a["pbr_buf_id"].tolist() == [2] * 3 + [0] * 3 + [1] * 3
    assert np.array_equal(a["color"][:3], np.concatenate([col.astype(np.float32) / np.float32(255), np.ones((3, 1), np.float32)], axis=1))
    assert np.all(a["tangent"] == 0)
    w, h, layers = geo.pbr_images(0)
    assert (w, h) == (3, 1) and len(layers) == 2 and np.array_equal(layers[0], pixels[1]) and np.array_equal(layers[1], pixels[4])
    w, h, layers = geo.pbr_images(2)
    assert (w, h) == (2, 1) and np.array_equal(layers[1], pixels[2])     # sic: the size is the first metallic-roughness layer's (image 0)
    # the reference's panics are errors here: a material without a name, without one of the three textures, a duplicate name
    for bad in (lambda d: d["materials"][1].pop("name"), lambda d: d["materials"][0].pop("normalTexture"),
                lambda d: d["materials"][2].update(name="a")):
        d2 = json.loads(json.dumps(doc)); bad(d2)
        path.write_bytes(make_glb(d2, blob))
        with pytest.raises(rt.RtError):
            rt.host.Geometry(str(path))


def test_mesh_scene_teapot_stands_on_the_floor(rt):
    assets = rt.host.Assets(need=("uvgrid01", "uvgrid09", "teapot"))
    s = rt.host.Scene.build("MeshTest", assets=assets)
    g = s.graph().contents
    assert g.n_meshes == 1
    m = g.meshes[0]
    v = np.ctypeslib.as_array(C.cast(m.vertices, rt._ffi.c_float_p), (m.n_vertices, 8))
    raw = np.concatenate(read_glb_positions(os.path.join(rt.host.DATA_DIR, "models/teapot/teapot.glb")))
    np.testing.assert_allclose(v[:, :3], raw * np.array([-1, 1, -1], np.float32), atol=2e-6)   # net (x,y,z) -> (-x,y,-z)
    assert v[:, 1].min() > -1e-5 and 12.9 < v[:, 1].max() < 13.0
    f = s.flatten()
    assert f.desc.n_prims == 22646 + 2
    check_bvh(f)


def test_sample_range_sharding(rt):
    sr = rt.sharding.sample_range
    for spp, world in ((1000, 8), (10, 3), (7, 8), (16384, 8)):
        ranges = [sr(r, world, spp) for r in range(world)]
        pos = 0
        for b, c in ranges:
            assert b == pos and c >= 0
            pos += c
        assert pos == spp and max(c for _, c in ranges) - min(c for _, c in ranges) <= 1


def test_tile_sharding_covers_every_pixel_once(rt):
    import numpy as np
    for (w, h, block, world) in ((1200, 675, 32, 8), (600, 600, 8, 3), (37, 19, 16, 2), (64, 64, 64, 8)):
        cover = np.zeros((h, w), np.int32)
        counts = []
        for r in range(world):
            rects = rt.sharding.tile_rects(r, world, w, h, block)
            counts.append(len(rects))
            for x0, y0, x1, y1 in rects:
                assert 0 <= x0 < x1 <= w and 0 <= y0 < y1 <= h
                cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all()
        assert max(counts) - min(counts) <= 1


def test_c5_workload_is_the_4k_frame_split_over_eight_ranks(rt):
    # BASELINE configs[4]: 3840x2160 Book-2 final scene, 16k spp (128^2) sharded over 8 GPUs by sample range
    name, params, grid = rt.configs.workload("C5")
    assert (name, params.image_width, params.image_height, params.samples_per_pixel) == ("Chapter2Final", 3840, 2160, 16384)
    ranges = [rt.sharding.sample_range(r, 8, params.samples_per_pixel) for r in range(8)]
    assert ranges == [(2048 * r, 2048) for r in range(8)]


def test_lights_records_for_cylinders_and_nested_lists(rt):
    """`lights` entries beyond spheres and rectangles (rt_api.h "light"): a Cylinder (cylinder.rs:134-161) and lists inside the
    list (hittable_list.rs:59-75), laid out behind a root record run by run; the oracle's nested lists give the hand-computed pdf."""
    import oracle_ffi as O
    s = rt.host.Scene()
    dark = s.diffuse_light(s.solid(0, 0, 0))
    two_pi = float(np.float32(np.pi) * np.float32(2.0))
    world = s.hittable_list([s.sphere((0, -100, 0), 1, dark)])
    s.set_roots(world, s.hittable_list([s.cylinder(0.5, 1.0, -1.0, two_pi, dark), s.xz_rect(-1, 1, -1, 1, 6, dark)]))
    f = s.flatten()
    rec = np.ctypeslib.as_array(f.desc.lights, (2 * 8,)).copy().reshape(2, 8)
    assert rec[:, 0].view(np.int32).tolist() == [6, 3]                               # no root record without nested lists
    assert np.array_equal(rec[0, 1:], np.array([-1.0, 1.0, two_pi, 0.5, 0.0, 0.0, 0.0], np.float32))   # zmin / zmax as Cylinder::new sorts them
    o = O.OracleScene(s)
    got = o.lights_pdf_value(np.array([0, 3, 0], np.float32), np.array([0, -1, 0], np.float32))
    assert got == pytest.approx(0.5 * 6.25 / (2.0 * np.pi), rel=1e-6)                # p = (0, 0.5, 0), d^2 = 6.25, area = 2 * 0.5 * 2 pi
    inner = s.hittable_list([s.cylinder(0.5, -1.0, 1.0, two_pi, dark), s.xz_rect(-1, 1, -1, 1, 6, dark)])
    s.set_roots(world, s.hittable_list([s.xz_rect(-1, 1, -1, 1, 6, dark), s.hittable_list([s.sphere((4, 2, -1), 0.5, dark), inner]),
                                        s.translate(s.hittable_list([s.sphere((-4, 3, 1), 0.4, dark)]), (9, 9, 9))]))
    f = s.flatten()
    n = f.desc.n_lights
    rec = np.ctypeslib.as_array(f.desc.lights, (n * 8,)).copy().reshape(n, 8)
    assert rec[:, 0].view(np.int32).tolist() == [253, 3, 254, 254, 0, 254, 0, 6, 3]  # root | rect, list, list | sphere, list | sphere | cylinder, rect
    runs = rec[:, 1:3].view(np.int32)
    assert [runs[k].tolist() for k in (0, 2, 3, 5)] == [[1, 3], [4, 2], [6, 1], [7, 2]]
    got = O.OracleScene(s).lights_pdf_value(np.array([0, 3, 0], np.float32), np.array([0, 1, 0], np.float32))
    assert got == pytest.approx((2.25 + 0.25 * 2.25) / 3.0, rel=1e-6)                # the rectangle at y = 6, as a member of two lists
    s.set_roots(world, s.hittable_list([s.hittable_list([])]))
    with pytest.raises(rt.RtError):                                                  # random() would index objects[random_int(0, -1)]
        s.flatten()
    deep = s.sphere((0, 5, 0), 1, dark)
    for _ in range(5):
        deep = s.hittable_list([deep])
    s.set_roots(world, deep)
    with pytest.raises(rt.RtError):                                                  # at most 4 lists deep
        s.flatten()
