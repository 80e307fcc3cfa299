"""Error behaviour of the C-ABI (include/rt_api.h) on a live context.

The reference panics on a bad scene (an empty `lights` indexes objects[random_int(0,-1)], hittable_list.rs:71-75;
a non-square spp, sampling.rs:235-251) — a library cannot, so every entry point checks what its kernels will
index BEFORE anything is launched and returns a status with a message in rt_last_error.  What is tested here:
each check answers with the documented code, launches nothing, leaves the context usable and the scene that was
uploaded before in place (the next render gives the same bits), and the caller's buffers untouched.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RT_E_INVALID, RT_E_UNSUPPORTED_PRIMITIVE, RT_E_CANCELLED = -1, -5, -7


@pytest.fixture(scope="module")
def ctx(rt):
    """A context of its own (the session's one keeps its scene), with C1 (BVH + always-list) uploaded."""
    b = rt.backend.HipBackend(0)
    scene, params, camera, sampler = rt.configs.build("C1", image_width=64, samples_per_pixel=4)
    flat = scene.flatten()
    yield b, flat, params, camera, sampler
    b.close()


def copy_of(struct):
    return type(struct).from_buffer_copy(struct)


def status_and_text(b, st):
    return st, b._L.rt_last_error(b._h).decode()


def test_render_needs_a_scene_and_a_sampler(rt, ctx):
    b, flat, params, camera, sampler = ctx
    b._sampler = sampler
    with pytest.raises(rt._ffi.RtError) as e:
        b.render(params, camera)
    assert e.value.status == RT_E_INVALID and "upload a scene" in str(e.value)
    info = rt._ffi.rt_scene_info()
    info.struct_size = C.sizeof(info)
    assert b._L.rt_scene_query(b._h, C.byref(info)) == RT_E_INVALID
    b.upload_scene(flat)
    with pytest.raises(rt._ffi.RtError):          # scene but no sampler yet
        b.render(params, camera)
    b.upload_sampler(sampler)
    img, _ = b.render(params, camera)
    assert np.all(img[..., 3] == sampler.spp) and np.isfinite(img).all()


def test_bad_scene_descriptors_are_refused_and_the_old_scene_stays(rt, ctx):
    b, flat, params, camera, sampler = ctx
    b.upload_scene(flat)
    b.upload_sampler(sampler)
    ref, _ = b.render(params, camera)
    n = flat.desc.n_slots
    assert flat.desc.n_bvh_nodes > 0 and n > 0
    keep = []                                      # arrays the corrupted descriptors point into

    def with_slot_word(slot, word, value):
        d = copy_of(flat.desc)
        a = np.ctypeslib.as_array(flat.desc.slots, shape=(n * 16,)).copy()
        a.view(np.int32)[slot * 16 + word] = value
        keep.append(a)
        d.slots = a.ctypes.data_as(rt._ffi.c_float_p)
        return d

    def field(**kv):
        d = copy_of(flat.desc)
        for k, v in kv.items():
            setattr(d, k, v)
        return d

    cases = [
        (field(struct_size=flat.desc.struct_size - 4), RT_E_INVALID, "descriptor size"),
        (field(n_lights=0), RT_E_INVALID, "lights"),
        (field(always_count=n + 1), RT_E_INVALID, "always-list"),
        (field(bvh_root_ref=int(flat.desc.n_bvh_nodes) + 5), RT_E_INVALID, "bvh_root_ref"),
        (field(bvh_stack_depth=0), RT_E_INVALID, "bvh_stack_depth"),
        (field(n_media=5), RT_E_UNSUPPORTED_PRIMITIVE, "ConstantMedium"),
        (field(slots=None), RT_E_INVALID, "slots missing"),
        (field(lights=None), RT_E_INVALID, "record array"),
        (field(materials=None), RT_E_INVALID, "record array"),
        (field(textures=None), RT_E_INVALID, "record array"),
        (with_slot_word(n - 1, 5, int(flat.desc.n_materials)), RT_E_INVALID, "material out of range"),
        (with_slot_word(0, 5, -1), RT_E_INVALID, "material out of range"),
        (with_slot_word(n // 2, 4, 15), RT_E_UNSUPPORTED_PRIMITIVE, "primitive type"),
    ]
    for d, want, text in cases:
        st, msg = status_and_text(b, b._L.rt_scene_upload(b._h, C.byref(d)))
        assert st == want and text in msg, (st, msg, text)
    assert b._L.rt_scene_upload(b._h, None) == RT_E_INVALID
    again, _ = b.render(params, camera)            # nothing was freed or replaced by the refused uploads
    assert np.array_equal(again.view(np.uint32), ref.view(np.uint32))


def test_a_node_with_a_child_out_of_range_is_found_on_the_host(rt, ctx):
    b, flat, params, camera, sampler = ctx
    d = copy_of(flat.desc)
    nodes = np.ctypeslib.as_array(flat.desc.bvh_nodes, shape=(int(d.n_bvh_nodes) * 16,)).copy()
    nodes.view(np.int32)[(int(d.n_bvh_nodes) - 1) * 16 + 12] = int(d.n_bvh_nodes)      # inner-node ref one past the end
    d.bvh_nodes = nodes.ctypes.data_as(rt._ffi.c_float_p)
    st, msg = status_and_text(b, b._L.rt_scene_upload(b._h, C.byref(d)))
    assert st == RT_E_INVALID and "child out of range" in msg
    nodes.view(np.int32)[12] = ~np.int32((int(d.n_slots) << 3) | 1)                      # leaf run starting past the last slot
    st, msg = status_and_text(b, b._L.rt_scene_upload(b._h, C.byref(d)))
    assert st == RT_E_INVALID and "child out of range" in msg


def test_a_cyclic_tree_or_an_under_declared_stack_is_found_on_the_host(rt, ctx):
    """The walk pushes unchecked and its persistent waves end only when every stack has drained: a cycle would hang the GPU,
    a stack_depth below the tree's need would write past the LDS stack.  rt_scene_upload walks the tree and refuses both."""
    b, flat, params, camera, sampler = ctx
    b.upload_scene(flat)
    b.upload_sampler(sampler)
    ref, _ = b.render(params, camera)
    d = copy_of(flat.desc)
    nn = int(d.n_bvh_nodes)
    nodes = np.ctypeslib.as_array(flat.desc.bvh_nodes, shape=(nn * 16,)).copy()
    refs = nodes.view(np.int32).reshape(nn, 16)[:, 12:16]
    inner = [(i, c) for i in range(nn) for c in range(4) if 0 <= refs[i, c] < nn]
    assert inner, "C1's tree has more than one level"
    i, c = inner[-1]
    saved = refs[i, c]
    refs[i, c] = int(d.bvh_root_ref)                                            # a back edge to the root: a cycle
    d.bvh_nodes = nodes.ctypes.data_as(rt._ffi.c_float_p)
    st, msg = status_and_text(b, b._L.rt_scene_upload(b._h, C.byref(d)))
    assert st == RT_E_INVALID and "reachable twice" in msg, (st, msg)
    refs[i, c] = saved
    d.bvh_stack_depth = 3                                                       # one inner level's worth: C1 has several
    st, msg = status_and_text(b, b._L.rt_scene_upload(b._h, C.byref(d)))
    assert st == RT_E_INVALID and "too small" in msg, (st, msg)
    d.bvh_stack_depth = flat.desc.bvh_stack_depth                               # the intact copy is accepted
    assert b._L.rt_scene_upload(b._h, C.byref(d)) == 0
    again, _ = b.render(params, camera)
    assert np.array_equal(again.view(np.uint32), ref.view(np.uint32))


def test_bad_chain_and_light_list_records_are_refused(rt, ctx):
    """Chain records (xform flags 8) and the list records of `lights` (253 / 254) are indices the kernels follow: a run outside
    its array, a chain inside a chain, a list that points at or before itself (a cycle) or a record that no list owns are all
    found by rt_scene_upload."""
    b, flat0, params, camera, sampler = ctx
    s = rt.host.Scene()
    m, dark = s.lambertian(s.solid(0.5, 0.5, 0.5)), s.diffuse_light(s.solid(0, 0, 0))
    world = [s.rotate_y(s.translate(s.sphere((0, 0, 0), 1, m), (1, 0, 0)), 30.0), s.xz_rect(-5, 5, -5, 5, -1, m)]
    lights = [s.xz_rect(-1, 1, -1, 1, 6, dark), s.hittable_list([s.sphere((4, 2, -1), 0.5, dark), s.sphere((0, 3, 0), 0.5, dark)])]
    s.set_roots(s.hittable_list(world), s.hittable_list(lights))
    flat = s.flatten()
    d0 = flat.desc
    assert b._L.rt_scene_upload(b._h, C.byref(d0)) == 0
    x = np.ctypeslib.as_array(d0.xforms, shape=(int(d0.n_xforms) * 8,)).copy().view(np.int32).reshape(-1, 8)
    chain = int(np.nonzero(x[:, 0] == 8)[0][0])
    li = np.ctypeslib.as_array(d0.lights, shape=(int(d0.n_lights) * 8,)).copy().view(np.int32).reshape(-1, 8)
    assert li[:, 0].tolist() == [253, 3, 254, 0, 0]
    keep = []

    def xf(row, word, value):
        a = x.copy(); a[row, word] = value; keep.append(a)
        d = copy_of(d0); d.xforms = a.view(np.float32).ctypes.data_as(rt._ffi.c_float_p)
        return d

    def lt(row, word, value, n=None):
        a = li.copy(); a[row, word] = value; keep.append(a)
        d = copy_of(d0); d.lights = a.view(np.float32).ctypes.data_as(rt._ffi.c_float_p)
        if n is not None:
            d.n_lights = n
        return d

    cases = [
        (xf(chain, 2, 65), RT_E_INVALID, "xform chain"),                   # more than 64 records
        (xf(chain, 2, int(d0.n_xforms)), RT_E_INVALID, "xform chain"),     # (would also run past the array)
        (xf(chain, 1, -1), RT_E_INVALID, "xform chain"),
        (xf(int(x[chain, 1]), 0, 8), RT_E_INVALID, "not a plain record"),  # a chain inside a chain
        (lt(0, 2, 9), RT_E_INVALID, "root record"),                        # the root's run past the array
        (lt(0, 1, 0), RT_E_INVALID, "root record"),                        # ... or starting at itself
        (lt(2, 1, 2), RT_E_INVALID, "bad list record"),                    # a list that is its own member
        (lt(2, 1, 1), RT_E_INVALID, "bad list record"),                    # members before the record: a cycle
        (lt(2, 2, 1), RT_E_INVALID, "belongs to no list"),                 # the second sphere is nobody's member now
        (lt(1, 0, 254), RT_E_INVALID, "bad list record"),                  # a list record whose words are no run (a rectangle's floats)
        (lt(0, 0, 253, n=2), RT_E_INVALID, "root record"),                 # fewer records than the root's run
        (lt(0, 0, 254), RT_E_UNSUPPORTED_PRIMITIVE, "light 0: type 254"),  # list records need the root record
        (lt(1, 0, 9), RT_E_UNSUPPORTED_PRIMITIVE, "light 1: type 9"),
    ]
    for d, want, text in cases:
        st, msg = status_and_text(b, b._L.rt_scene_upload(b._h, C.byref(d)))
        assert st == want and text in msg, (st, msg, text)
    b.upload_scene(flat0)                                                  # the module's scene back in place
    b.upload_sampler(sampler)


def test_bad_sampler_tables_are_refused(rt, ctx):
    b, flat, params, camera, sampler = ctx
    d = copy_of(sampler.desc)
    d.sets = 0
    assert b._L.rt_sampler_upload(b._h, C.byref(d)) == RT_E_INVALID
    d = copy_of(sampler.desc)
    count = int(d.sets) * int(d.samples_in_set)
    idx = np.ctypeslib.as_array(sampler.desc.shuffled_indices, shape=(count,)).copy()
    idx[count // 2] = d.samples_in_set             # the kernels index samples[set][shuffled]: one past the end
    d.shuffled_indices = idx.ctypes.data_as(rt._ffi.c_u32_p)
    st, msg = status_and_text(b, b._L.rt_sampler_upload(b._h, C.byref(d)))
    assert st == RT_E_INVALID and "shuffled index" in msg


def test_bad_render_parameters_leave_the_image_alone(rt, ctx):
    b, flat, params, camera, sampler = ctx
    b.upload_scene(flat)
    b.upload_sampler(sampler)
    good, W, H = b._params(params, camera, 0x5EED, 0, None, None, None, None)
    good.out_location = rt._ffi.RT_MEM_HOST
    out = np.full((H, W, 4), 7.0, np.float32)
    stats = rt._ffi.rt_stats()

    def call(p, out_ptr=None, st=None):
        return status_and_text(b, b._L.rt_render(b._h, C.byref(p), out.ctypes.data if out_ptr is None else out_ptr, st))

    def changed(**kv):
        p = copy_of(good)
        for k, v in kv.items():
            setattr(p, k, v)
        return p

    cancel = C.c_int32(1)
    cases = [
        (changed(struct_size=good.struct_size + 8), RT_E_INVALID, "bad arguments"),
        (changed(image_width=1), RT_E_INVALID, "at least 2x2"),
        (changed(image_width=1 << 16, image_height=1 << 15), RT_E_INVALID, "too large"),
        (changed(samples_per_pixel=sampler.spp + 1), RT_E_INVALID, "does not match the sampler"),
        (changed(spp_begin=2, spp_count=sampler.spp), RT_E_INVALID, "sample range"),
        (changed(spp_begin=-1), RT_E_INVALID, "sample range"),
        (changed(projection=9), RT_E_INVALID, "projection"),
        (changed(max_ray_depth=65536), RT_E_INVALID, "max_ray_depth"),
        (changed(collect_stats=1), RT_E_INVALID, "stats"),
        (changed(cancel=C.pointer(cancel)), RT_E_CANCELLED, "cancelled"),
    ]
    for p, want, text in cases:
        st, msg = call(p)
        assert st == want and text in msg, (st, msg, text)
    assert b._L.rt_render(b._h, C.byref(good), None, None) == RT_E_INVALID          # no output buffer
    assert b._L.rt_render(b._h, None, out.ctypes.data, None) == RT_E_INVALID
    assert np.all(out == 7.0)
    # a tile outside the image is empty, one that overlaps it is clipped: both are fine (main.rs:1326-1352 clips its blocks too)
    assert call(changed(tile_x0=W, tile_x1=W + 8))[0] == 0 and np.all(out == 7.0)
    out[:] = 0.0
    assert call(changed(tile_x0=W - 8, tile_x1=W + 100, tile_y0=-5, tile_y1=4))[0] == 0
    full, _ = b.render(params, camera)
    assert np.array_equal(out[:4, W - 8:].view(np.uint32), full[:4, W - 8:].view(np.uint32))
    out[:4, W - 8:] = 0.0
    assert np.all(out == 0.0)
    cancel.value = 0                                # a cancel flag that stays 0 changes nothing
    st, _ = call(changed(cancel=C.pointer(cancel)), st=C.byref(stats))
    assert st == 0 and np.array_equal(out.view(np.uint32), full.view(np.uint32))


def test_resolve_and_first_hit_argument_checks(rt, ctx):
    b, flat, params, camera, sampler = ctx
    s = np.ones((4, 4), np.float32)
    o = np.zeros_like(s)
    assert b._L.rt_resolve(b._h, s.ctypes.data, 4, 0, o.ctypes.data, rt._ffi.RT_MEM_HOST, None) == RT_E_INVALID    # spp 0 divides by zero
    assert b._L.rt_resolve(b._h, None, 4, 4, o.ctypes.data, rt._ffi.RT_MEM_HOST, None) == RT_E_INVALID
    assert b._L.rt_resolve(b._h, s.ctypes.data, -1, 4, o.ctypes.data, rt._ffi.RT_MEM_HOST, None) == RT_E_INVALID
    assert np.all(o == 0.0)
    pid, t, nrm, ff, uv = b.trace_first_hit(np.zeros((7, 0), np.float32))             # no rays: nothing to do, no launch
    assert len(pid) == 0


def test_a_boundary_medium_needs_an_isotropic_material_whose_texture_reads_no_uv(rt, ctx):
    """rt_scene_upload reads the texture of a kind-1 medium's material: that material has to be Isotropic (the texture index of a
    Metal or Dielectric record is not range-checked and means nothing), and the walk through a checker must find an image below it."""
    b, flat, params, camera, sampler = ctx
    scene, *_ = rt.configs.build("SMOKE", image_width=32, samples_per_pixel=4)
    smoke = scene.flatten()
    d0 = smoke.desc
    assert d0.n_media >= 1
    media = np.ctypeslib.as_array(d0.media, shape=(d0.n_media * 8,)).copy()
    kinds = media.view(np.int32)[7::8]
    m = int(np.nonzero(kinds == 1)[0][0])                       # a medium with boundary slots (a Block under Translate / RotateY)
    mat = int(media.view(np.int32)[m * 8 + 5])
    mats = np.ctypeslib.as_array(d0.materials, shape=(d0.n_materials * 8,)).copy()
    texs = np.ctypeslib.as_array(d0.textures, shape=(d0.n_textures * 8,)).copy()
    assert mats.view(np.int32)[mat * 8] == 4                    # RTG_MAT_ISOTROPIC
    b.upload_scene(smoke)                                       # as built: accepted

    def with_arrays(materials=None, textures=None, n_textures=None):
        d = copy_of(d0)
        if materials is not None:
            d.materials = materials.ctypes.data_as(rt._ffi.c_float_p)
        if textures is not None:
            d.textures = textures.ctypes.data_as(rt._ffi.c_float_p)
            d.n_textures = n_textures
        return d

    metal = mats.copy()
    metal.view(np.int32)[mat * 8] = 1                           # RTG_MAT_METAL, and a texture word that points nowhere
    metal.view(np.int32)[mat * 8 + 1] = 0x7fffff00
    st, msg = status_and_text(b, b._L.rt_scene_upload(b._h, C.byref(with_arrays(materials=metal))))
    assert st == RT_E_INVALID and "not Isotropic" in msg, (st, msg)
    # a checker over (solid, image) as the phase function's texture: refused like a bare image
    nt = d0.n_textures
    more = np.zeros((nt + 2) * 8, np.float32)
    more[:nt * 8] = texs
    more.view(np.int32)[nt * 8:nt * 8 + 8] = [3, 0, 0, 0, -1, 0, 0, 0]                       # RTG_TEX_IMAGE without texels (cyan)
    more.view(np.int32)[(nt + 1) * 8:(nt + 1) * 8 + 8] = [1, 0, 0, 0, 0, nt, 0, 0]           # RTG_TEX_CHECKER: odd = texture 0, even = the image
    more[(nt + 1) * 8 + 6] = 4.0
    iso = mats.copy()
    iso.view(np.int32)[mat * 8 + 1] = nt + 1
    st, msg = status_and_text(b, b._L.rt_scene_upload(b._h, C.byref(with_arrays(materials=iso, textures=more, n_textures=nt + 2))))
    assert st == RT_E_UNSUPPORTED_PRIMITIVE and "image texture" in msg, (st, msg)
    b.upload_scene(flat)                                        # the context is still usable
    b.upload_sampler(sampler)
    assert np.isfinite(b.render(params, camera)[0]).all()


def test_stream_calls_check_their_arguments(rt, ctx):
    b, *_ = ctx
    L = b._L
    assert L.rt_stream_create(None, None) == RT_E_INVALID
    st, msg = status_and_text(b, L.rt_stream_create(b._h, None))
    assert st == RT_E_INVALID and "out is NULL" in msg
    s = b.stream_create()
    assert s
    assert L.rt_stream_wait(b._h, s, s) == 0 and L.rt_stream_wait(b._h, s, None) == 0 and L.rt_stream_wait(b._h, None, s) == 0
    assert L.rt_stream_wait(None, s, None) == RT_E_INVALID
    assert L.rt_stream_destroy(b._h, None) == 0 and L.rt_stream_destroy(None, s) == RT_E_INVALID
    b.stream_destroy(s)


def test_tuning_knobs_are_ignored_unless_RT_KNOBS_is_set(rt, ctx):
    # a 1 MB sample buffer cannot hold one sample of this window: with the knobs on the render is refused, without RT_KNOBS the
    # variable is not even looked at (a product run is what the defaults say)
    import os
    b, flat, params, camera, sampler = ctx
    b.upload_scene(flat)
    b.upload_sampler(sampler)
    want, _ = b.render(params, camera, width=512, height=512)
    old = {k: os.environ.pop(k, None) for k in ("RT_KNOBS", "RT_WF_SAMPLES_MB")}
    try:
        os.environ["RT_WF_SAMPLES_MB"] = "1"
        got, _ = b.render(params, camera, width=512, height=512)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        os.environ["RT_KNOBS"] = "1"
        with pytest.raises(rt._ffi.RtError) as e:
            b.render(params, camera, width=512, height=512)
        assert e.value.status == -4 and "does not fit the sample buffer" in str(e.value)      # RT_E_OOM
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_a_cancel_raised_in_mid_render_ends_the_call_and_leaves_the_context_usable(rt, ctx):
    """main.rs:1338: the workers poll a flag between blocks; rt_render polls `cancel` between groups of iterations.  A render that
    is cancelled half-way returns RT_E_CANCELLED once the device is idle; the next render gives the bits it always gave."""
    import os
    import threading
    b, flat, params, camera, sampler = ctx
    b.upload_scene(flat)
    b.upload_sampler(sampler)
    want, _ = b.render(params, camera)
    p, W, H = b._params(params, camera, 0x5EED, 0, None, None, 2048, 2048)
    p.out_location = rt._ffi.RT_MEM_DEVICE
    acc = b.device_alloc(W * H * 16)
    cancel = C.c_int32(0)
    p.cancel = C.pointer(cancel)
    # 1.7e7 paths through a pool of 16 Ki entries per pipeline: thousands of iterations, a tenth of a second at least
    old = {k: os.environ.get(k) for k in ("RT_KNOBS", "RT_WF_POOL_LOG2")}
    os.environ.update(RT_KNOBS="1", RT_WF_POOL_LOG2="14")
    try:
        timer = threading.Timer(0.02, lambda: setattr(cancel, "value", 1))
        timer.start()
        st = b._L.rt_render(b._h, C.byref(p), acc, None)          # (ctypes releases the GIL for the call)
        timer.join()
        assert st == RT_E_CANCELLED, (st, b._L.rt_last_error(b._h))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        b.device_free(acc)
    got, _ = b.render(params, camera)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
