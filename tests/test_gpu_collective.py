"""The RCCL side of the C-ABI (rt_device_*, rt_comm_*, rt_group_*; include/rt_api.h "multi-GPU") on the one GPU a test box has.

With one rank the reduce is the identity, so what is tested is the plumbing a multi-GPU host relies on: RCCL is found and
initialised through the library (the copy that belongs to the HIP runtime the library runs on), device buffers come from the
library itself (a Rust host has no HIP binding), the calls are ordered on the render's stream, the image on the root is the
render's, and bad arguments come back as RT_E_INVALID without touching anything.  The N > 1 arithmetic (sample ranges and tiles
add up to the single-process frame) is covered on CPU by tests/test_distributed_cpu.py (gloo, 2 ranks) and, on one GPU, by
tests/test_gpu_fullsize.py::test_c5_book2_4k_one_rank_of_eight.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RT_E_INVALID, RT_E_RCCL = -1, -3


def test_comm_of_one_rank_reduces_in_place_on_the_render_stream(rt, gpu):
    scene, params, camera, sampler = rt.configs.build("C2", image_width=48, samples_per_pixel=4)
    gpu.upload_scene(scene.flatten())
    gpu.upload_sampler(sampler)
    ref, _ = gpu.render(params, camera)
    shape = (params.image_height, params.image_width, 4)
    nbytes = int(np.prod(shape)) * 4
    acc = gpu.device_alloc(nbytes)
    comm = rt.sharding.Communicator(gpu, 0, 1)
    try:
        for _ in range(2):      # twice: the communicator is reusable
            gpu.device_zero(acc, nbytes)
            gpu.render_device(params, camera, acc, None)
            comm.reduce_to_root(acc, nbytes // 4, 0, None)
            got = gpu.device_read(acc, shape)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        L = gpu._L
        assert L.rt_comm_reduce_to_root(comm._h, acc, nbytes // 4, 1, None) == RT_E_INVALID      # root outside the world
        assert L.rt_comm_reduce_to_root(comm._h, None, 16, 0, None) == RT_E_INVALID
        assert L.rt_comm_reduce_to_root(comm._h, acc, 0, 0, None) == 0                            # nothing to do
    finally:
        comm.close()
        gpu.device_free(acc)
    h = C.c_void_p()
    ident = C.create_string_buffer(rt._ffi.RT_COMM_ID_BYTES)
    assert gpu._L.rt_comm_create(gpu._h, ident, 3, 2, C.byref(h)) == RT_E_INVALID and not h.value            # rank outside the world
    assert gpu._L.rt_device_alloc(gpu._h, 0, C.byref(h)) == RT_E_INVALID


def test_group_of_one_device(rt):
    L = rt._ffi.hip_lib()
    g = C.c_void_p()
    devs = (C.c_int32 * 2)(0, 0)
    assert L.rt_group_create(devs, 2, C.byref(g)) == RT_E_INVALID and b"twice" in L.rt_last_error(None)
    assert L.rt_group_create(devs, 0, C.byref(g)) == RT_E_INVALID
    st = L.rt_group_create(devs, 1, C.byref(g))
    assert st == 0, L.rt_last_error(None)
    try:
        assert L.rt_group_size(g) == 1 and L.rt_group_ctx(g, 1) is None
        b = rt.backend.HipBackend.__new__(rt.backend.HipBackend)      # a backend object over the group's context (owned by the group)
        b._L, b._h, b.device_id, b._flat, b._sampler = L, C.c_void_p(L.rt_group_ctx(g, 0)), 0, None, None
        scene, params, camera, sampler = rt.configs.build("C1", image_width=64, samples_per_pixel=4)
        b.upload_scene(scene.flatten())
        b.upload_sampler(sampler)
        ref, _ = b.render(params, camera)
        shape = (params.image_height, params.image_width, 4)
        nbytes = int(np.prod(shape)) * 4
        acc = b.device_alloc(nbytes)
        b.device_zero(acc, nbytes)
        b.render_device(params, camera, acc, None)
        bufs = (C.c_void_p * 1)(acc)
        assert L.rt_group_reduce_to_root(g, bufs, nbytes // 4, 0, None) == 0, L.rt_last_error(b._h)
        assert np.array_equal(b.device_read(acc, shape).view(np.uint32), ref.view(np.uint32))
        assert L.rt_group_reduce_to_root(g, bufs, nbytes // 4, 2, None) == RT_E_INVALID
        b.device_free(acc)
        b._h = None      # the group destroys its contexts
    finally:
        L.rt_group_destroy(g)
