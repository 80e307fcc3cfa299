"""N>1 path on CPU: two gloo ranks shard the samples of one frame, the per-rank sums are
reduced to rank 0 and must equal the single-process frame.  The renderer here is the oracle
(no GPU in this container); the sharding + reduce code is the product's (sharding.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    rt = O.rt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, params, camera, sampler = rt.configs.build("C2", image_width=40, samples_per_pixel=9)
    begin, count = rt.sharding.sample_range(rank, world, sampler.spp)
    img, rays = O.OracleScene(scene).render(params, camera, sampler, spp_begin=begin, spp_count=count, threads=2)
    acc = torch.from_numpy(img)
    rt.sharding.reduce_to_root(acc, dist)
    r = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(r)
    if rank == 0:
        np.savez(out_path, img=acc.numpy(), rays=r.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _tile_worker(rank, world, port, out_path):
    """BASELINE's wording, "image tiles sharded across the GPUs": every rank renders its interleaved tiles of the whole sample
    range into a zero-initialised frame; the same reduce assembles the image."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    rt = O.rt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, params, camera, sampler = rt.configs.build("C2", image_width=40, samples_per_pixel=4)
    oracle = O.OracleScene(scene)
    acc = np.zeros((params.image_height, params.image_width, 4), np.float32)
    for rect in rt.sharding.tile_rects(rank, world, params.image_width, params.image_height, block=16):
        img, _ = oracle.render(params, camera, sampler, tile=rect, threads=2)
        acc += img
    t = torch.from_numpy(acc)
    rt.sharding.reduce_to_root(t, dist)
    if rank == 0:
        np.savez(out_path, img=t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_of_interleaved_tiles_equal_one_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "t0.npz")
    mp.spawn(_tile_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)["img"]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    rt = O.rt
    scene, params, camera, sampler = rt.configs.build("C2", image_width=40, samples_per_pixel=4)
    full, _ = O.OracleScene(scene).render(params, camera, sampler)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))      # disjoint pixels: x + 0 = x, bit for bit


def test_two_ranks_equal_one_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    rt = O.rt
    scene, params, camera, sampler = rt.configs.build("C2", image_width=40, samples_per_pixel=9)
    full, rays = O.OracleScene(scene).render(params, camera, sampler)
    assert int(got["rays"][0]) == rays
    np.testing.assert_allclose(got["img"], full, rtol=1e-5, atol=1e-6)
    assert (got["img"][..., 3] == 9).all()
