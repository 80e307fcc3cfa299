"""Multi-GPU split of a frame: who renders what, and the one exchange that makes the image.

The reference tiles the image over threads (main.rs:1265-1282) with no inter-tile state, so pixels and samples are
independent.  Two partitions are offered, both with the scene replicated on every GPU:

* sample ranges (`sample_range`, the default): every rank renders the WHOLE frame for its own contiguous sample range —
  perfect balance, no tile skew; the counter RNG is keyed by the absolute sample index, so the image does not depend on the
  world size (up to the association of the final f32 sum);
* interleaved tiles (`tile_rects`, BASELINE.json's wording): tile t of the B x B grid belongs to rank t mod world; a rank
  renders its rectangles into a zero-initialised full-frame buffer, so the same sum assembles the frame.

The exchange is ONE sum of W*H*4 f32 to rank 0 — `Communicator.reduce_to_root`, i.e. `rt_comm_reduce_to_root` of the C-ABI
(include/rt_api.h): an in-place RCCL reduce on the render's stream.  Every GPU has a direct xGMI link to the root and nobody
else needs the result, so no all-reduce and no ring.  `rt_group_*` is the same for one process driving n GPUs.
`reduce_to_root(acc, dist)` keeps the torch.distributed form for CPU tensors (gloo; the world-size-2 tests on a machine
without GPUs).
"""
import ctypes as C

from . import _ffi
from ._ffi import RtError


def sample_range(rank, world, spp):
    """[begin, begin+count) of rank: contiguous, disjoint, covering [0, spp); earlier ranks take the remainder."""
    if not (0 <= rank < world) or spp < 0:
        raise ValueError("bad rank/world/spp")
    base, rem = divmod(spp, world)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def tile_rects(rank, world, width, height, block=32):
    """The (x0, y0, x1, y1) rectangles of rank under interleaved tile ownership: tile t (row-major over the block grid, as
    main.rs:1265-1282 numbers its work blocks) belongs to rank t mod world.  Rectangles are clipped to the image; together
    the ranks' rectangles cover every pixel exactly once."""
    if not (0 <= rank < world) or width <= 0 or height <= 0 or block <= 0:
        raise ValueError("bad rank/world/size")
    nx, ny = (width + block - 1) // block, (height + block - 1) // block
    out = []
    for t in range(rank, nx * ny, world):
        ty, tx = divmod(t, nx)
        out.append((tx * block, ty * block, min(width, (tx + 1) * block), min(height, (ty + 1) * block)))
    return out


def reduce_to_root(acc, dist=None, root=0):
    """Sum the per-rank accumulation images into rank `root` (in place) through torch.distributed: `acc` is a torch tensor
    on the CPU (gloo, tests) or on the rank's device (nccl = RCCL)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(acc, dst=root)
    return acc


class Communicator:
    """rt_comm of the C-ABI: one per rank of a one-process-per-GPU job.

    `exchange(obj_list)` is whatever channel the host has for the 128-byte id: it receives a one-element list (the id on rank
    0, None elsewhere) and must fill it with rank 0's value on every rank — `torch.distributed.broadcast_object_list` fits."""

    def __init__(self, backend, rank, world, exchange=None):
        self._L = _ffi.hip_lib()
        self._backend = backend
        self.rank, self.world = int(rank), int(world)
        box = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(_ffi.RT_COMM_ID_BYTES)
            st = self._L.rt_comm_unique_id(buf)
            if st < 0:
                raise RtError(st, self._L.rt_last_error(None).decode())
            box[0] = buf.raw
        if self.world > 1:
            if exchange is None:
                raise ValueError("a world of %d ranks needs an exchange function for the communicator id" % self.world)
            exchange(box)
        self._h = C.c_void_p()
        st = self._L.rt_comm_create(backend._h, box[0], self.rank, self.world, C.byref(self._h))
        if st < 0:
            raise RtError(st, self._L.rt_last_error(backend._h).decode())

    def reduce_to_root(self, device_ptr, n_floats, root=0, stream_ptr=None):
        """Asynchronous on stream_ptr: on `root` the buffer then holds the sum over all ranks."""
        st = self._L.rt_comm_reduce_to_root(self._h, device_ptr, int(n_floats), int(root), stream_ptr)
        if st < 0:
            raise RtError(st, self._L.rt_last_error(self._backend._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
