"""Multi-GPU split of a frame: one process per GPU, sample ranges, one reduce to rank 0.

The reference tiles the image over threads (main.rs:1265-1282) with no inter-tile state, so
pixels and samples are independent.  Samples are the better shard key on 8 GPUs: every rank
renders the WHOLE frame for its own sample range (perfect balance, no tile skew), the counter
RNG is keyed by the absolute sample index so the image does not depend on the world size, and
the only exchange is one sum of W*H*4 f32 to rank 0 (RCCL reduce over xGMI: every GPU has a
direct link to the root, nobody else needs the result, so no all-reduce).
"""


def sample_range(rank, world, spp):
    """[begin, begin+count) of rank: contiguous, disjoint, covering [0, spp); earlier ranks take the remainder."""
    if not (0 <= rank < world) or spp < 0:
        raise ValueError("bad rank/world/spp")
    base, rem = divmod(spp, world)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def reduce_to_root(acc, dist=None, root=0):
    """Sum the per-rank accumulation images into rank `root` (in place).  `acc` is a torch tensor
    on the rank's device (nccl = RCCL) or on the CPU (gloo, tests)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(acc, dst=root)
    return acc
