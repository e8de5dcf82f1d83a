"""Multi-GPU partitioning of a render: one process per GPU, torch.distributed for the exchange.

The reference is single-device (one OpenCL queue, /root/reference/src/renderer/main.cpp:522);
nothing here has a counterpart in it.  Every pixel-sample is independent, so
the path shards two ways:

  - sample split: every rank renders the whole frame for its own contiguous
    range of sample indices (STRIDED generator mode + SUM accumulation) and the
    partial sums are combined by one reduce(sum) to a root rank;
  - tile split: every rank renders all samples of its own pixel rows and the
    root gathers the bands (no arithmetic in the exchange).

The functions take the renderer as a small protocol (set_sample_offset,
set_region, num_samples, render) so that the GPU context and a CPU stand-in used
by the multi-process tests drive the same code.
"""


def sample_range(total_spp, world, rank):
    """Contiguous sample indices [first, first+count) of `rank`; remainders go to the low ranks."""
    if world < 1 or not 0 <= rank < world or total_spp < 0:
        raise ValueError("bad split %r/%r of %r samples" % (rank, world, total_spp))
    base, rem = divmod(total_spp, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def row_band(height, world, rank, align=8):
    """Rows [y0, y0+h) of `rank` for a tile split; bands are multiples of `align`
    rows (the wavefront tile height) except the last non-empty one."""
    if world < 1 or not 0 <= rank < world or height < 0:
        raise ValueError("bad split %r/%r of %r rows" % (rank, world, height))
    blocks = (height + align - 1) // align
    base, rem = divmod(blocks, world)
    b0 = rank * base + min(rank, rem)
    nb = base + (1 if rank < rem else 0)
    y0 = min(b0 * align, height)
    y1 = min((b0 + nb) * align, height)
    return y0, y1 - y0


def render_sample_split(renderer, total_spp, world, rank, wait=True):
    """Renders this rank's share of `total_spp` samples of a fresh image. Returns the samples rendered.
    wait=False only enqueues the launch (the reduce that follows is ordered after it by the stream)."""
    first, count = sample_range(total_spp, world, rank)
    renderer.set_sample_offset(first)
    renderer.num_samples = 0
    if count > 0:
        (renderer.render if wait else renderer.render_async)(count)
    else:
        renderer.clear_accum()      # fewer samples than ranks: this rank adds nothing, not what an earlier frame left
    return count


def combine_sample_split(partial_sum, total_spp, dist=None, root=0):
    """reduce(sum) of the per-rank partial sums to `root`, then the mean on root.
    `partial_sum` is a torch tensor (device for nccl, cpu for gloo), updated in place."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(partial_sum, dst=root, op=dist.ReduceOp.SUM)
        if dist.get_rank() != root:
            return None
    return partial_sum / float(total_spp)


def render_tile_split(renderer, height, width, spp, world, rank):
    """Renders `spp` more samples of this rank's row band. Returns (y0, h)."""
    y0, h = row_band(height, world, rank)
    if h > 0:
        renderer.set_region(0, y0, width, h)
        renderer.render(spp)
    return y0, h


def gather_tile_split(image, height, world, dist=None, root=0):
    """Gathers the row bands of every rank's `image` ([H, W, 4] torch tensor) into root's copy."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return image
    rank = dist.get_rank()
    if rank == root:
        for r in range(world):
            if r == root:
                continue
            y0, h = row_band(height, world, r)
            if h > 0:
                dist.recv(image[y0:y0 + h], src=r)
        return image
    y0, h = row_band(height, world, rank)
    if h > 0:
        dist.send(image[y0:y0 + h].contiguous(), dst=root)
    return None
