"""Multi-GPU partitioning of a render: one process per GPU, torch.distributed for the exchange.

The reference is single-device (one OpenCL queue, /root/reference/src/renderer/main.cpp:522);
nothing here has a counterpart in it.  Every pixel-sample is independent, so
the path shards two ways:

  - sample split: every rank renders the whole frame for its own contiguous
    range of sample indices (STRIDED generator mode + SUM accumulation) and the
    partial sums are combined by one reduce(sum) to a root rank;
  - tile split: the frame is cut into tiles dealt round-robin to the ranks,
    every rank renders all samples of its own tiles and the root gathers them
    (no arithmetic in the exchange).

The functions take the renderer as a small protocol (set_sample_offset,
set_region, set_tile_interleave, num_samples, render, render_async,
clear_accum) so that the GPU context and a CPU stand-in used
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
        if dist.get_backend() == "gloo" and partial_sum.is_cuda:
            host = partial_sum.cpu()            # rehearsal of the N > 1 path on one GPU: gloo reduces host tensors
            dist.reduce(host, dst=root, op=dist.ReduceOp.SUM)
            if dist.get_rank() == root:
                partial_sum.copy_(host)
        else:
            dist.reduce(partial_sum, dst=root, op=dist.ReduceOp.SUM)
        if dist.get_rank() != root:
            return None
    return partial_sum / float(total_spp)


TILE = (64, 32)     # pixels; multiples of the workgroup's 16 x 8


def tile_owner_map(height, width, world, tile=TILE):
    """int32 [height, width]: the rank that owns each pixel when the frame's tiles (numbered row by row) are dealt
    round-robin to `world` ranks."""
    import numpy as np
    tw, th = tile
    tiles_x = (width + tw - 1) // tw
    ty, tx = np.meshgrid(np.arange(height) // th, np.arange(width) // tw, indexing="ij")
    return ((ty * tiles_x + tx) % world).astype(np.int32)


def render_tile_split(renderer, height, width, spp, world, rank, tile=TILE, wait=True):
    """Renders `spp` more samples of this rank's tiles: tile t of the frame belongs to rank t mod world.  Eye rays
    that miss the scene end a path after one ray, so contiguous bands of a frame are badly balanced; interleaved
    tiles give every rank the same mix."""
    renderer.set_region(0, 0, width, height)
    renderer.set_tile_interleave(tile[0], tile[1], world, rank)
    (renderer.render if wait else renderer.render_async)(spp)


def render_band_split(renderer, height, width, spp, world, rank):
    """The contiguous alternative (rank r renders row band r); kept to measure the balance against. Returns (y0, h)."""
    y0, h = row_band(height, world, rank)
    renderer.set_tile_interleave(0, 0, 1, 0)
    if h > 0:
        renderer.set_region(0, y0, width, h)
        renderer.render(spp)
    return y0, h


_owned_cache = {}


def _owned_pixels(height, width, world, rank, tile, device):
    """Flat indices (y*width + x) of the pixels of `rank`, as a torch int64 tensor on `device`."""
    import torch
    key = (height, width, world, rank, tile, str(device))
    if key not in _owned_cache:
        owner = tile_owner_map(height, width, world, tile)
        _owned_cache[key] = torch.from_numpy((owner.reshape(-1) == rank).nonzero()[0]).to(device)
    return _owned_cache[key]


def gather_tile_split(image, height, world, dist=None, root=0, tile=TILE):
    """Gathers every rank's tiles of `image` ([H, W, 4] torch tensor, device for nccl, device or cpu for gloo) into
    root's copy: each rank packs its pixels into one contiguous buffer and sends it; no arithmetic."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return image
    width = image.shape[1]
    rank = dist.get_rank()
    via_host = dist.get_backend() == "gloo" and image.is_cuda
    flat = image.view(-1, image.shape[2])
    if rank == root:
        for r in range(world):
            if r == root:
                continue
            idx = _owned_pixels(height, width, world, r, tile, image.device)
            if idx.numel() == 0:
                continue
            buf = flat.new_empty((idx.numel(), flat.shape[1]), device="cpu" if via_host else image.device)
            dist.recv(buf, src=r)
            flat[idx] = buf.to(image.device)
        return image
    idx = _owned_pixels(height, width, world, rank, tile, image.device)
    if idx.numel() > 0:
        buf = flat[idx].contiguous()
        dist.send(buf.cpu() if via_host else buf, dst=root)
    return None
