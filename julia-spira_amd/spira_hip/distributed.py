"""Multi-GPU sharding of the path: one process per GPU, image rows dealt to the
ranks as interleaved stripes, NO data-path collective while rendering, one gather of the tiles
at the end (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The RNG is keyed by the global pixel, so the assembled image is bit-identical for any
world size / stripe height (tests/test_distributed_cpu.py, tests/test_gpu_parity.py).
"""
import numpy as np

DEFAULT_STRIPE_H = 8


def rows_of_rank(height, world, rank, stripe_h=DEFAULT_STRIPE_H):
    """Global output rows (top-based) rendered by `rank`, in its local order."""
    if world <= 1:
        return list(range(height))
    ys = []
    for y0 in range(rank * stripe_h, height, world * stripe_h):
        ys.extend(range(y0, min(y0 + stripe_h, height)))
    return ys


def tile_params(height, world, rank, stripe_h=DEFAULT_STRIPE_H):
    """Tiling fields of spira_params for this rank (include/spira_hip.h, "Tiling")."""
    if world <= 1:
        return dict(row0=0, rows=0, stripe_h=0, stripe_count=0, stripe_rank=0)
    return dict(row0=0, rows=len(rows_of_rank(height, world, rank, stripe_h)), stripe_h=stripe_h, stripe_count=world,
                stripe_rank=rank)


def max_rows(height, world, stripe_h=DEFAULT_STRIPE_H):
    return max(len(rows_of_rank(height, world, r, stripe_h)) for r in range(world))


def assemble(tiles, height, world, stripe_h=DEFAULT_STRIPE_H):
    """tiles[r]: array-like [3, >= rows_r, W] (numpy or torch) -> full image [3, height, W]."""
    first = tiles[0]
    if hasattr(first, "new_empty"):   # torch
        import torch
        out = first.new_empty((3, height, first.shape[2]))
        for r in range(world):
            ys = rows_of_rank(height, world, r, stripe_h)
            out[:, torch.as_tensor(ys, device=out.device)] = tiles[r][:, :len(ys)]
        return out
    out = np.empty((3, height, first.shape[2]), dtype=first.dtype)
    for r in range(world):
        ys = rows_of_rank(height, world, r, stripe_h)
        out[:, ys] = np.asarray(tiles[r])[:, :len(ys)]
    return out


def gather_image(local_tile, height, stripe_h=DEFAULT_STRIPE_H, dst=0):
    """Gather every rank's tile (torch tensor [3, rows_r, W]) to `dst` and assemble [3, height, W].

    One collective: tiles are padded to the largest tile so a single all_gather moves
    3*rows*W values per rank (3.1 MB per GPU at 1080p / 8 GPUs).  Returns None off `dst`.
    """
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return local_tile
    mr = max_rows(height, world, stripe_h)
    if local_tile.shape[1] != mr:
        padded = local_tile.new_zeros((3, mr, local_tile.shape[2]))
        padded[:, :local_tile.shape[1]] = local_tile
    else:
        padded = local_tile.contiguous()
    # all_gather is the collective every backend implements natively (RCCL: one ncclAllGather); the extra
    # copies on the non-destination ranks are 25 MB at 1080p and keep the code to one well-trodden call
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    if rank != dst:
        return None
    return assemble(bufs, height, world, stripe_h)
