"""Multi-GPU sharding of the path: one process per GPU, image rows dealt to the
ranks as interleaved stripes, NO data-path collective while rendering, one gather of the tiles
at the end (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The RNG is keyed by the global pixel, so the assembled image is bit-identical for any world
size / stripe height (tests/test_distributed_cpu.py, tests/test_gpu_parity.py).
"""
import numpy as np

# Single rows: every rank gets height / world rows +- 1 whatever the height.  (8-row stripes until round 4: at world 8 a rank's tile then cost 4 % more than
# an eighth of the frame — ragged 128 / 136 rows and, row for row, slower launches; profiles/experiments/r04_stripe_height_sweep.py.)
DEFAULT_STRIPE_H = 1
_plans = {}


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


def source_of_rows(height, world, stripe_h=DEFAULT_STRIPE_H):
    """For every global row y: (rank that rendered it, its local row there)."""
    src_rank = np.zeros(height, dtype=np.int64)
    src_row = np.zeros(height, dtype=np.int64)
    for r in range(world):
        for lr, y in enumerate(rows_of_rank(height, world, r, stripe_h)):
            src_rank[y], src_row[y] = r, lr
    return src_rank, src_row


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


def collective_mode(backend, device_type):
    """Which collective brings the tiles to `dst`, decided from facts every rank shares (never by catching an error on one
    rank and retrying with another collective: a rank-local failure would leave the ranks inside different collectives)."""
    if backend == "nccl" or device_type == "cpu":
        return "gather"           # RCCL (ncclSend/ncclRecv under the hood) and gloo-on-CPU implement gather
    return "all_gather"           # gloo with device tensors (the one-GPU rehearsal): gather is not implemented there


def gather_image(local_tile, height, stripe_h=DEFAULT_STRIPE_H, dst=0):
    """Gather every rank's tile (torch tensor [3, rows_r, W]) on `dst` and assemble [3, height, W] there.

    One collective per frame: tiles are padded to the largest tile and land in one [world, 3, rows, W] buffer on `dst`
    (3.1 MB per GPU at 1080p / 8 GPUs in f32).  With RCCL a gather is world-1 point-to-point transfers that arrive over
    separate xGMI links at once; an all-gather would also ship every tile to every other rank (7x the traffic on a ring)
    and is used only where the backend has no gather for the tensor's device.  The row permutation back to image order is
    one indexed copy with index tensors built once per geometry.  Returns None off `dst`.
    """
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return local_tile
    W = local_tile.shape[2]
    key = (height, world, stripe_h, W, local_tile.dtype, str(local_tile.device))
    plan = _plans.get(key)
    if plan is None:
        mr = max_rows(height, world, stripe_h)
        sr, sl = source_of_rows(height, world, stripe_h)
        plan = dict(mr=mr, stacked=local_tile.new_empty((world, 3, mr, W)), padded=local_tile.new_zeros((3, mr, W)),
                    src_rank=torch.as_tensor(sr, device=local_tile.device), src_row=torch.as_tensor(sl, device=local_tile.device),
                    mode=collective_mode(dist.get_backend(), local_tile.device.type))
        _plans[key] = plan
    if local_tile.shape[1] == plan["mr"]:
        padded = local_tile.contiguous()
    else:
        padded = plan["padded"]
        padded[:, :local_tile.shape[1]] = local_tile
    if plan["mode"] == "gather":
        dist.gather(padded, list(plan["stacked"].unbind(0)) if rank == dst else None, dst=dst)
    else:
        dist.all_gather(list(plan["stacked"].unbind(0)), padded)
    if rank != dst:
        return None
    return plan["stacked"][plan["src_rank"], :, plan["src_row"]].permute(1, 0, 2).contiguous()
