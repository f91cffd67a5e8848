"""CPU, world_size 2 (gloo): the N>1 path — stripe sharding, one gather, reassembly — reproduces the
unsharded image bit for bit.  The tile renderer here is the CPU oracle (tests may use it); on the
GPU box the same code path runs with the HIP renderer (tests/test_gpu_parity.py, bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spira_hip import distributed as D
from spira_hip import scenes

W, H, SPP, DEPTH, SEED = 48, 27, 2, 3, 11


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, stripe_h, out_path, force_mode=None):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "julia-spira_amd"), os.path.join(root, "oracle")]
    import oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if force_mode:        # the branch taken for gloo with device tensors (one-GPU rehearsal), exercised here on CPU tensors
        D.collective_mode = lambda backend, device_type: force_mode
    s = scenes.scene_s2()
    tp = D.tile_params(H, world, rank, stripe_h)
    p = O.make_params(W, H, SPP, DEPTH, 5, 6, 1, seed=SEED, **tp)
    tile, _, _ = O.render(s["spheres5"], s["materials8"], s["triangles10"], s["camera12"], p, "f32", n_threads=1)
    img = D.gather_image(torch.from_numpy(tile), H, stripe_h)
    if rank == 0:
        np.save(out_path, img.numpy())
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_stripes_equal_full(tmp_path, oracle):
    s = scenes.scene_s2()
    full, _, _ = oracle.render(s["spheres5"], s["materials8"], s["triangles10"], s["camera12"],
                               oracle.make_params(W, H, SPP, DEPTH, 5, 6, 1, seed=SEED), "f32")
    for stripe_h in (4, 8):   # 27 rows: ragged last stripe, unequal tile heights (padding path)
        out = str(tmp_path / ("img_%d.npy" % stripe_h))
        mp.spawn(_worker, args=(2, _free_port(), stripe_h, out), nprocs=2, join=True)
        assert np.array_equal(np.load(out), full)


def _worker8(rank, world, port, out_path, W8, H8, spp, depth):
    """configs[3]'s sharding on 8 ranks: H = 1080 rows dealt row by row, spp 256, depth 8 on the closed box S3 (a narrow frame: the CPU renders it)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "julia-spira_amd"), os.path.join(root, "oracle")]
    import oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = scenes.scene_s3()
    tp = D.tile_params(H8, world, rank)
    assert abs(tp["rows"] - H8 // world) <= 1, tp
    p = O.make_params(W8, H8, spp, depth, len(s["spheres5"]), len(s["materials8"]), len(s["triangles10"]), seed=scenes.seed_for(4), **tp)
    tile, _, seg = O.render(s["spheres5"], s["materials8"], s["triangles10"], s["camera12"], p, "f64", n_threads=1)
    assert 0.9 * W8 * tp["rows"] * spp * depth < seg <= W8 * tp["rows"] * spp * depth      # the closed box: (nearly) every path runs all its segments
    img = D.gather_image(torch.from_numpy(tile), H8)
    if rank == 0:
        np.save(out_path, img.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_stripes_at_1080_rows_spp_256(tmp_path, oracle):
    """BASELINE configs[3] (1080p, spp 256, depth 8, tile-sharded over 8 GPUs) at its own height and spp on 8 gloo ranks: every rank gets
    135 rows (rows r, r + 8, ...), and the frame
    assembled from the 8 tiles is bit-identical to the unsharded render.  (Width 12 instead of 1920: the oracle is the renderer here.)"""
    W8, H8, spp, depth, world = 12, 1080, 256, 8, 8
    rows = [D.tile_params(H8, world, r)["rows"] for r in range(world)]
    assert rows == [135] * 8, rows
    assert sorted(sum((D.rows_of_rank(H8, world, r) for r in range(world)), [])) == list(range(H8))
    s = scenes.scene_s3()
    full, _, _ = oracle.render(s["spheres5"], s["materials8"], s["triangles10"], s["camera12"],
                               oracle.make_params(W8, H8, spp, depth, len(s["spheres5"]), len(s["materials8"]), len(s["triangles10"]), seed=scenes.seed_for(4)), "f64")
    out = str(tmp_path / "img8.npy")
    mp.spawn(_worker8, args=(world, _free_port(), out, W8, H8, spp, depth), nprocs=world, join=True)
    assert np.array_equal(np.load(out), full)


def test_two_rank_all_gather_branch(tmp_path, oracle):
    s = scenes.scene_s2()
    full, _, _ = oracle.render(s["spheres5"], s["materials8"], s["triangles10"], s["camera12"],
                               oracle.make_params(W, H, SPP, DEPTH, 5, 6, 1, seed=SEED), "f32")
    out = str(tmp_path / "img_ag.npy")
    mp.spawn(_worker, args=(2, _free_port(), 8, out, "all_gather"), nprocs=2, join=True)
    assert np.array_equal(np.load(out), full)


def test_collective_is_chosen_from_shared_facts_only():
    assert D.collective_mode("nccl", "cuda") == "gather" and D.collective_mode("gloo", "cpu") == "gather"
    assert D.collective_mode("gloo", "cuda") == "all_gather"


def test_assemble_numpy():
    rng = np.random.default_rng(0)
    full = rng.random((3, 37, 5)).astype(np.float32)
    for world, sh in [(1, 8), (2, 8), (3, 4), (8, 8), (5, 1)]:
        tiles = [full[:, D.rows_of_rank(37, world, r, sh)] for r in range(world)]
        mr = D.max_rows(37, world, sh)
        padded = [np.concatenate([t, np.zeros((3, mr - t.shape[1], 5), np.float32)], axis=1) for t in tiles]
        assert np.array_equal(D.assemble(padded, 37, world, sh), full)
        assert sorted(sum((D.rows_of_rank(37, world, r, sh) for r in range(world)), [])) == list(range(37))
