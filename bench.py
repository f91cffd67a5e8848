#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-trace hot path on MI355X (BASELINE.json metric).

A "step" is one whole render of the workload.  At N=1 the workload is BASELINE configs[2]:
1920x1080, spp=64, max_depth=8 on the reference's own scene (create_scene() of
src/spira-metal-optimized.jl:429-510 with main()'s camera :1499-1505), synthetic by construction.
At N>1 the frame is tile-sharded over the ranks (interleaved 8-row stripes, one process per GPU,
no collective while rendering, ONE RCCL gather of the tiles per step) and spp = 64*N, so the
per-GPU work is fixed ("weak").  Scene and camera are uploaded per render (a few hundred bytes);
outputs stay in HBM.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (k_bounce) from a
separate, event-bracketed render after the timed region; `cpu_baseline` times the CPU oracle
(a port: the Julia reference cannot run here) on the host cores, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "julia-spira_amd"), os.path.join(ROOT, "oracle")]

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(c, prec_bytes):
    """HBM bytes the wavefront formulation must move (DESIGN.md "Roofline"), from device counters:
    every enqueued ray is written once and read once (10 values), a radiance term is a 3-value store, or a
    read-modify-write (2 x 3 values) when the path already holds radiance."""
    return (2 * 10 * c["rays_enqueued"] + 3 * c["radiance_stores"] + 6 * c["radiance_rmw"]) * prec_bytes


def host_cpu_share():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota (cpu.max / cfs_quota_us)."""
    n = len(os.sched_getaffinity(0))
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, -(-int(quota) // int(period))))
            break
        except (OSError, ValueError):
            continue
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel PER GPU (total = spp * gpus)")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--scene", default="s1", choices=["s1", "s2", "s3", "s4"])
    ap.add_argument("--kernel", default="wavefront", choices=["wavefront", "mega"])
    ap.add_argument("--prec", default="f64", choices=["f32", "f64"],
                    help="arithmetic type; f64 is the precision of the parity oracle examples/julia-raytracer.jl (default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the informational run in the other precision")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="all ranks use GPU 0 and the gloo backend (not a measurement)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from spira_hip import _binding as B
    from spira_hip import distributed as D
    from spira_hip import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    if args.rehearse_on_one_gpu:      # N ranks share GPU 0 over gloo: exercises the N>1 code path on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    B.set_device(local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    W, H, depth = args.width, args.height, args.depth
    spp_total = args.spp * world
    s = {"s1": scenes.scene_s1, "s2": scenes.scene_s2, "s3": scenes.scene_s3, "s4": scenes.scene_s4}[args.scene]()
    sc = (s["spheres5"], s["materials8"], s["triangles10"], s["camera12"])
    ns, nm = len(s["spheres5"]), len(s["materials8"])
    nt = 0 if s["triangles10"] is None else len(s["triangles10"])
    kflag = B.KERNEL_MEGA if args.kernel == "mega" else B.KERNEL_WAVEFRONT
    tile = D.tile_params(H, world, rank)
    rows = tile["rows"] or H
    params = B.make_params(W, H, spp_total, depth, ns, nm, nt, flags=kflag | B.POST_NONE, seed=scenes.seed_for(3), **tile)
    tdt = torch.float32 if args.prec == "f32" else torch.float64
    out = torch.empty((3, rows, W), dtype=tdt, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        B.render_device(*sc, params, out.data_ptr(), 0, stream.cuda_stream, args.prec)
        return D.gather_image(out, H) if world > 1 else out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:       # connection setup only (no rendering): RCCL opens its point-to-point channels at the first gather
        out.zero_()
        D.gather_image(out, H)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    c_timed = B.counters()
    samples_per_step = W * H * spp_total
    value = samples_per_step * args.steps / dt / 1e6

    result = None
    if rank == 0:
        assert bool(torch.isfinite(img).all()), "non-finite pixels"
        # ---- roofline leg: one extra render with every bounce launch bracketed by HIP events
        def roofline_leg(prec, out_t):
            pp = B.make_params(W, H, spp_total, depth, ns, nm, nt, flags=kflag | B.POST_NONE | B.FLAG_PROFILE,
                               seed=scenes.seed_for(3), **tile)
            B.render_device(*sc, pp, out_t.data_ptr(), 0, stream.cuda_stream, prec)
            torch.cuda.synchronize()
            c = B.counters()
            nbytes = algorithmic_bytes(c, 4 if prec == "f32" else 8)
            launches = max(1, c["bounce_launches"])
            avg_ms = c["bounce_kernel_ms"] / launches
            achieved = nbytes / (c["bounce_kernel_ms"] * 1e-3) / 1e9 if c["bounce_kernel_ms"] > 0 else 0.0
            traffic, tsrc = None, None
            tfile = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (args.scene, prec))
            if os.path.exists(tfile) and (W, H, spp_total, depth, world) == (1920, 1080, 64, 8, 1):
                tj = json.load(open(tfile))      # PMC bytes per k_bounce launch of this same command (profiles/run_profile.sh)
                traffic, tsrc = round(tj["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(tfile)
            return {"bound": "hbm", "kernel": "k_bounce", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                    "bytes_per_launch": round(nbytes / launches), "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                    "bytes_per_sample": round(nbytes / c["samples"], 2), "segments_per_sample": round(c["segments"] / c["samples"], 4),
                    "bounce_kernel_share": round(c["bounce_kernel_ms"] / max(c["kernel_ms"], 1e-9), 4)}

        roof = roofline_leg(args.prec, out) if args.kernel == "wavefront" else None
        # ---- the same workload in the other precision (N=1 only; informational, never `value`)
        alt = None
        if world == 1 and not args.no_alt_precision:
            ap_ = "f32" if args.prec == "f64" else "f64"
            out2 = torch.empty((3, rows, W), dtype=torch.float32 if ap_ == "f32" else torch.float64, device="cuda")
            B.render_device(*sc, params, out2.data_ptr(), 0, stream.cuda_stream, ap_)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                B.render_device(*sc, params, out2.data_ptr(), 0, stream.cuda_stream, ap_)
            torch.cuda.synchronize()
            adt = (time.perf_counter() - t1) / 3
            aroof = roofline_leg(ap_, out2) if args.kernel == "wavefront" else None
            alt = {"dtype": ap_, "value": round(samples_per_step / adt / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(adt * 1e3, 3),
                   "roofline_frac": aroof and aroof["frac"], "achieved_GBps": aroof and aroof["achieved"],
                   "bytes_per_sample": aroof and aroof["bytes_per_sample"]}
            del out2
        # ---- CPU baseline leg (rank 0, N=1 only): the oracle port on the host cores, bounded sample
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import oracle_py as O
            cores = min(O.max_threads(), host_cpu_share())     # the box's cgroup CPU share, not the host's core count
            O.render(*sc, O.make_params(W, H, 1, depth, ns, nm, nt, seed=scenes.seed_for(3), rows=8), "f64", n_threads=cores)   # spin up the threads
            t1 = time.perf_counter()
            O.render(*sc, O.make_params(W, H, 2, depth, ns, nm, nt, seed=scenes.seed_for(3)), "f64", n_threads=cores)
            cal = time.perf_counter() - t1
            rate = 2 * W * H / max(cal, 1e-6)                    # samples/s from a whole-frame spp=2 calibration
            cpu_spp = max(1, min(512, int(rate * args.cpu_seconds / (W * H))))   # short runs read fast: cap the sample
            t1 = time.perf_counter()
            O.render(*sc, O.make_params(W, H, cpu_spp, depth, ns, nm, nt, seed=scenes.seed_for(3)), "f64", n_threads=cores)
            cdt = time.perf_counter() - t1
            # the reference's own loop is serial (examples/julia-raytracer.jl:392): one thread, the middle 64 rows at spp 16
            t1 = time.perf_counter()
            O.render(*sc, O.make_params(W, H, 16, depth, ns, nm, nt, seed=scenes.seed_for(3), row0=H // 2 - 32, rows=64), "f64", n_threads=1)
            sdt = time.perf_counter() - t1
            cpu = {"value": round(W * H * cpu_spp / cdt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                   "sample": "%dx%d spp=%d depth=%d, same scene/seed, Float64 oracle (oracle/spira_oracle.c, OpenMP over rows), %.1f s"
                             % (W, H, cpu_spp, depth, cdt),
                   "single_thread_value": round(W * 64 * 16 / sdt / 1e6, 4),
                   "single_thread_sample": "rows %d..%d of the same frame at spp=16, 1 thread, %.1f s" % (H // 2 - 32, H // 2 + 31, sdt)}
        result = {
            "metric": "Msamples/sec at 1920x1080 spp=64 depth=8; fraction of HBM roofline",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %dx%d spp=%d depth=%d, scene %s (%s), semantics A, %s kernels, tile-sharded "
                                   "over %d GPU(s) in 8-row stripes + one RCCL gather" %
                                   (W, H, spp_total, depth, args.scene,
                                    {"s1": "create_scene() of src/spira-metal-optimized.jl", "s2": "create_scene() of examples/julia-raytracer.jl",
                                     "s3": "S1 inside a closed box", "s4": "create_scene_with_obj() of examples/julia-raytracer.jl with an 81 920-triangle procedural mesh (BVH)"}[args.scene], args.kernel, world),
                       "width": W, "height": H, "spp": spp_total, "max_depth": depth, "scene": args.scene, "kernel": args.kernel,
                       "samples_per_step": samples_per_step, "segments_per_step_rank0": c_timed["segments"],
                       "passes_per_step": c_timed["passes"], "launches_per_step": c_timed["launches"]},
            "roofline": roof, "cpu_baseline": cpu, "other_precision": alt,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
