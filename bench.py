#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-trace hot path on MI355X (BASELINE.json metric).

A "step" is one whole render of the workload.  At N=1 the workload is BASELINE configs[2]:
1920x1080, spp=64, max_depth=8 on the reference's own scene (create_scene() of
src/spira-metal-optimized.jl:429-510 with main()'s camera :1499-1505), synthetic by construction.
At N>1 the frame is tile-sharded over the ranks (interleaved rows, one process per GPU,
no collective while rendering, ONE RCCL gather of the tiles per step) and spp = 64*N, so the
per-GPU work is fixed ("weak").  `--config c4` is BASELINE configs[3] instead: spp 256 in TOTAL at any
N ("strong"); `--config c5` is configs[4] (mesh scene, depth 12).  The scene is resident in HBM (a scene handle, created before the
timed region); a step passes the camera and the parameters; outputs stay in HBM.

Prints ONE JSON line on rank 0.
  roofline      the dominant kernel (k_path: one launch per pass) against the HBM roofline: algorithmic bytes from the
                device counters of the LAST TIMED step divided by that step's own kernel time (HIP events the library
                records around every k_path launch on the render stream) — so avg_launch_ms * launches <= ms_per_step.
                `valu` (from the committed PMC summary of the same command) is the VALU-issue side; `bound` names
                whichever fraction is higher.
  stress        the same frame on S3 (closed box: every path runs all 8 segments), N=1 only.
  organisations the round-1 per-bounce organisation and the megakernel on the same workload, N=1 only (informational).
  estimators    the reference's two other estimators on the same frame (SPIRA_SEM_METAL in wavefront and in one-lane-per-pixel
                form, SPIRA_SEM_CPU), N=1 only (informational).
  cpu_baseline  the CPU oracle (a port: the Julia reference cannot run here) on the host cores, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "julia-spira_amd"), os.path.join(ROOT, "oracle")]

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SCENE_DESC = {"s1": "create_scene() of src/spira-metal-optimized.jl", "s2": "create_scene() of examples/julia-raytracer.jl",
              "s3": "S1 inside a closed box", "s4": "create_scene_with_obj() of examples/julia-raytracer.jl with an 81 920-triangle procedural mesh (BVH)",
              "s5": "S4's objects seen from 16 cm in front of the mesh: it fills 70 % of the frame (mesh-path stress scene)",
              "s2g": "create_scene() of examples/julia-raytracer.jl with two glass spheres and a glass triangle (extension scene, parity unpinned)"}
CONFIGS = {   # BASELINE.json configs[2..4]: scene, spp, depth, spp is per GPU (weak) or in total (strong)
    "c3": dict(scene="s1", spp=64, depth=8, scaling="weak", name="BASELINE configs[2]"),
    "c4": dict(scene="s3", spp=256, depth=8, scaling="strong", name="BASELINE configs[3]"),      # BASELINE.md §3 / SURVEY §8d: c4 runs on S3
    "c5": dict(scene="s4", spp=64, depth=12, scaling="weak", name="BASELINE configs[4]"),
}


def algorithmic_bytes(c, prec_bytes, kernel):
    """HBM bytes the wavefront formulation must move (DESIGN.md "Roofline"), from device counters: every queued packet is
    written once and read once, a radiance term is a 3-value store, or a read-modify-write (2 x 3 values) when the path
    already holds radiance.  Packet: 10 values (+ a 4-byte hit reference in the Float32 hit queues of k_path).  Mesh scenes: a ray that
    reaches the mesh's box waits on its wave's mesh list for a traversal session — an entry of 12 values, written once and read once."""
    packet = 10 * prec_bytes + (4 if (kernel == "wavefront" and prec_bytes == 4) else 0)
    return (2 * packet * c["rays_enqueued"] + 2 * 12 * prec_bytes * c.get("rays_parked", 0)
            + (3 * c["radiance_stores"] + 6 * c["radiance_rmw"]) * prec_bytes)


def kernel_source_hash():
    """The id a library built NOW from the files on disk with the Makefile's default flags would carry (`make -s build_id`: sha256 over every file the
    library is compiled from, the effective flag strings of its three translation units and the compiler's version line; first 16 hex digits).
    profiles/traffic_*.json carry the id of the library they were measured on (profiles/summarize.py), and PMC figures of another build are never
    attached to a bench line."""
    import subprocess
    # (a clean environment for the child: under rocprofv3 the profiler's preloaded library would initialise the GPU in `make` and in everything make starts)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER", "HSA_TOOLS"))}
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "julia-spira_amd", "csrc"), "build_id"], check=True, capture_output=True, text=True, env=env).stdout
    return out.strip().splitlines()[-1].strip()


def metric_string(W, H, spp, depth):
    return "Msamples/sec at %dx%d spp=%d depth=%d; fraction of HBM roofline" % (W, H, spp, depth)


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


def roofline_record(c, prec, kernel, scene, is_headline_shape, source_hash=None):
    """Roofline of the dominant kernel from one render's counters (spira_get_counters).  `is_headline_shape`: the run has the launch
    shape the committed PMC summaries were taken on (1080p, 64 sample slots per pass, one GPU) — only then are they attached, and
    only when they were measured on the kernel sources of this run (`source_hash`; None = the sources as they are on disk)."""
    pb = 4 if prec == "f32" else 8
    nbytes = algorithmic_bytes(c, pb, kernel)
    launches = max(1, c["bounce_launches"])
    kms = c["bounce_kernel_ms"]
    achieved = nbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    rec = {"bound": "hbm", "kernel": "k_path" if kernel == "wavefront" else "k_bounce", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "traffic_source": None, "traffic_stale": False,
           "bytes_per_launch": round(nbytes / launches), "avg_launch_ms": round(kms / launches, 5), "launches": launches,
           "kernel_ms_per_step": round(kms, 4), "bytes_per_sample": round(nbytes / c["samples"], 2),
           "segments_per_sample": round(c["segments"] / c["samples"], 4), "packets_per_sample": round(c["rays_enqueued"] / c["samples"], 4),
           "parked_per_sample": round(c.get("rays_parked", 0) / c["samples"], 4),
           "kernel_share_of_step": round(kms / max(c["kernel_ms"], 1e-9), 4),
           # k_path: the speculative-division launch + the exact follow-up over the waves it reported; both inside avg_launch_ms
           "waves_rendered_again": int(c.get("redone_waves", 0)), "valu": None, "compute": None}
    if c.get("rays_parked", 0) and c.get("mesh_wave_trips", 0):
        # the traversal side of a mesh scene: rays that reach the mesh's box wait on a list and are walked through the 8-wide tree in sessions
        # (second, fat-wave launch of the pass); a trip = one memory round trip of a lane (a node visit, a triangle test, or both in Float32)
        wms = c.get("walk_kernel_ms", 0.0)
        rec["traversal"] = {"rays": int(c["rays_parked"]), "rays_per_sample": round(c["rays_parked"] / c["samples"], 4),
                            "walk_launch_ms": round(wms, 4), "walk_share_of_kernel": round(wms / kms, 4) if kms > 0 else None,
                            "Mrays_per_s": round(c["rays_parked"] / (wms * 1e-3) / 1e6, 1) if wms > 0 else None,
                            "trips_per_ray": round(c["mesh_lane_trips"] / c["rays_parked"], 3),
                            "walk_lane_utilisation": round(c["mesh_lane_trips"] / (64.0 * c["mesh_wave_trips"]), 4),
                            "what": "rays = entries written to the wave-owned mesh lists (every one is walked once); walk_launch_ms = device time of the pass's second "
                                    "launch (fat waves: sessions + the shading rounds of the paths they carry on), HIP events; Mrays_per_s = rays / walk_launch_ms"}
    tag = "%s_%s" % (scene, prec) if kernel == "wavefront" else "%s_%s_%s" % (kernel, scene, prec)
    tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % tag)
    if is_headline_shape and os.path.exists(tfile):      # PMC figures of this same command (profiles/run_profile.sh)
        tj = json.load(open(tfile))
        if source_hash is None:
            source_hash = kernel_source_hash()
        if tj.get("kernel") != rec["kernel"]:
            pass
        elif tj.get("source_hash") != source_hash:
            # measured on other kernel sources: nothing of it reaches the line, and no bound is claimed beyond what this run measured itself
            rec["traffic_stale"], rec["bound"] = True, None
            rec["traffic_source"] = "profiles/%s (stale: taken on sources %s, this run is %s)" % (os.path.basename(tfile), tj.get("source_hash"), source_hash)
        else:
            rec["traffic"], rec["traffic_source"] = round(tj["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(tfile)
            v = tj.get("valu")
            if v and v.get("issue_slots"):
                rec["valu"] = {"issue_slots": v["issue_slots"], "simd_cycles_per_valu_inst": v["simd_cycles_per_valu_inst"], "lane_utilisation": v["lane_utilisation"],
                               "wave_cycle_shares": v.get("wave_cycle_shares"), "valubusy_rocprof": v.get("valubusy_rocprof"),
                               "what": "measured by PMC on this same command (rocprofv3 --pmc passes, profiles/run_profile.sh): issue_slots = 2 issue passes x VALU "
                                       "wave-instructions / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) — a hard-bounded share of the VALU issue port that understates "
                                       "Float64 / transcendental instructions (they hold the port 4.7-16 cycles); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x "
                                       "SQ_ACTIVE_INST_VALU); valubusy_rocprof = rocprof's VALUBusy, which charges 4 cycles per instruction and exceeds 1 on Float32",
                               "source": rec["traffic_source"]}
                if v["issue_slots"] > rec["frac"]:
                    rec["bound"] = "valu"
            # the compute side as a fraction in (0, 1]: the kernel's dynamic instruction mix at data-sheet issue costs over the measured SIMD cycles per instruction
            # (profiles/summarize.py); the bound is whichever side of the roofline is closer to its ceiling
            cp = tj.get("compute")
            if cp and cp.get("issue_util"):
                rec["compute"] = {k: cp.get(k) for k in ("issue_util", "issue_util_priced", "min_cycles_per_inst_spec", "min_cycles_per_inst_priced", "flops_frac", "tflops",
                                                         "non_arithmetic_share", "what")}
                rec["bound"] = "valu" if cp["issue_util"] > rec["frac"] else "hbm"
    return rec


def first_call_probe(prec):
    """Run in a FRESH process (bench.py --first-call-probe f64): what a caller of the reference's only test pays — tests/bunny-test.jl:37-60 builds the mesh scene
    and renders it ONCE at 64x64, spp 1, depth 25 (render_example's depth, examples/julia-raytracer.jl:717) through the host-array entry.  The first call holds
    everything: device context, validation, triangle hash, the host BVH build (spira_bvh.h, on the host cores), uploads, first kernel launches, the frame back."""
    import numpy as np
    from spira_hip import _binding as B
    from spira_hip import scenes
    s = scenes.scene_s4()
    npdt = np.float64 if prec == "f64" else np.float32
    arrs = [np.ascontiguousarray(s[k], dtype=npdt) for k in ("spheres5", "materials8", "triangles10", "camera12")]      # (conversion is the caller's, not timed)
    ns, nm, nt = len(arrs[0]), len(arrs[1]), len(arrs[2])
    B.lib()
    pp = B.make_params(64, 64, 1, 25, ns, nm, nt, seed=scenes.seed_for(5))
    out = {"prec": prec, "triangles": nt, "host_threads": host_cpu_share()}
    t0 = time.perf_counter()
    B.device_count(); B.set_device(0)
    hdr, _ = B.render(*arrs, pp, prec)
    out["first_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    assert np.isfinite(hdr).all()
    t0 = time.perf_counter()
    B.render(*arrs, pp, prec)
    out["second_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3)          # the tree is cached by the hash of the triangle bytes
    arrs[2] = arrs[2].copy(); arrs[2][0, 0] += npdt(1e-3)                        # another mesh: validation + hash + build + upload again, context warm
    t0 = time.perf_counter()
    B.render(*arrs, pp, prec)
    out["new_mesh_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        h = B.Scene(arrs[0], arrs[1], arrs[2], prec)
        ts.append((time.perf_counter() - t0) * 1e3)
        h.destroy()
    out["scene_create_ms"] = round(min(ts), 3)
    out["scene_create_ms_all"] = [round(t, 3) for t in ts]
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first-call-probe", default=None, choices=["f32", "f64"], help="internal: time the first call of a fresh process on the mesh scene, print JSON, exit")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps first: the first ~6 launches of a process run up to 20 %% slower (profiles/r02_s1_f64.md)")
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS), help="BASELINE.json configuration (c3 = configs[2], the headline)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=None, help="override the configuration's spp")
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--scene", default=None, choices=["s1", "s2", "s3", "s4", "s5", "s2g"])
    ap.add_argument("--ext", default=None, choices=["spectral", "dielectric", "both"], help="SPIRA_EXT_* extension flags for the timed workload (parity unpinned; profiles/)")
    ap.add_argument("--kernel", default="wavefront", choices=["wavefront", "bounce", "mega"],
                    help="wavefront = persistent hit-queue kernel (default); bounce = round-1 per-bounce launches; mega = one lane per path")
    ap.add_argument("--prec", default="f64", choices=["f32", "f64"],
                    help="arithmetic type; f64 is the precision of the parity oracle examples/julia-raytracer.jl (default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the informational run in the other precision")
    ap.add_argument("--no-extras", action="store_true", help="skip the stress scene and the other kernel organisations")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="all ranks use GPU 0 and the gloo backend (not a measurement)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    if args.first_call_probe:
        return first_call_probe(args.first_call_probe)

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

    cfg = CONFIGS[args.config]
    scene_name = args.scene or cfg["scene"]
    W, H = args.width, args.height
    depth = args.depth if args.depth is not None else cfg["depth"]
    spp_cfg = args.spp if args.spp is not None else cfg["spp"]
    spp_total = spp_cfg * world if cfg["scaling"] == "weak" else spp_cfg
    builders = {"s1": scenes.scene_s1, "s2": scenes.scene_s2, "s3": scenes.scene_s3, "s4": scenes.scene_s4, "s5": scenes.scene_s5, "s2g": scenes.scene_s2_glass}
    kflags = {"wavefront": B.KERNEL_WAVEFRONT, "bounce": B.KERNEL_BOUNCE, "mega": B.KERNEL_MEGA}
    seed = scenes.seed_for({"c3": 3, "c4": 4, "c5": 5}[args.config])
    tile = D.tile_params(H, world, rank)
    rows = tile["rows"] or H
    stream = torch.cuda.current_stream()
    # the committed PMC summaries were taken at 1080p with 64 sample slots per pass on one GPU: any spp that is a multiple of 64 launches the same shape
    pmc_shape = (W, H, world) == (1920, 1080, 1) and spp_total % 64 == 0 and depth == cfg["depth"]
    src_hash = kernel_source_hash()
    lib_id = B.build_id()          # the sources the LOADED library was built from: a library older than the files on disk gets no PMC figures either
    if lib_id != src_hash:
        src_hash = "library:" + lib_id

    def workload(name):
        s = builders[name]()
        sc = (s["spheres5"], s["materials8"], s["triangles10"], s["camera12"])
        ns, nm = len(s["spheres5"]), len(s["materials8"])
        nt = 0 if s["triangles10"] is None else len(s["triangles10"])
        return sc, (ns, nm, nt)

    sc, counts = workload(scene_name)
    ext_flags = {None: 0, "spectral": B.EXT_SPECTRAL, "dielectric": B.EXT_DIELECTRIC, "both": B.EXT_SPECTRAL | B.EXT_DIELECTRIC}[args.ext]
    params = B.make_params(W, H, spp_total, depth, *counts, flags=kflags[args.kernel] | B.POST_NONE | ext_flags, seed=seed, **tile)
    tdt = {"f32": torch.float32, "f64": torch.float64}
    out = torch.empty((3, rows, W), dtype=tdt[args.prec], device="cuda")

    # the scene is made resident once (spira_scene_create_*: validated, BVH built, uploaded), like every other input of the
    # timed region; a step passes the camera and the parameters only
    scene_h = B.Scene(sc[0], sc[1], sc[2], args.prec)

    step_events = []        # N > 1: (before render, after render, after gather) of every timed step, on the stream both are enqueued on

    def step(timed=False):
        if world == 1:
            scene_h.render_device(sc[3], params, out.data_ptr(), 0, stream.cuda_stream)
            return out
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timed else None
        if ev:
            ev[0].record(stream)
        scene_h.render_device(sc[3], params, out.data_ptr(), 0, stream.cuda_stream)
        if ev:
            ev[1].record(stream)
        img_ = D.gather_image(out, H)
        if ev:
            ev[2].record(stream)
            step_events.append(ev)
        return img_

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
        img = step(timed=True)
    fence()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # where a scaling loss comes from: the slowest and the fastest rank's render (load balance between tiles) and gather (the one exchange;
        # on the other ranks it includes waiting for the slowest renderer), per step, from events on this rank's stream — one all_reduce each
        r_ms = sum(e[0].elapsed_time(e[1]) for e in step_events) / max(1, len(step_events))
        g_ms = sum(e[1].elapsed_time(e[2]) for e in step_events) / max(1, len(step_events))
        hi = torch.tensor([r_ms, g_ms], dtype=torch.float64, device="cuda")
        lo = hi.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        per_rank = {"render_ms": {"max": round(float(hi[0]), 4), "min": round(float(lo[0]), 4)}, "gather_ms": {"max": round(float(hi[1]), 4), "min": round(float(lo[1]), 4)},
                    "rows_per_rank": {"max": int(max(D.tile_params(H, world, r)["rows"] for r in range(world))), "min": int(min(D.tile_params(H, world, r)["rows"] for r in range(world)))},
                    "what": "per step, over ranks: device time of the rank's tile render and of the gather that follows it (events on the rank's stream)"}
    c_timed = B.counters()          # device counters + event times of the LAST timed step on this rank
    samples_per_step = W * H * spp_total
    value = samples_per_step * args.steps / dt / 1e6

    def side_run(scene, kernel, prec, reps=5, sem=0, spp=None, dep=None, seed_=None, ext=0):
        """An extra, untimed-region measurement on rank 0: (Msamples/s, ms per step, roofline record).  ext: SPIRA_EXT_* flags."""
        sc2, counts2 = workload(scene)
        fl = kflags[kernel] | B.POST_NONE | sem | ext
        spp_ = spp_total if spp is None else spp
        dep_ = depth if dep is None else dep
        pp = B.make_params(W, H, spp_, dep_, *counts2, flags=fl, seed=seed if seed_ is None else seed_, **tile)
        o2 = torch.empty((3, rows, W), dtype=tdt[prec], device="cuda")
        with B.Scene(sc2[0], sc2[1], sc2[2], prec) as h2:
            for _ in range(3):                 # warm-up: a configuration's first launches run slower (profiles/r02_*.md)
                h2.render_device(sc2[3], pp, o2.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                h2.render_device(sc2[3], pp, o2.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize()
            adt = (time.perf_counter() - t1) / reps
        c = B.counters()
        roof = None
        if sem == B.SEM_METAL and kernel == "wavefront":
            # k_path_metal: packets of 10 values + {hit sphere, LCG state}; per sample and pixel one read-modify-write of the running
            # sum (4 values each way) and of the LCG state (4 B each way); the rare radiance terms parked in L
            pb = 4 if prec == "f32" else 8
            nbytes = 2 * (10 * pb + 8) * c["rays_enqueued"] + c["samples"] * (8 * pb + 8) + 6 * pb * c["radiance_rmw"]
            kms = max(c["bounce_kernel_ms"], 1e-9)
            roof = {"bound": "valu", "kernel": "k_path_metal", "achieved": round(nbytes / (kms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "bytes_per_sample": round(nbytes / c["samples"], 2),
                    "avg_launch_ms": round(kms, 4), "launches": 1, "packets_per_sample": round(c["rays_enqueued"] / c["samples"], 4),
                    "segments_per_sample": round(c["segments"] / c["samples"], 4), "traffic": None, "valu": None}
        elif sem:
            roof = {"segments_per_sample": round(c["segments"] / c["samples"], 4)}
        elif kernel == "wavefront":
            roof = roofline_record(c, prec, kernel, scene + ("_ext" if ext else ""), (W, H, world) == (1920, 1080, 1) and spp_ % 64 == 0 and dep_ == {"s4": 12, "s5": 12}.get(scene, 8), src_hash)
        elif kernel == "bounce":     # per-bounce launches are only bracketed on request (it slows the render): one extra, profiled render
            B.render_device(*sc2, B.make_params(W, H, spp_, dep_, *counts2, flags=fl | B.FLAG_PROFILE, seed=seed, **tile),
                            o2.data_ptr(), 0, stream.cuda_stream, prec)
            torch.cuda.synchronize()
            roof = roofline_record(B.counters(), prec, kernel, scene, (W, H, world) == (1920, 1080, 1) and spp_ % 64 == 0 and dep_ == {"s4": 12, "s5": 12}.get(scene, 8), src_hash)
            roof["note"] = "kernel time from a separate event-bracketed render (slower than the timed steps)"
        del o2
        return W * H * spp_ / adt / 1e6, adt * 1e3, roof

    result = None
    if rank == 0:
        assert bool(torch.isfinite(img).all()), "non-finite pixels"
        if args.kernel == "wavefront":
            roof = roofline_record(c_timed, args.prec, args.kernel, scene_name + ("_ext" if ext_flags else ""), pmc_shape, src_hash)
        elif args.kernel == "bounce":
            roof = side_run(scene_name, "bounce", args.prec, reps=1)[2]
        else:
            roof = None
        # ---- the same workload in the other precision (N=1 only; informational, never `value`)
        alt = None
        if world == 1 and not args.no_alt_precision:
            ap_ = "f32" if args.prec == "f64" else "f64"
            v, ms, aroof = side_run(scene_name, args.kernel, ap_)
            alt = {"dtype": ap_, "value": round(v, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 3), "roofline": aroof}
        # ---- stress scene and the other kernel organisations (N=1 only; informational)
        stress, stress_mesh, extensions, orgs, estimators = None, None, None, None, None
        if world == 1 and not args.no_extras:
            if scene_name != "s3" and args.config != "c5":
                v, ms, sroof = side_run("s3", args.kernel, args.prec)
                stress = {"scene": "s3", "what": "the same frame with the scene inside a closed box: every path runs all max_depth segments (SURVEY 8d)",
                          "dtype": args.prec, "value": round(v, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 3), "roofline": sroof}
            if (W, H) == (1920, 1080):
                # the stress scene of the MESH path: S4's mesh filling 70 % of the frame, depth 12 — nearly every segment walks the tree
                stress_mesh = {"scene": "s5", "what": "BASELINE configs[4]'s 81 920-triangle mesh seen from close up (70 % of the frame), spp 64, depth 12: the traversal-bound frame",
                               "metric": metric_string(W, H, 64, 12)}
                for pr in ("f64", "f32"):
                    v, ms, sroof = side_run("s5", "wavefront", pr, reps=3, spp=64, dep=12, seed_=scenes.seed_for(6))
                    stress_mesh[pr] = {"value": round(v, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 3), "roofline": sroof}
            if (W, H) == (1920, 1080):
                # the extensions north_star / configs[4] name ("spectral SPD", dielectric BSDF; README.md:10): no reference code exists for them (parity unpinned),
                # their own EXT = true instantiations of k_path.  configs[4] with SPIRA_EXT_SPECTRAL, and the glass scene with both extensions at the headline shape.
                extensions = {"what": "SPIRA_EXT_* extensions (include/spira_hip.h; the reference only names them: parity unpinned): c5_spectral = BASELINE configs[4] with hero-wavelength "
                                      "spectral transport (SPD tables in LDS); s2_glass = examples/julia-raytracer.jl's scene with two glass spheres and a glass triangle, dielectric + spectral, "
                                      "1080p spp 64 depth 8; vs_plain = frame time over the same frame without the extension flags"}
                for key, scn, spp_e, dep_e, fl_e, seed_e in (("c5_spectral", "s4", 64, 12, B.EXT_SPECTRAL, scenes.seed_for(5)),
                                                          ("s2_glass", "s2g", 64, 8, B.EXT_DIELECTRIC | B.EXT_SPECTRAL, scenes.seed_for(7))):
                    entry = {"workload": "%dx%d spp=%d depth=%d, scene %s (%s), flags 0x%x" % (W, H, spp_e, dep_e, scn, SCENE_DESC[scn], fl_e), "metric": metric_string(W, H, spp_e, dep_e)}
                    for pr in ("f64", "f32"):
                        v, ms, eroof = side_run(scn, "wavefront", pr, reps=3, spp=spp_e, dep=dep_e, seed_=seed_e, ext=fl_e)
                        v0, ms0, _ = side_run(scn, "wavefront", pr, reps=3, spp=spp_e, dep=dep_e, seed_=seed_e)
                        entry[pr] = {"value": round(v, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 3), "plain_ms_per_step": round(ms0, 3), "vs_plain": round(ms / ms0, 4), "roofline": eroof}
                    extensions[key] = entry
            orgs = {}
            for k in ("wavefront", "bounce", "mega"):
                if k != args.kernel:
                    v, ms, oroof = side_run(scene_name, k, args.prec)
                    orgs[k] = {"value": round(v, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 3), "dtype": args.prec,
                               "hbm_frac": oroof and oroof["frac"], "bytes_per_sample": oroof and oroof["bytes_per_sample"],
                               "avg_launch_ms": oroof and oroof["avg_launch_ms"], "launches": oroof and oroof["launches"]}
            if scene_name in ("s1",) and args.config in ("c3", "c4"):        # the reference's other estimators on the same frame (sphere scenes only)
                estimators = {}
                for name, sem, k in (("metal_wavefront", B.SEM_METAL, "wavefront"), ("metal_one_lane_per_pixel", B.SEM_METAL, "mega"), ("cpu_one_lane_per_path", B.SEM_CPU, "mega"),
                                     ("hybrid_as_written", B.SEM_HYBRID, "mega")):      # render_hybrid_gpu's own estimator: (max_depth + 1) launches per sample, a fidelity path
                    v, ms, eroof = side_run(scene_name, k, args.prec, sem=sem)
                    estimators[name] = {"value": round(v, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 3), "dtype": args.prec, "roofline": eroof}
        # ---- the other BASELINE configurations that fit one GPU (N=1 only): configs[3] = 1080p spp 256 depth 8 on S3 (4 passes of 64 sample slots; on
        # 8 GPUs every rank renders its 135-row share of this), configs[4] = the 81 920-triangle mesh scene S4 at depth 12 — both precisions each
        other_configs = None
        if world == 1 and not args.no_extras and (W, H) == (1920, 1080):
            other_configs = {}
            for cname in ("c4", "c5"):
                if cname == args.config:
                    continue
                cc = CONFIGS[cname]
                entry = {"workload": "%s: %dx%d spp=%d depth=%d, scene %s (%s)" % (cc["name"], W, H, cc["spp"], cc["depth"], cc["scene"], SCENE_DESC[cc["scene"]]),
                         "metric": metric_string(W, H, cc["spp"], cc["depth"])}
                for pr in ("f64", "f32"):
                    v, ms, croof = side_run(cc["scene"], "wavefront", pr, reps=3, spp=cc["spp"], dep=cc["depth"], seed_=scenes.seed_for({"c4": 4, "c5": 5}[cname]))
                    entry[pr] = {"value": round(v, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 3), "roofline": croof}
                other_configs[cname] = entry
        # ---- end to end through the host-buffer entry point (N=1 only; never `value`): scene arrays in, image out to host memory — validation,
        # upload, the same kernels, one device-to-host copy of the frame (SURVEY §8d asks for it beside the device-resident figure)
        end_to_end = None
        if world == 1 and not args.no_extras:
            hdr_host, _ = B.render(sc[0], sc[1], sc[2], sc[3], params, args.prec)      # warm-up (pageable host memory, like a caller's)
            t1 = time.perf_counter()
            for _ in range(3):
                hdr_host, _ = B.render(sc[0], sc[1], sc[2], sc[3], params, args.prec)
            e2e = (time.perf_counter() - t1) / 3
            end_to_end = {"ms": round(e2e * 1e3, 3), "value": round(samples_per_step / e2e / 1e6, 3), "unit": "Msamples/s", "d2h_bytes": int(hdr_host.nbytes),
                          "what": "spira_render_%s with host pointers: scene validation + upload, kernels, the planar frame to freshly allocated pageable host memory (rendered as row slabs: a slab goes device -> pinned staging -> host threads -> the caller's pages beside the next slab's kernels), synchronous; the loop also pays the allocator for dropping the previous 50 MB array" % args.prec}
            # the call shape of SPIRA.jl's render(scene, camera, W, H) (src/spira-metal-optimized.jl:1453-1490 returns Matrix{RGB{Float32}}): Float32, the display image
            # only (out_hdr = NULL), ACES + gamma: 24.9 MB to the host
            s32 = [np.ascontiguousarray(a, dtype=np.float32) if a is not None else None for a in sc]
            p32 = B.make_params(W, H, spp_total, depth, *counts, flags=kflags[args.kernel] | B.POST_ACES_GAMMA, seed=seed, **tile)
            B.render(*s32, p32, "f32", want_hdr=False, want_img=True)
            t1 = time.perf_counter()
            for _ in range(3):
                _, img_host = B.render(*s32, p32, "f32", want_hdr=False, want_img=True)
            e32 = (time.perf_counter() - t1) / 3
            c32 = B.counters()
            end_to_end["f32_img_only_ms"] = round(e32 * 1e3, 3)
            end_to_end["f32_img_only_device_ms"] = round(c32["kernel_ms"], 3)
            end_to_end["f32_img_only_d2h_bytes"] = int(img_host.nbytes)
            if (W, H) == (1920, 1080):
                # the call shape of the oracle script's render(world, camera, W, H) with a mesh (julia/Raytracer.jl; configs[4]): Float64 host arrays; the first call
                # of a mesh validates, hashes and builds its tree on the host (cached afterwards by the hash of the triangle bytes)
                m = builders["s4"]()
                m64 = [np.ascontiguousarray(m[k], dtype=np.float64) for k in ("spheres5", "materials8", "triangles10", "camera12")]
                pm = B.make_params(W, H, 64, 12, len(m64[0]), len(m64[1]), len(m64[2]), flags=B.POST_NONE, seed=scenes.seed_for(5))
                t1 = time.perf_counter()
                B.render(*m64, pm, "f64")
                end_to_end["c5_host_arrays_first_ms"] = round((time.perf_counter() - t1) * 1e3, 3)
                t1 = time.perf_counter()
                for _ in range(3):
                    B.render(*m64, pm, "f64")
                end_to_end["c5_host_arrays_steady_ms"] = round((time.perf_counter() - t1) / 3 * 1e3, 3)
                end_to_end["c5_host_arrays_device_ms"] = round(B.counters()["kernel_ms"], 3)
        # ---- scene build latency (N=1 only): the host BVH build + uploads of the configs[4] mesh, and the first call of a fresh process in the shape of the
        # reference's own test (tests/bunny-test.jl:37-60: 64x64, spp 1, depth 25 — one render of a new mesh); each precision in its own fresh process
        scene_build = None
        if world == 1 and not args.no_extras:
            import subprocess
            scene_build = {"what": "fresh process per precision (bench.py --first-call-probe): first_call_ms = spira_render_* with host arrays on the 81 920-triangle scene at "
                                   "64x64 spp 1 depth 25, everything included (context, validation, hash, host BVH build on host_threads cores, uploads, first launches, frame back); "
                                   "second_call_ms = the same call again (tree cached); new_mesh_call_ms = another mesh in the warm context; scene_create_ms = spira_scene_create_* "
                                   "(validate + build + upload), best of 5"}
            for pr in ("f64", "f32"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--first-call-probe", pr], capture_output=True, text=True, timeout=300)
                try:
                    scene_build[pr] = json.loads(r.stdout.strip().splitlines()[-1])
                except (IndexError, ValueError):
                    scene_build[pr] = {"error": (r.stderr or r.stdout)[-400:]}
        # ---- what a rank's render of an 8-GPU run costs (N=1 only, the headline shape): this GPU renders each of the 8 ranks' tiles of the weak-scaling
        # workload (spp 64 x 8 on rows r, r + 8, ...) in turn; the step of the 8-GPU run costs the slowest rank's render plus the exchange (not measurable here)
        tiles8 = None
        if world == 1 and not args.no_extras and (W, H) == (1920, 1080) and cfg["scaling"] == "weak":
            ms8 = []
            for r8 in range(8):
                t8 = D.tile_params(H, 8, r8)
                p8 = B.make_params(W, H, spp_cfg * 8, depth, *counts, flags=kflags[args.kernel] | B.POST_NONE | ext_flags, seed=seed, **t8)
                o8 = torch.empty((3, t8["rows"], W), dtype=tdt[args.prec], device="cuda")
                for _ in range(3):                 # (a configuration's first launches run slower)
                    scene_h.render_device(sc[3], p8, o8.data_ptr(), 0, stream.cuda_stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    scene_h.render_device(sc[3], p8, o8.data_ptr(), 0, stream.cuda_stream)
                torch.cuda.synchronize()
                ms8.append((time.perf_counter() - t1) / 5 * 1e3)
            tiles8 = {"world": 8, "spp": spp_cfg * 8, "rows_per_rank": sorted(set(D.tile_params(H, 8, r8)["rows"] for r8 in range(8))),
                      "tile_render_ms": {"max": round(max(ms8), 3), "min": round(min(ms8), 3)}, "frame_ms_one_gpu": round(dt / args.steps * 1e3, 3),
                      "render_bound_of_weak_scaling_efficiency": round(dt / args.steps * 1e3 / max(ms8), 4),
                      "what": "each of the 8 ranks' tiles of `bench.py --gpus 8` rendered on this one GPU in turn (same samples per rank as the one-GPU frame): "
                              "frame_ms / slowest tile bounds the weak-scaling efficiency from the render side; the gather (6.2 MB per rank over xGMI) comes on top"}
        # ---- CPU baseline leg (rank 0, N=1 only): the oracle port on the host cores, bounded sample
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import oracle_py as O
            ns, nm, nt = counts
            cores = min(O.max_threads(), host_cpu_share())     # the box's cgroup CPU share, not the host's core count
            O.render(*sc, O.make_params(W, H, 1, depth, ns, nm, nt, seed=seed, rows=8), "f64", n_threads=cores)   # spin up the threads
            cal_rows = H if nt <= 32 else 16                    # the oracle scans every triangle: calibrate a mesh scene on a few rows
            r0 = (H - cal_rows) // 2
            t1 = time.perf_counter()
            O.render(*sc, O.make_params(W, H, 2, depth, ns, nm, nt, seed=seed, row0=r0, rows=cal_rows), "f64", n_threads=cores)
            cal = time.perf_counter() - t1
            rate = 2 * W * cal_rows / max(cal, 1e-6)             # samples/s from the calibration
            cpu_spp = max(1, min(512, int(rate * args.cpu_seconds / (W * cal_rows))))   # short runs read fast: cap the sample
            t1 = time.perf_counter()
            O.render(*sc, O.make_params(W, H, cpu_spp, depth, ns, nm, nt, seed=seed, row0=r0, rows=cal_rows), "f64", n_threads=cores)
            cdt = time.perf_counter() - t1
            cpu = {"value": round(W * cal_rows * cpu_spp / cdt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                   "sample": "%dx%d (rows %d..%d) spp=%d depth=%d, same scene/seed, Float64 oracle (oracle/spira_oracle.c, OpenMP over rows), %.1f s"
                             % (W, H, r0, r0 + cal_rows - 1, cpu_spp, depth, cdt)}
            if nt <= 32:
                # the reference's own loop is serial (examples/julia-raytracer.jl:392): one thread, the middle 64 rows at spp 16
                t1 = time.perf_counter()
                O.render(*sc, O.make_params(W, H, 16, depth, ns, nm, nt, seed=seed, row0=H // 2 - 32, rows=64), "f64", n_threads=1)
                sdt = time.perf_counter() - t1
                cpu["single_thread_value"] = round(W * 64 * 16 / sdt / 1e6, 4)
                cpu["single_thread_sample"] = "rows %d..%d of the same frame at spp=16, 1 thread, %.1f s" % (H // 2 - 32, H // 2 + 31, sdt)
        result = {
            "metric": metric_string(W, H, spp_total, depth),
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
            "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "%s: %dx%d spp=%d depth=%d, scene %s (%s), semantics A, %s kernel organisation, tile-sharded "
                                   "over %d GPU(s), rows dealt round-robin + one RCCL gather" %
                                   (cfg["name"], W, H, spp_total, depth, scene_name, SCENE_DESC[scene_name], args.kernel, world),
                       "width": W, "height": H, "spp": spp_total, "max_depth": depth, "scene": scene_name, "kernel": args.kernel,
                       "samples_per_step": samples_per_step, "segments_per_step_rank0": c_timed["segments"],
                       "passes_per_step": c_timed["passes"], "launches_per_step": c_timed["launches"]},
            "roofline": roof, "cpu_baseline": cpu, "end_to_end": end_to_end, "scene_build": scene_build, "tiles_of_8_gpus": tiles8, "configs": other_configs, "other_precision": alt, "stress": stress, "stress_mesh": stress_mesh, "extensions": extensions,
            "organisations": orgs, "estimators": estimators, "per_rank": per_rank, "kernel_source_hash": src_hash,
        }
        print(json.dumps(result), flush=True)
    scene_h.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
