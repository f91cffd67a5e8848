#!/usr/bin/env python3
"""Summarise a profiles/run_profile.sh output directory into one markdown file.

usage: python profiles/summarize.py gpurun_out/prof_<tag> profiles/<name>.md
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is given both
raw and doubled (the doubled figure applies to the 16-byte packet loads of the ray queues).
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# SIMD cycles (at a nominal 2.4 GHz, i.e. a time) one wave64 VALU instruction of a class holds the issue port, measured on MI355X with
# profiles/microbench/valu_peak.hip (inline-assembly streams of independent instructions, 4 waves per SIMD; profiles/r02_valu_peak.txt).  "other" = what the class
# counters do not cover: moves, compares, v_cndmask, the v_div_scale / v_div_fmas / v_div_fixup helpers, v_ldexp, readlane ...
MICROBENCH_CLOCK_GHZ = 2.3
VALU_COST = {"SQ_INSTS_VALU_FMA_F32": 3.04, "SQ_INSTS_VALU_MUL_F32": 2.72, "SQ_INSTS_VALU_ADD_F32": 2.66, "SQ_INSTS_VALU_TRANS_F32": 8.3,
             "SQ_INSTS_VALU_INT32": 3.0,      # the integer work is the hash RNG: per mix32 two v_mul_lo_u32 (4.66) and six shifts / xors (2.5)
             "SQ_INSTS_VALU_CVT": 4.3, "SQ_INSTS_VALU_FMA_F64": 5.37, "SQ_INSTS_VALU_MUL_F64": 4.92, "SQ_INSTS_VALU_ADD_F64": 4.67,
             "SQ_INSTS_VALU_TRANS_F64": 16.4, "SQ_INSTS_VALU_INT64": 4.5,
             # not covered by a class counter: in a Float32 kernel v_mov_b32 2.6, v_cmp 4.3, v_cndmask 4.6, v_div_scale/fmas/fixup 4.4;
             # in a Float64 kernel v_mov_b64 4.4, v_cmp_f64 4.7, the f64 division helpers 4.8, v_ldexp_f64 4.6
             "other_f32": 3.9, "other_f64": 4.5}


def short(name):
    n = name.replace("void ", "").replace("spira::", "")
    return n.split("(")[0][:60]


def is_spec(name):
    """k_path<T, R, BVH, EXT, SPEC, MODE, TRI>: is this the speculative-division instantiation (the first launch of a pass)?"""
    n = short(name)
    if not n.startswith("k_path<"):
        return False
    args = [a.strip() for a in n[n.index("<") + 1:n.rindex(">")].split(",")]
    return len(args) >= 5 and args[4] == "true"


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = ["# rocprofv3 summary: %s" % os.path.basename(src.rstrip("/")), ""]
    tlog = os.path.join(src, "trace.log")
    if os.path.exists(tlog):
        for ln in open(tlog):
            if ln.startswith("{"):
                lines += ["bench line of the traced run (its `roofline.traffic*` fields look at the summary committed BEFORE this run — the PMC passes below are this "
                          "run's own and become `profiles/traffic_*.json` for the next one):", "", "```", ln.strip(), "```", ""]
    st = os.path.join(src, "trace", "trace_kernel_stats.csv")
    if os.path.exists(st):
        lines += ["## --kernel-trace --stats", "", "| kernel | calls | total ms | avg us | % | min us | max us |", "|---|---:|---:|---:|---:|---:|---:|"]
        tot_b = cnt_b = spec_calls = 0
        path_calls = []
        for r in csv.DictReader(open(st)):
            if float(r["Percentage"]) < 0.05:
                continue
            lines.append("| %s | %s | %.3f | %.2f | %s | %.2f | %.2f |" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                                           float(r["AverageNs"]) / 1e3, r["Percentage"], int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
            if "k_bounce" in r["Name"] or "k_path" in r["Name"]:
                tot_b += int(r["TotalDurationNs"])
                cnt_b += int(r["Calls"])
                if "k_path" in r["Name"]:
                    path_calls.append(int(r["Calls"]))
                    if is_spec(r["Name"]):
                        spec_calls += int(r["Calls"])
        if path_calls:
            # one k_path "launch" (= one pass) is the speculative-division instantiation <..., true>, the exact one <..., false> behind it, which
            # renders again the waves the first reported (normally none: its workgroups return at once) and, on mesh scenes, the exact
            # instantiation once more for the fat waves that walk the tree; HIP events in bench.py bracket all of them
            cnt_b = spec_calls if spec_calls else max(path_calls)
        if cnt_b:
            lines += ["", "dominant kernel (k_path: speculative launch + its exact follow-up counted as one; k_bounce: all instantiations): %d launches, average %.2f us" %
                      (cnt_b, tot_b / cnt_b / 1e3), ""]
    kt = os.path.join(src, "trace", "trace_kernel_trace.csv")
    if os.path.exists(kt):       # launch by launch: the first launches of a process are slower than the steady state bench.py times
        d = []
        rows = sorted((r for r in csv.DictReader(open(kt)) if "k_path" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        had_spec = False
        for r in rows:
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            spec = is_spec(r["Kernel_Name"])
            if d and not spec and had_spec:
                d[-1] += us                   # the exact launches behind a speculative one belong to its pass (follow-up; mesh scenes: the fat waves)
            else:
                d.append(us)
            had_spec = had_spec or spec
        if len(d) >= 4:
            half = d[len(d) // 2:]
            lines += ["k_path launch by launch (us): " + ", ".join("%.0f" % x for x in d),
                      "steady state (mean of the last %d launches): %.2f us — compare `roofline.avg_launch_ms` of the bench line above (HIP events around the last timed step's launch)" %
                      (len(half), sum(half) / len(half)), ""]
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(float)
    pass_dur = defaultdict(lambda: defaultdict(float))     # per pass directory: summed dispatch durations (ns) per kernel
    owner = {}
    for d in sorted(os.listdir(src), key=lambda x: (x != "pmc_sq1", x)):      # pmc_sq1 first: it holds SQ_INSTS_VALU together with GRBM_GUI_ACTIVE
        f = os.path.join(src, d, "pmc_counter_collection.csv")
        if not d.startswith("pmc_") or not os.path.exists(f):
            continue
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if owner.setdefault((k, r["Counter_Name"]), d) != d:
                continue                      # a counter collected by several passes is taken from the first one only
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                pass_dur[d][k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                if d == "pmc_fetch":
                    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines += ["## --pmc passes (sums over all dispatches of the run; one pass per counter group)", ""]
    for k in sorted(agg, key=lambda x: -dur.get(x, 0)):
        if not k.startswith("k_"):
            continue
        c = agg[k]
        n = max(calls[k].values())
        lines.append("### %s  (%d dispatches)" % (k, n))
        lines.append("")
        for name in sorted(c):
            lines.append("- %s = %.6g" % (name, c[name]))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd, wr = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
            lines.append("- HBM read bytes: raw %.4g, x2-corrected (16 B/lane streams) %.4g; write bytes %.4g" % (rd, 2 * rd, wr))
            lines.append("- per dispatch: read %.4g (x2: %.4g) B, write %.4g B" % (rd / n, 2 * rd / n, wr / n))
            if dur.get(k):
                lines.append("- during the FETCH pass the kernel ran %.3f ms in total -> HBM rate (x2 read + write) %.1f GB/s" %
                             (dur[k] / 1e6, (2 * rd + wr) / dur[k]))
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
            lines.append("- VALU lane utilisation SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = %.3f" %
                         (c.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64 * c["SQ_ACTIVE_INST_VALU"])))
            lines.append("- VALU instructions per wave = %.1f" % (c["SQ_INSTS_VALU"] / max(1.0, c["SQ_WAVES"])))
        if c.get("SQ_INSTS_VALU") and c.get("GRBM_GUI_ACTIVE"):
            # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md); 1024 SIMDs
            lines.append("- SIMD cycles per VALU wave-instruction = 1024 x (GRBM_GUI_ACTIVE / 8) / SQ_INSTS_VALU = %.2f" %
                         (1024 * c["GRBM_GUI_ACTIVE"] / 8 / c["SQ_INSTS_VALU"]))
        if "SQ_WAIT_ANY" in c and "SQ_ACTIVE_INST_ANY" in c:
            tot = c["SQ_WAIT_ANY"] + c["SQ_WAIT_INST_ANY"] + c["SQ_ACTIVE_INST_ANY"]
            lines.append("- wave-cycle shares: WAIT_ANY %.2f, WAIT_INST_ANY %.2f, ACTIVE_INST_ANY %.2f" %
                         (c["SQ_WAIT_ANY"] / tot, c["SQ_WAIT_INST_ANY"] / tot, c["SQ_ACTIVE_INST_ANY"] / tot))
        if "TCC_HIT_sum" in c:
            lines.append("- L2 hit rate = %.3f" % (c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
        lines.append("")
    # machine-readable PMC figures of the dominant kernel (k_path, or all k_bounce instantiations), per launch, for bench.py's
    # roofline.traffic / roofline.valu fields
    for dom in ("k_path", "k_bounce"):
        ks = [k for k in agg if k.startswith(dom)]
        # launches: k_bounce instantiations are launches of their own; the k_path instantiations of one run are the speculative launch and
        # its exact follow-up, together one launch
        spec_ks = [k for k in ks if is_spec(k)]
        launches_of = ((lambda name: sum(calls[k].get(name, 0) for k in spec_ks) if spec_ks else max([calls[k].get(name, 0) for k in ks] or [0]))
                       if dom == "k_path" else (lambda name: sum(calls[k].get(name, 0) for k in ks)))
        nl = launches_of("FETCH_SIZE")
        if not nl:
            continue
        tot = lambda name: sum(agg[k].get(name, 0) for k in ks)
        rd, wr = tot("FETCH_SIZE") * 1024, tot("WRITE_SIZE") * 1024
        import bench
        out = {"kernel": dom, "launches": nl, "source_hash": bench.kernel_source_hash(), "fetch_bytes_raw": rd, "fetch_bytes_x2": 2 * rd, "write_bytes": wr,
               "hbm_bytes_per_launch": (2 * rd + wr) / nl,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB units); FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md HBM section (gfx950 reports half of a 16 B/lane streaming read)",
               "source": os.path.basename(src.rstrip("/"))}
        if tot("SQ_ACTIVE_INST_VALU") and tot("GRBM_GUI_ACTIVE"):
            w = tot("SQ_WAIT_ANY") + tot("SQ_WAIT_INST_ANY") + tot("SQ_ACTIVE_INST_ANY")
            # VALU-issue roofline: class counts x measured issue cost per class, against the SIMD cycles of the same dispatches
            # The microbenchmark prints times as "cycles at 2.4 GHz"; its dispatches held 2.14-2.43 GHz (GRBM_GUI_ACTIVE / 8 / duration,
            # profiles/r02_valu_peak.txt), 2.3 on average, so a real cycle count is 0.96 x the printed one.  The capacity is counted in
            # real cycles too: GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 per SIMD, whatever clock DVFS let the kernel hold.
            n_inst = tot("SQ_INSTS_VALU")
            wall_ns = sum(pass_dur["pmc_sq1"][k] for k in ks)        # the pass that counted SQ_INSTS_VALU and GRBM_GUI_ACTIVE
            simd_cycles = 1024 * tot("GRBM_GUI_ACTIVE") / 8
            # the class counters come from other passes of the same (deterministic) command: scale them to this pass's dispatch count
            n_disp = max(1, launches_of("SQ_INSTS_VALU"))
            mix, covered = {}, 0.0
            for cname, cost in VALU_COST.items():
                if cname.startswith("other"):
                    continue
                nd = launches_of(cname)
                if nd:
                    cnt = tot(cname) * n_disp / nd
                    mix[cname.replace("SQ_INSTS_VALU_", "").lower()] = round(cnt / n_inst, 4)
                    covered += cnt
            if mix:
                mix["other"] = round(max(0.0, n_inst - covered) / n_inst, 4)
            # MEASURED.  issue_slots = VALU wave-instructions x the 2 issue passes each needs at least (MI355X_MICROARCH.md) over the SIMD cycles
            # of the same dispatches: a hard-bounded (<= 1) share of the VALU issue port; Float64 and transcendental instructions hold the port
            # longer (profiles/r02_valu_peak.txt: 4.7-5.4 and 8-16 cycles), so it understates a Float64 kernel — simd_cycles_per_valu_inst next
            # to it says how far from 2 the kernel's mix can be at best.  valubusy_rocprof is rocprof's derived VALUBusy, 4 x SQ_ACTIVE_INST_VALU
            # / (SIMDs x cycles): on this chip SQ_ACTIVE_INST_VALU comes out at one quad-cycle per instruction (~= SQ_INSTS_VALU), i.e. the metric
            # charges 4 cycles to every instruction and reads ABOVE 1 on Float32 kernels (1.08-1.21) — listed for reference, not a ceiling.
            # (Round 2's priced model — instruction classes x microbenchmarked issue costs — is gone: a third of the instructions belong to no
            # class counter, and pricing them with one average put it above 1.)
            busy = 4.0 * tot("SQ_ACTIVE_INST_VALU") / simd_cycles
            # ---- the compute side as a fraction.  Two floors for the SIMD cycles per executed VALU instruction of THIS kernel's mix:
            #   spec   : the data-sheet issue cost of every class — 2 cycles for a 32-bit wave64 instruction (SIMD-32, MI355X_MICROARCH.md; = 157.3 TFLOP/s),
            #            4 for Float64 FMA / MUL / ADD and 64-bit integer (78.6 TFLOP/s), 8 / 16 for Float32 / Float64 transcendentals — over the DYNAMIC class
            #            counts of the PMC passes.  issue_util = spec floor / measured cycles per instruction: a fraction in (0, 1] by construction.
            #   priced : the same counts priced with the issue costs MEASURED on this chip (profiles/microbench/valu_peak.hip: a class's opcodes as the kernel's ISA
            #            holds them, profiles/isa_cost.json; an opcode costs the lower of its own stream and its marginal cost beside v_fma_f32).  What a perfectly
            #            scheduled, never-stalled wave mix of this kernel would need on the real issue port; issue_util_priced = priced / measured.
            # flops_frac: Float64 / Float32 (2 FMA + MUL + ADD) lane-operations (wave-instructions x 64 x the measured lane utilisation) over 78.6 / 157.3 TFLOP/s.
            SPEC = {"fma_f32": 2, "mul_f32": 2, "add_f32": 2, "trans_f32": 8, "int32": 2, "cvt": 2, "fma_f64": 4, "mul_f64": 4, "add_f64": 4, "trans_f64": 16, "int64": 4, "other": 2}
            lane_util = tot("SQ_THREAD_CYCLES_VALU") / max(1.0, 64 * tot("SQ_ACTIVE_INST_VALU"))
            compute = None
            if mix and "fma_f32" in mix:
                is_f64 = "double" in "".join(ks)
                # `other` in a Float64 kernel holds 64-bit moves / compares / min-max / the division helpers: priced as the kernel's ISA has them (isa_cost.json), spec floor 2
                isa = {}
                try:
                    isa_all = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "isa_cost.json")))["kernels"]
                    wsum = 0.0
                    for k in ks:                      # the instantiations of this run, weighted by their dynamic VALU counts
                        w_k = agg[k].get("SQ_INSTS_VALU", 0.0)
                        if k in isa_all and w_k:
                            for cls, cst in isa_all[k]["class_cost"].items():
                                isa[cls] = isa.get(cls, 0.0) + w_k * cst
                            wsum += w_k
                    isa = {c: v / wsum for c, v in isa.items()} if wsum else {}
                except (OSError, KeyError, ValueError):
                    isa = {}
                spec_c = sum(mix.get(c, 0.0) * SPEC[c] for c in SPEC)
                priced_c = sum(mix.get(c, 0.0) * isa.get(c, SPEC[c] * 1.5) for c in SPEC) if isa else None
                measured_c = simd_cycles / n_inst
                secs = wall_ns * 1e-9
                f64_flops = (2 * mix.get("fma_f64", 0) + mix.get("mul_f64", 0) + mix.get("add_f64", 0)) * n_inst * 64 * lane_util
                f32_flops = (2 * mix.get("fma_f32", 0) + mix.get("mul_f32", 0) + mix.get("add_f32", 0)) * n_inst * 64 * lane_util
                arith = sum(mix.get(c, 0.0) for c in ("fma_f32", "mul_f32", "add_f32", "trans_f32", "fma_f64", "mul_f64", "add_f64", "trans_f64"))
                compute = {"min_cycles_per_inst_spec": round(spec_c, 3), "issue_util": round(spec_c / measured_c, 4),
                           "min_cycles_per_inst_priced": round(priced_c, 3) if priced_c else None, "issue_util_priced": round(priced_c / measured_c, 4) if priced_c else None,
                           "flops_frac": round(f64_flops / secs / 78.6e12 + f32_flops / secs / 157.3e12, 4),
                           "tflops": {"f64": round(f64_flops / secs / 1e12, 3), "f32": round(f32_flops / secs / 1e12, 3)},
                           "non_arithmetic_share": round(1.0 - arith, 4), "float64_kernel": is_f64,
                           "what": "issue_util = (dynamic class counts x data-sheet issue cycles: 2 per 32-bit, 4 per 64-bit, 8 / 16 per transcendental wave64 instruction) / measured SIMD cycles "
                                   "per VALU instruction: the share of the VALU issue port this mix needs at best, in (0, 1]; issue_util_priced = the same with the issue costs measured on this "
                                   "chip (profiles/isa_cost.json, r04_valu_peak.txt); flops_frac = (2 FMA + MUL + ADD) lane-operations over 78.6 (f64) / 157.3 (f32) TFLOP/s; "
                                   "non_arithmetic_share = instructions that are no FMA / MUL / ADD / transcendental (integer, conversions, moves, compares, selects, division helpers)"}
            out["compute"] = compute
            out["valu"] = {"issue_slots": round(2.0 * n_inst / simd_cycles, 4), "valubusy_rocprof": round(busy, 4), "simd_cycles_per_valu_inst": round(simd_cycles / n_inst, 3),
                           "effective_clock_GHz": round(tot("GRBM_GUI_ACTIVE") / 8 / max(wall_ns, 1.0), 3),
                           "lane_utilisation": round(tot("SQ_THREAD_CYCLES_VALU") / max(1.0, 64 * tot("SQ_ACTIVE_INST_VALU")), 4),
                           "valu_insts_per_launch": tot("SQ_INSTS_VALU") / max(1, launches_of("SQ_INSTS_VALU")),
                           "wave_cycle_shares": {"wait_any": round(tot("SQ_WAIT_ANY") / w, 3), "wait_inst_any": round(tot("SQ_WAIT_INST_ANY") / w, 3),
                                                 "active_inst_any": round(tot("SQ_ACTIVE_INST_ANY") / w, 3)} if w else None,
                           "instruction_mix": mix or None}
        json.dump(out, open(os.path.splitext(dst)[0] + ".json", "w"), indent=1)
        break
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
