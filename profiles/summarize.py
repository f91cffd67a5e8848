#!/usr/bin/env python3
"""Summarise a profiles/run_profile.sh output directory into one markdown file.

usage: python profiles/summarize.py gpurun_out/prof_<tag> profiles/<name>.md
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is given both
raw and doubled (the doubled figure applies to the 16-byte packet loads of the ray queues).
"""
import csv
import os
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void ", "").replace("spira::", "")
    return n.split("(")[0][:60]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = ["# rocprofv3 summary: %s" % os.path.basename(src.rstrip("/")), ""]
    tlog = os.path.join(src, "trace.log")
    if os.path.exists(tlog):
        for ln in open(tlog):
            if ln.startswith("{"):
                lines += ["bench line of the traced run:", "", "```", ln.strip(), "```", ""]
    st = os.path.join(src, "trace", "trace_kernel_stats.csv")
    if os.path.exists(st):
        lines += ["## --kernel-trace --stats", "", "| kernel | calls | total ms | avg us | % | min us | max us |", "|---|---:|---:|---:|---:|---:|---:|"]
        tot_b = cnt_b = 0
        for r in csv.DictReader(open(st)):
            if float(r["Percentage"]) < 0.05:
                continue
            lines.append("| %s | %s | %.3f | %.2f | %s | %.2f | %.2f |" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                                           float(r["AverageNs"]) / 1e3, r["Percentage"], int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
            if "k_bounce" in r["Name"] or "k_path" in r["Name"]:
                tot_b += int(r["TotalDurationNs"])
                cnt_b += int(r["Calls"])
        if cnt_b:
            lines += ["", "dominant kernel (k_path / k_bounce, all instantiations): %d launches, average %.2f us" % (cnt_b, tot_b / cnt_b / 1e3), ""]
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(float)
    for d in sorted(os.listdir(src)):
        f = os.path.join(src, d, "pmc_counter_collection.csv")
        if not d.startswith("pmc_") or not os.path.exists(f):
            continue
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
            if d == "pmc_fetch" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
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
        if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
            # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs; SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)
            lines.append("- VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = %.3f" %
                         (c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * c["GRBM_GUI_ACTIVE"] / 8)))
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
        nl = sum(calls[k].get("FETCH_SIZE", 0) for k in ks)
        if not nl:
            continue
        import json
        tot = lambda name: sum(agg[k].get(name, 0) for k in ks)
        rd, wr = tot("FETCH_SIZE") * 1024, tot("WRITE_SIZE") * 1024
        out = {"kernel": dom, "launches": nl, "fetch_bytes_raw": rd, "fetch_bytes_x2": 2 * rd, "write_bytes": wr,
               "hbm_bytes_per_launch": (2 * rd + wr) / nl,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB units); FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md HBM section (gfx950 reports half of a 16 B/lane streaming read)",
               "source": os.path.basename(src.rstrip("/"))}
        if tot("SQ_ACTIVE_INST_VALU") and tot("GRBM_GUI_ACTIVE"):
            w = tot("SQ_WAIT_ANY") + tot("SQ_WAIT_INST_ANY") + tot("SQ_ACTIVE_INST_ANY")
            out["valu"] = {"busy_frac": round(tot("SQ_ACTIVE_INST_VALU") * 4 / (1024 * tot("GRBM_GUI_ACTIVE") / 8), 4),
                           "lane_utilisation": round(tot("SQ_THREAD_CYCLES_VALU") / max(1.0, 64 * tot("SQ_ACTIVE_INST_VALU")), 4),
                           "valu_insts_per_launch": tot("SQ_INSTS_VALU") / max(1, sum(calls[k].get("SQ_INSTS_VALU", 0) for k in ks)),
                           "wave_cycle_shares": {"wait_any": round(tot("SQ_WAIT_ANY") / w, 3), "wait_inst_any": round(tot("SQ_WAIT_INST_ANY") / w, 3),
                                                 "active_inst_any": round(tot("SQ_ACTIVE_INST_ANY") / w, 3)} if w else None}
        json.dump(out, open(os.path.splitext(dst)[0] + ".json", "w"), indent=1)
        break
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
