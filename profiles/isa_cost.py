#!/usr/bin/env python3
"""The compute side of the roofline as a FRACTION: what a kernel's own VALU instruction mix costs at best.

Inputs: the ISA of every k_path / k_bounce instantiation (`make -C julia-spira_amd/csrc asm` -> spira_tu_*.s) and the per-opcode issue
costs measured on MI355X (profiles/microbench/valu_peak.hip -> profiles/r04_valu_peak.txt: SIMD cycles one wave64 instruction of an
independent stream holds the issue port, best of 1 / 4 / 8 waves per SIMD).  Output (profiles/isa_cost.json), per kernel: the static
VALU histogram by class and, per class, the mean issue cost of the opcodes the kernel actually contains.  profiles/summarize.py prices
the DYNAMIC class counts of the PMC passes with these (the classes the counters know: FMA / MUL / ADD / TRANS in F32 and F64, INT32,
INT64, CVT; the rest — moves, compares, selects, min / max, the division helpers, lane ops — is `other`, priced with the mean cost of
exactly the `other` opcodes in this kernel's ISA), which gives `min_cycles_per_inst`; valu.issue_util = that / the measured SIMD
cycles per VALU instruction.

usage: python profiles/isa_cost.py [valu_peak.txt] -> profiles/isa_cost.json (+ a table on stdout)
"""
import json
import os
import re
import subprocess
import sys
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "julia-spira_amd", "csrc")
CXXFILT = "c++filt"        # (binutils)


def measured_costs(path):
    """opcode -> lowest measured SIMD cycles per wave64 instruction: the best occupancy of its own (same-opcode) stream, or — lower for compares, selects,
    conversions, integer and logic operations — its MARGINAL cost in a stream that alternates it with v_fma_f32 (`pair:fma+<opcode>` rows: the pair minus half
    the fma+fma pair at the same occupancy): a same-opcode stream serialises on one SGPR pair / one sub-unit in ways real code does not."""
    best, pair = {}, {}
    for ln in open(path):
        m = re.match(r"^(\S+)(?: \(([^)]*)\))?\s+waves/SIMD=(\d+).*=\s*([0-9.]+) cycles/instr", ln)
        if not m:
            continue
        name, occ, cyc = m.group(1) + ("(%s)" % m.group(2) if m.group(2) else ""), int(m.group(3)), float(m.group(4))
        if name.startswith("pair:fma+"):
            pair.setdefault(name[len("pair:fma+"):], {})[occ] = cyc
        else:
            best[name] = min(best.get(name, 1e9), cyc)
    alias = {"v_cmp_lt_f32_sgpr": "v_cmp_lt_f32(sgpr)", "v_cmp_lt_f32_vcc": "v_cmp_lt_f32(vcc)", "v_cndmask_b32": "v_cndmask_b32(sgpr)"}
    fma_pair = pair.get("v_fma_f32", {})
    for op, by_occ in pair.items():
        if op == "v_fma_f32":
            continue
        marg = min((c - fma_pair[o] / 2.0) for o, c in by_occ.items() if o in fma_pair)
        key = alias.get(op, op)
        best[key] = min(best.get(key, 1e9), max(marg, 1.0))
    if fma_pair:
        best["v_fma_f32"] = min(best.get("v_fma_f32", 1e9), min(fma_pair.values()) / 2.0)
    return best


# opcode (regex) -> (PMC class, measured opcode its cost is taken from)
RULES = [
    (r"v_(fma|fmac|mad|fmaak|fmamk)_f32", "fma_f32", "v_fma_f32"), (r"v_pk_fma_f32", "fma_f32", "v_pk_fma_f32"),
    (r"v_mul_f32|v_mul_legacy_f32", "mul_f32", "v_mul_f32"), (r"v_pk_mul_f32", "mul_f32", "v_pk_fma_f32"),
    (r"v_(add|sub|subrev)_f32", "add_f32", "v_add_f32"), (r"v_pk_add_f32", "add_f32", "v_pk_fma_f32"),
    (r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32|v_rcp_iflag_f32", "trans_f32", "v_rcp_f32"),
    (r"v_(fma|fmac)_f64", "fma_f64", "v_fma_f64"), (r"v_mul_f64", "mul_f64", "v_mul_f64"), (r"v_add_f64", "add_f64", "v_add_f64"),
    (r"v_(rcp|rsq)_f64", "trans_f64", "v_rcp_f64"), (r"v_sqrt_f64", "trans_f64", "v_sqrt_f64"),
    (r"v_cvt_f32_ubyte\d", "cvt", "v_cvt_f32_ubyte1"), (r"v_cvt_f32_f64|v_cvt_f64_f32", "cvt", "v_cvt_f32_f64"),
    (r"v_cvt_f64_(u32|i32)|v_cvt_(u32|i32)_f64", "cvt", "v_cvt_f64_u32"), (r"v_cvt_", "cvt", "v_cvt_f32_u32"),
    (r"v_mul_lo_u32|v_mul_lo_i32", "int32", "v_mul_lo_u32"), (r"v_mul_hi_(u32|i32)", "int32", "v_mul_hi_u32"),
    (r"v_mul_(u32|i32)_(u24|i24)", "int32", "v_mul_u32_u24"), (r"v_mad_(u32|i32)_(u24|i24)", "int32", "v_mad_u32_u24"),
    (r"v_mad_u64_u32|v_mad_i64_i32", "int64", "v_mul_lo_u32"),
    (r"v_(and|or|xor|not|xnor)_b32", "int32", "v_xor_b32"), (r"v_(lshlrev|lshrrev|ashrrev)_b32", "int32", "v_lshrrev_b32"),
    (r"v_(lshlrev|lshrrev|ashrrev)_b64", "int64", "v_mov_b64"),
    (r"v_(lshl_or|and_or|or3|add3|lshl_add|xad|add_lshl|xor3)_(b32|u32)", "int32", "v_lshl_or_b32"),
    (r"v_bfe_(u32|i32)", "int32", "v_bfe_u32"), (r"v_bfi_b32", "int32", "v_bfi_b32"), (r"v_perm_b32|v_alignbit_b32|v_alignbyte_b32", "int32", "v_perm_b32"),
    (r"v_(add|sub|subrev)_(u32|i32|co_u32)|v_(addc|subb|subbrev)_co_u32", "int32", "v_add_u32"),
    (r"v_(min|max)_(u32|i32)", "int32", "v_add_u32"), (r"v_(min3|max3|med3)_(u32|i32)", "int32", "v_min3_u32"),
    (r"v_(ffbl|ffbh|bcnt|bfrev)_", "int32", "v_ffbl_b32"), (r"v_mbcnt_", "int32", "v_mbcnt_lo_u32_b32"),
    # ---- what no class counter covers
    (r"v_(min|max)_f32", "other", "v_max_f32"), (r"v_(min3|max3|med3)_f32", "other", "v_max3_f32"), (r"v_(min|max)_f64", "other", "v_max_f64"),
    (r"v_cmpx?_\w+_f64|v_cmp_class_f64", "other", "v_cmp_lt_f64"), (r"v_cmpx?_", "other", "v_cmp_lt_f32(sgpr)"),
    (r"v_cndmask_b32", "other", "v_cndmask_b32(sgpr)"),
    (r"v_mov_b64|v_pk_mov_b32", "other", "v_mov_b64"), (r"v_mov_b32|v_accvgpr_|v_swap_b32", "other", "v_mov_b32"),
    (r"v_div_scale_f32", "other", "v_div_scale_f32"), (r"v_div_fmas_f32", "other", "v_div_fmas_f32"), (r"v_div_fixup_f32", "other", "v_div_fixup_f32"),
    (r"v_div_scale_f64", "other", "v_div_scale_f64"), (r"v_div_fmas_f64", "other", "v_div_fmas_f64"), (r"v_div_fixup_f64", "other", "v_div_fixup_f64"),
    (r"v_ldexp_f64|v_frexp_\w+_f64|v_(trunc|floor|ceil|rndne|fract)_f64", "other", "v_ldexp_f64"),
    (r"v_ldexp_f32|v_frexp_\w+_f32|v_(trunc|floor|ceil|rndne|fract)_f32", "other", "v_mul_f32"),
    (r"v_(readlane|readfirstlane|writelane)_b32|v_permlane|v_mov_b32_dpp|v_\w+_dpp", "other", "v_mov_b32"),
]
ARITH = ("fma_f32", "mul_f32", "add_f32", "trans_f32", "fma_f64", "mul_f64", "add_f64", "trans_f64")


def classify(op, costs, unpriced):
    for pat, cls, ref in RULES:
        if re.fullmatch(pat + r"(_e32|_e64|_sdwa|_dpp)?", op) or re.match(pat, op):
            if ref not in costs:
                unpriced[op] += 1
                return cls, 4.0
            return cls, costs[ref]
    unpriced[op] += 1
    return "other", 4.0


def kernels_of(path):
    """mangled kernel name -> list of VALU mnemonics (static)."""
    out, cur = {}, None
    for ln in open(path, errors="replace"):
        m = re.match(r"^(_ZN5spira\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur and re.match(r"^\.Lfunc_end", ln):
            cur = None
            continue
        if cur:
            m = re.match(r"^\s+(v_\w+)", ln)
            if m and not m.group(1).startswith("v_mfma"):
                out[cur].append(m.group(1))
    return out


def short(name):
    n = name.replace("void ", "").replace("spira::", "")
    return n.split("(")[0][:60]


def main():
    peak = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_valu_peak.txt")
    costs = measured_costs(peak)
    result, unpriced = {}, Counter()
    for tu in ("spira_tu_main.s", "spira_tu_f32.s", "spira_tu_f64mesh.s"):
        f = os.path.join(CSRC, tu)
        if not os.path.exists(f):
            raise SystemExit("missing %s: run `make -C julia-spira_amd/csrc asm` first" % f)
        ks = kernels_of(f)
        names = subprocess.run([CXXFILT] + list(ks), capture_output=True, text=True, check=True).stdout.strip().splitlines()
        for mangled, dem in zip(ks, names):
            k = short(dem)
            if not (k.startswith("k_path") or k.startswith("k_bounce")) or not ks[mangled]:
                continue
            n_cls, c_cls = Counter(), defaultdict(float)
            for op in ks[mangled]:
                cls, cost = classify(op, costs, unpriced)
                n_cls[cls] += 1
                c_cls[cls] += cost
            n = sum(n_cls.values())
            result[k] = {"valu_static": n, "translation_unit": tu,
                         "class_share_static": {c: round(n_cls[c] / n, 4) for c in sorted(n_cls)},
                         "class_cost": {c: round(c_cls[c] / n_cls[c], 3) for c in sorted(n_cls)},
                         "mean_cost_static": round(sum(c_cls.values()) / n, 3),
                         "non_arithmetic_share_static": round(1.0 - sum(n_cls[c] for c in ARITH) / n, 4)}
    out = {"what": "static VALU histogram of every k_path / k_bounce instantiation by PMC instruction class, and per class the mean measured issue cost "
                   "(SIMD cycles per wave64 instruction, best occupancy) of the opcodes the kernel contains; `other` = no class counter covers it",
           "costs_source": os.path.relpath(peak, ROOT), "opcode_costs": {k: round(v, 3) for k, v in sorted(costs.items())},
           "unpriced_opcodes_charged_4_cycles": dict(unpriced), "kernels": result}
    json.dump(out, open(os.path.join(ROOT, "profiles", "isa_cost.json"), "w"), indent=1)
    print("%-58s %7s %6s %6s  class costs" % ("kernel", "VALU", "mean", "nonar"))
    for k, v in sorted(result.items()):
        print("%-58s %7d %6.2f %6.2f  %s" % (k, v["valu_static"], v["mean_cost_static"], v["non_arithmetic_share_static"], v["class_cost"]))
    if unpriced:
        print("unpriced (charged 4.0):", dict(unpriced))


if __name__ == "__main__":
    main()
