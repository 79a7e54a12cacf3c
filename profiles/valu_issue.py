#!/usr/bin/env python3
"""valu_issue.py — an issue-cost-weighted VALU roof for the Monte-Carlo kernels, from measured pieces only.

    python profiles/valu_issue.py mixes <tag>     # csrc/tools/instr_probe_mixes.inc: each leg's own instruction mix as a probe stream
    python profiles/valu_issue.py write <tag>     # profiles/<tag>_valu_issue.json from the tag's digests (+ a cross-compile for the static mix)
    python profiles/valu_issue.py verify <tag>    # recompute every number of that file from the pieces it records and the digests it cites
    python profiles/valu_issue.py show <tag>

Why.  bench.py prices VALU kernels at "2 ticks per wave instruction at 2.4 GHz" (MI355X_MICROARCH.md's fp32 peak).  The honest
question is: of the issue ticks the chip had during the kernel, how many did the kernel's own instruction mix NEED?

What csrc/tools/instr_probe measures (profiles/<tag>_instr_probe.txt; one asm block per kernel, four waves per SIMD, sixteen
independent accumulators, loops aligned to 256 bytes): a stream of ONE instruction type issues at 1.50 ticks per wave instruction
for fp32 add / mul, v_mov, v_add_u32, v_and / v_xor; 1.87 for fma / fmaak; 2.06 for min / max / min3 / max3 / med3, v_cvt_i32_f32,
v_rndne, the packed fp32 pair; 2.25 for v_mul_lo / hi_u32; 2.84 for a compare into vcc; 3.37 for shifts, v_bfi, v_alignbit, v_bitop3,
v_fmac, v_cvt_f32_u32, v_mul_u32_u24, v_mbcnt, v_bcnt; 3.5 for v_mad_u64_u32 and compares into an SGPR pair; 4.02 for the
transcendentals; 12.9 for v_cndmask_b32 on vcc.  (Alignment matters that much: with the loop wherever the assembler put it the same
streams read 1.75 / 3.63 / 6.49 — the instruction fetch, not the issue port.)  And those prices DO NOT ADD: mul alternating with
v_bitop3 runs at 1.50 per instruction (not 2.44), with cndmask at 1.87 (not 7.2), with max3 at 1.87 (more than 1.78), with
v_mad_u64_u32 at 2.67 (2.52).  A sum of count x price therefore is no roof — for mc_pair_kernel it EXCEEDS the kernel's own run
time, for the polygon kernels it is below what any stream of their mix reaches (recorded as `additive_ticks_per_wave_instr`, for
the record only).  The roof used instead is measured directly:

    the leg's own mix as a dependency-free stream — 512 instructions in the proportions below, in six orders (four shuffles, one
    that spreads every type evenly, one in runs of a type: the rate depends on which instructions are neighbours), run by the probe
    like any other stream (`mix <entry> order <k>` lines of the probe file); the FASTEST order counts -> ticks per wave instruction
    needed ticks per launch  =  wave instructions per launch (SQ_INSTS_VALU)  x  that
    frac_issue_weighted      =  needed ticks / (SIMDs x held clock x kernel time)                    (bench.py does this step live)

  proportions  DYNAMIC class totals per launch from the per-type PMC counters of gfx950 (SQ_INSTS_VALU_ADD_F32, _MUL_F32, _FMA_F32,
        _TRANS_F32, _INT32, _INT64, _CVT; OTHER = SQ_INSTS_VALU minus their sum: profiles/<tag>_pmc_valu_types.txt, a rocprofv3 --pmc
        pass over bench.py's Monte-Carlo legs), split INSIDE a class by how often the kernels' code uses each instruction in its
        loops (static counts of the gfx950 assembly hipcc emits, loop depth >= 1 by LLVM's block annotations).  Which class an
        instruction is counted in is not guessed: the probe binary ran under the same counters (profiles/<tag>_probe_types.txt).
  ticks are s_memtime ticks, the unit of the held clock the clock build records (tests/tools/mc_clock.py), so the two divide.

What is measured and what is modelled: class totals, prices and the mix stream's rate are measured; the split inside a class is
static; the ORDER of the stream is the best of six, not the kernel's own order (which has its dependencies, its scalar
code, its LDS traffic and its branches: that is what the fraction leaves room for).
"""
from __future__ import annotations

import collections
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import isa_digest  # noqa: E402

CLASSES = ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT")
SIMDS = 256 * 4

# kernel (digest name prefix) -> (source file, demangled-name prefix in the assembly, measured_counts entry)
KERNELS = {
    "c2d::mc_pair_kernel": ("c2d_mc.hip", "mc_pair_kernel", "mc_pair.config3"),
    "c2d::mc_scenes_advance_kernel<true>": ("c2d_mc.hip", "void mc_scenes_advance_kernel<true>", "mc_scenes.config4"),
    "c2d::mc_scenes_advance_kernel<false>": ("c2d_mc.hip", "void mc_scenes_advance_kernel<false>", "mc_scenes.config4"),
    "c2d::mc_poly_pair_kernel": ("c2d_mc_poly.hip", "mc_poly_pair_kernel", "mc_poly_pair.bench"),
    "c2d::mc_poly_scenes_advance_kernel<true>": ("c2d_mc_poly.hip", "void mc_poly_scenes_advance_kernel<true>", "mc_poly_scenes.bench"),
    "c2d::mc_poly_scenes_advance_kernel<false>": ("c2d_mc_poly.hip", "void mc_poly_scenes_advance_kernel<false>", "mc_poly_scenes.bench"),
}

# assembly mnemonic (suffixes stripped) -> the probed instruction that prices it.  Families share an execution path: the probe shows
# one price per family member it covers (v_min = v_max, v_min3 = v_max3 = v_med3, every compare, every shift).
PRICED_AS = [
    (r"v_(add|sub|subrev)_f32$", "v_add_f32"), (r"v_mul_f32$", "v_mul_f32"), (r"v_(fma|mad)_f32$", "v_fma_f32"),
    (r"v_(fmaak|fmamk|madak|madmk)_f32$", "v_fmaak_f32"), (r"v_(fmac|mac)_f32$", "v_fmac_f32"),
    (r"v_(max|min)_f32$", "v_max_f32"), (r"v_(max3|min3|med3)_f32$", "v_max3_f32"),
    (r"v_(rcp|rsq|rcp_iflag)_f32$", "v_rcp_f32"), (r"v_sqrt_f32$", "v_sqrt_f32"), (r"v_exp_f32$", "v_exp_f32"), (r"v_log_f32$", "v_log_f32"),
    (r"v_(sin|cos)_f32$", "v_sin_f32"),
    (r"v_cvt_(i32|u32)_f32$", "v_cvt_i32_f32"), (r"v_cvt_f32_(u32|i32|ubyte\d)$", "v_cvt_f32_u32"), (r"v_cvt_", "v_cvt_i32_f32"),
    (r"v_(rndne|floor|ceil|trunc|fract)_f32$", "v_rndne_f32"), (r"v_(ldexp|frexp_mant|frexp_exp_i32)_f32$", "v_rndne_f32"),
    (r"v_cmpx?_\w+_f32$", "v_cmp_lt_f32 (vcc)"), (r"v_cmpx?_class_f32$", "v_cmp_lt_f32 (vcc)"), (r"v_cmpx?_\w+_[iu](32|64|16)$", "v_cmp_ne_u32"),
    (r"v_cndmask_b32$", "v_cndmask_b32"),
    (r"v_mov_b(32|64)$", "v_mov_b32"), (r"v_(accvgpr_\w+|swap_b32)$", "v_mov_b32"),
    (r"v_(add|sub|subrev)_(u32|i32)$", "v_add_u32"), (r"v_(add|sub|subrev)_co_u32$", "v_add_u32"), (r"v_(addc|subb|subbrev)_co_u32$", "v_add_u32"),
    (r"v_(and|or|not)_b32$", "v_and_b32"), (r"v_(xor|xnor)_b32$", "v_xor_b32"),
    (r"v_(lshlrev|lshrrev|ashrrev)_[bi]32$", "v_lshlrev_b32"), (r"v_(lshlrev|lshrrev|ashrrev)_[bi]64$", "v_lshl_add_u64"),
    (r"v_(lshl_add|add_lshl|lshl_or|add3|and_or|or3|xad)_u32$", "v_lshl_add_u32"), (r"v_(and_or|or3)_b32$", "v_lshl_add_u32"),
    (r"v_(bfi|bfe)_[biu]32$", "v_bfi_b32"), (r"v_(alignbit|alignbyte|perm)_b32$", "v_alignbit_b32"),
    (r"v_mul_lo_u32$", "v_mul_lo_u32"), (r"v_mul_hi_[ui]32$", "v_mul_hi_u32"), (r"v_mul_[ui]32_[ui]24$", "v_mul_u32_u24"), (r"v_mad_[ui]32_[ui]24$", "v_mul_u32_u24"),
    (r"v_bitop3_b32$", "v_bitop3_b32"), (r"v_mbcnt_(lo|hi)_u32_b32$", "v_mbcnt_lo_u32_b32"), (r"v_bcnt_u32_b32$", "v_bcnt_u32_b32"),
    (r"v_mad_[ui]64_[ui]32$", "v_mad_u64_u32"), (r"v_lshl_add_u64$", "v_lshl_add_u64"),
    (r"v_pk_(mul|add)_f32$", "v_pk_mul_f32"), (r"v_pk_fma_f32$", "v_pk_fma_f32"),
    (r"v_div_(scale|fmas|fixup)_f32$", "v_fma_f32"),
]
# cross-lane moves into / out of scalar registers: issued by the VALU, counted in SQ_INSTS_VALU, not probed — priced as a compare
# (the other VALU instructions that talk to the scalar file)
UNPROBED_AS = [(r"v_(readlane|readfirstlane|writelane)_b32$", "v_cmp_lt_f32 (sgpr)")]


def strip(op: str) -> str:
    return re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)


def priced_as(op: str):
    o = strip(op)
    for pat, probe in PRICED_AS + UNPROBED_AS:
        if re.match(pat, o):
            return probe
    return None


def read_digest(path):
    """{kernel: {counter: (launches, mean)}} from a pmc_digest-style text file (any name width)"""
    out = collections.defaultdict(dict)
    for ln in open(path):
        m = re.match(r"(.+?)\s+(SQ_\w+|GRBM_\w+)\s+n=\s*(\d+) mean=\s*([\d.]+)", ln)
        if m:
            out[m.group(1).strip()][m.group(2)] = (int(m.group(3)), float(m.group(4)))
    return out


def read_probe(path):
    """{instruction: issue ticks per wave instruction}"""
    out = {}
    for ln in open(path):
        m = re.match(r"(v_\S+(?: \(\w+\))?)\s+ticks per own instr\s+[\d.]+ -> issue cost ([\d.]+) ticks", ln)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


def probe_classes(path):
    """{probed instruction: hardware class} from the probe binary's own run under the per-type counters"""
    names = {"k_fma": "v_fma_f32", "k_mul": "v_mul_f32", "k_add": "v_add_f32", "k_sub": "v_sub_f32", "k_max": "v_max_f32", "k_min": "v_min_f32",
             "k_min3": "v_min3_f32", "k_max3": "v_max3_f32", "k_med3": "v_med3_f32", "k_fmaak": "v_fmaak_f32", "k_fmac": "v_fmac_f32", "k_rcp": "v_rcp_f32",
             "k_sqrt": "v_sqrt_f32", "k_exp": "v_exp_f32", "k_log": "v_log_f32", "k_sin": "v_sin_f32", "k_cvt": "v_cvt_i32_f32", "k_cvtfu": "v_cvt_f32_u32",
             "k_rndne": "v_rndne_f32", "k_cmp32": "v_cmp_lt_f32 (vcc)", "k_cmp64": "v_cmp_lt_f32 (sgpr)", "k_cmpu": "v_cmp_ne_u32", "k_cndmask": "v_cndmask_b32",
             "k_mov": "v_mov_b32", "k_addu": "v_add_u32", "k_and": "v_and_b32", "k_xor": "v_xor_b32", "k_lshl": "v_lshlrev_b32", "k_lshladd": "v_lshl_add_u32",
             "k_bfi": "v_bfi_b32", "k_alignbit": "v_alignbit_b32", "k_mul_lo": "v_mul_lo_u32", "k_mul_hi": "v_mul_hi_u32", "k_mul_u24": "v_mul_u32_u24",
             "k_bitop3": "v_bitop3_b32", "k_mbcnt": "v_mbcnt_lo_u32_b32", "k_bcnt": "v_bcnt_u32_b32", "k_mad64": "v_mad_u64_u32", "k_lshladd64": "v_lshl_add_u64",
             "k_pkmul": "v_pk_mul_f32", "k_pkfma": "v_pk_fma_f32"}
    dig = read_digest(path)
    out = {}
    for k, c in dig.items():
        if k not in names or "SQ_INSTS_VALU" not in c:
            continue
        total = c["SQ_INSTS_VALU"][1]
        cls = "OTHER"
        for C in CLASSES:
            if c.get("SQ_INSTS_VALU_" + C, (0, 0.0))[1] > 0.9 * total:
                cls = C
        out[names[k]] = cls
    return out


def static_mix(src: str, name_prefix: str, min_depth: int = 1):
    """{mnemonic: count} of the VALU instructions of one kernel at loop depth >= min_depth"""
    text = isa_digest.assemble(os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "csrc", src), [])
    bodies = isa_digest.kernel_bodies(text)
    names = isa_digest.demangle(list(bodies))
    mix = collections.Counter()
    for k, body in bodies.items():
        short = re.sub(r"\(.*", "", names[k]).replace("c2d::", "")
        if short != name_prefix:
            continue
        depth = 0
        for ln in body:
            m = re.match(r"^\.LBB\d+_\d+:(.*)$", ln)
            if m:
                d = re.search(r"Depth=(\d+)", m.group(1))
                depth = int(d.group(1)) if d else 0
                continue
            m = re.match(r"\s+(v_[a-z_0-9]+)", ln)
            if m and depth >= min_depth:
                mix[strip(m.group(1))] += 1
    return dict(mix)


def class_prices(mix: dict, probe: dict, cls_of: dict):
    """per hardware class: (price = static-mix-weighted mean of the probed prices, what made it up); unmapped mnemonics are listed"""
    by_class = collections.defaultdict(lambda: collections.Counter())
    unmapped = collections.Counter()
    for op, n in mix.items():
        p = priced_as(op)
        if p is None or p not in probe or p not in cls_of:
            unmapped[op] += n
            continue
        cls = cls_of[p]
        if re.match(UNPROBED_AS[0][0], op):
            cls = "OTHER"
        by_class[cls][p] += n
    prices, detail = {}, {}
    for cls, members in by_class.items():
        tot = sum(members.values())
        prices[cls] = sum(n * probe[p] for p, n in members.items()) / tot
        detail[cls] = {p: n for p, n in members.most_common()}
    return prices, detail, dict(unmapped)


MIX_LEN = 512
MIX_ORDERS = 6   # orders per leg: four shuffles, one that spreads every type evenly, one in runs of 16 of a type; the roof is the
                 # FASTEST of them (the rate depends on the order: neighbours overlap or not)
TEMPLATES = {
    "v_fma_f32": "v_fma_f32 {R}, {R}, 1.0, v50", "v_mul_f32": "v_mul_f32 {R}, 1.0, {R}", "v_add_f32": "v_add_f32 {R}, 1.0, {R}",
    "v_sub_f32": "v_sub_f32 {R}, {R}, v50", "v_max_f32": "v_max_f32 {R}, {R}, v50", "v_min_f32": "v_min_f32 {R}, {R}, v50",
    "v_min3_f32": "v_min3_f32 {R}, {R}, 1.0, v50", "v_max3_f32": "v_max3_f32 {R}, {R}, 1.0, v50", "v_med3_f32": "v_med3_f32 {R}, {R}, 1.0, v50",
    "v_fmaak_f32": "v_fmaak_f32 {R}, {R}, v50, 0x3f8ccccd", "v_fmac_f32": "v_fmac_f32 {R}, v50, v50",
    "v_rcp_f32": "v_rcp_f32 {R}, {R}", "v_sqrt_f32": "v_sqrt_f32 {R}, {R}", "v_exp_f32": "v_exp_f32 {R}, {R}", "v_log_f32": "v_log_f32 {R}, {R}",
    "v_sin_f32": "v_sin_f32 {R}, {R}", "v_cvt_i32_f32": "v_cvt_i32_f32 {R}, {R}", "v_cvt_f32_u32": "v_cvt_f32_u32 {R}, {R}", "v_rndne_f32": "v_rndne_f32 {R}, {R}",
    "v_cmp_lt_f32 (vcc)": "v_cmp_lt_f32 vcc, {R}, v50", "v_cmp_lt_f32 (sgpr)": "v_cmp_lt_f32 s[20:21], {R}, v50", "v_cmp_ne_u32": "v_cmp_ne_u32 vcc, {R}, v50",
    "v_cndmask_b32": "v_cndmask_b32 {R}, {R}, v50, vcc", "v_mov_b32": "v_mov_b32 {R}, v50", "v_add_u32": "v_add_u32 {R}, {R}, v50",
    "v_and_b32": "v_and_b32 {R}, {R}, v50", "v_xor_b32": "v_xor_b32 {R}, {R}, v50", "v_lshlrev_b32": "v_lshlrev_b32 {R}, 1, {R}",
    "v_lshl_add_u32": "v_lshl_add_u32 {R}, {R}, 1, v50", "v_bfi_b32": "v_bfi_b32 {R}, v50, {R}, v50", "v_alignbit_b32": "v_alignbit_b32 {R}, {R}, v50, 7",
    "v_mul_lo_u32": "v_mul_lo_u32 {R}, {R}, v50", "v_mul_hi_u32": "v_mul_hi_u32 {R}, {R}, v50", "v_mul_u32_u24": "v_mul_u32_u24 {R}, {R}, v50",
    "v_bitop3_b32": "v_bitop3_b32 {R}, {R}, v50, v50 bitop3:0x96", "v_mbcnt_lo_u32_b32": "v_mbcnt_lo_u32_b32 {R}, v50, {R}", "v_bcnt_u32_b32": "v_bcnt_u32_b32 {R}, {R}, v50",
    "v_mad_u64_u32": "v_mad_u64_u32 {P}, vcc, v50, v50, 0", "v_lshl_add_u64": "v_lshl_add_u64 {P}, {P}, 1, {P}",
    "v_pk_mul_f32": "v_pk_mul_f32 {P}, {P}, {P}", "v_pk_fma_f32": "v_pk_fma_f32 {P}, {P}, v[50:51], v[50:51]",
}


def entry_mix(class_counts: dict, members: dict, length: int = MIX_LEN):
    """{probed instruction: instances in a stream of `length`}: class share (dynamic) x member share inside the class (static),
    rounded by largest remainders"""
    total = sum(class_counts.values())
    want = {}
    for C, n in class_counts.items():
        mem = members.get(C)
        if not mem or n <= 0:
            continue
        ms = sum(mem.values())
        for p_, k in mem.items():
            want[p_] = want.get(p_, 0.0) + n / total * k / ms * length
    scale = length / sum(want.values())
    want = {k: v * scale for k, v in want.items()}
    got = {k: int(v) for k, v in want.items()}
    for k in sorted(want, key=lambda k: (want[k] - got[k], k), reverse=True)[:length - sum(got.values())]:
        got[k] += 1
    return {k: v for k, v in sorted(got.items()) if v}


def mix_sequence(mix: dict, order: int = 0):
    import random

    seq = [p_ for p_, n in sorted(mix.items()) for _ in range(n)]
    if order < 4:    # a shuffle
        random.Random(20260105 + order).shuffle(seq)
        return seq
    if order == 4:   # every type spread evenly over the stream: instance i of a type with n instances sits at (i + 1/2) / n
        keyed = sorted(((i + 0.5) / n, p_) for p_, n in sorted(mix.items()) for i in range(n))
        return [p_ for _, p_ in keyed]
    runs = []        # order 5: runs of up to 16 of one type, the runs of the types taking turns
    left = dict(sorted(mix.items()))
    while left:
        for p_ in list(left):
            take = min(16, left[p_])
            runs += [p_] * take
            left[p_] -= take
            if not left[p_]:
                del left[p_]
    return runs


def mix_asm(seq):
    lines, j32, j64 = [], 0, 0
    for p_ in seq:
        t = TEMPLATES[p_]
        if "{P}" in t:
            r = 52 + 2 * (j64 % 8)
            lines.append(t.replace("{P}", f"v[{r}:{r + 1}]"))
            j64 += 1
        else:
            lines.append(t.replace("{R}", f"v{32 + j32 % 16}"))
            j32 += 1
    return lines


def entry_name(entry: str) -> str:
    return re.sub(r"\W", "_", entry)


def write_mixes(tag: str) -> str:
    res = build(tag, with_mix_rates=False)
    out = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "csrc", "tools", "instr_probe_mixes.inc")
    with open(out, "w") as f:
        f.write(f"// GENERATED by `python profiles/valu_issue.py mixes {tag}` (do not edit): the instruction mix of each Monte-Carlo leg of bench.py\n"
                "// as a dependency-free stream for csrc/tools/instr_probe.hip.  Proportions: per-type PMC counts of the leg's kernels x the static\n"
                "// mix of their loop code; order: a shuffle with a fixed seed.  profiles/valu_issue.py documents the method.\n")
        ks = []
        for entry, e in res["entries"].items():
            for order in range(MIX_ORDERS):
                name = "%s_o%d" % (entry_name(entry), order)
                f.write(f"#define MIX_BODY_{name} \\\n")
                f.write(" \\\n".join('    "%s\\n"' % ln for ln in mix_asm(mix_sequence(e["mix"], order))) + "\n")
                f.write(f"PROBE_BODY(k_mix_{name}, REP16(INIT_F), MIX_BODY_{name})\n")
                ks.append('{"mix %s order %d", k_mix_%s, %d}' % (entry, order, name, MIX_LEN))
        f.write("#define MIX_KERNELS " + ", ".join(ks) + "\n")
    return out


def read_mix_rates(path):
    """{entry: [ticks per wave instruction of each order]} from the `mix <entry> order <k>` lines of a probe file"""
    out = collections.defaultdict(list)
    for ln in open(path):
        m = re.match(r"mix (\S+) order (\d+)\s+ticks per own instr\s+[\d.]+ -> issue cost ([\d.]+) ticks", ln)
        if m:
            out[m.group(1)].append(float(m.group(3)))
    return dict(out)


def needed_ticks(counts: dict, prices: dict):
    """sum over classes of dynamic count x class price; a class the static mix never saw is priced at the mean of the others"""
    fallback = sum(prices.values()) / len(prices)
    return sum(n * prices.get(c, fallback) for c, n in counts.items())


def build(tag: str, with_mix_rates: bool = True) -> dict:
    probe_path = os.path.join(HERE, f"{tag}_instr_probe.txt")
    probe = read_probe(probe_path)
    cls_of = probe_classes(os.path.join(HERE, f"{tag}_probe_types.txt"))
    dyn = read_digest(os.path.join(HERE, f"{tag}_pmc_valu_types.txt"))
    res = {"_doc": "GENERATED by profiles/valu_issue.py (do not edit): issue ticks the VALU port needs for each Monte-Carlo leg's instruction mix. "
                   "dynamic class counts: per-type PMC counters; split inside a class: static mix of the kernels' loop code; rate: the mix as a "
                   "dependency-free stream in csrc/tools/instr_probe.",
           "_tag": tag, "_sources": [f"profiles/{tag}_instr_probe.txt", f"profiles/{tag}_probe_types.txt", f"profiles/{tag}_pmc_valu_types.txt"],
           "_probe_ticks": probe, "_probe_class": cls_of, "kernels": {}, "entries": {}}
    for kname, (src, asm_name, entry) in KERNELS.items():
        key = next((k for k in dyn if k.startswith(kname[:48])), None)
        if key is None:
            continue
        c = dyn[key]
        launches, total = c["SQ_INSTS_VALU"]
        counts = {C: c.get("SQ_INSTS_VALU_" + C, (0, 0.0))[1] for C in CLASSES}
        counts["OTHER"] = total - sum(counts.values())
        mix = static_mix(src, asm_name)
        prices, detail, unmapped = class_prices(mix, probe, cls_of)
        ticks = needed_ticks(counts, prices)
        res["kernels"][kname] = {"entry": entry, "launches_in_the_pass": launches, "wave_instr_per_launch": total, "class_wave_instr_per_launch": counts,
                                 "class_price_ticks": {k: round(v, 4) for k, v in prices.items()}, "class_members_static": detail,
                                 "static_valu_in_loops": sum(mix.values()), "static_unpriced": unmapped,
                                 "additive_ticks_per_wave_instr": round(ticks / total, 4)}
    # per measured_counts entry: the adaptive legs sum their two kernels over all launches of the pass
    rates = read_mix_rates(probe_path) if with_mix_rates else {}
    by_entry = collections.defaultdict(list)
    for kname, k in res["kernels"].items():
        by_entry[k["entry"]].append(k)
    for entry, ks in by_entry.items():
        w = [k["launches_in_the_pass"] if len(ks) > 1 else 1 for k in ks]
        counts = {C: sum(k["class_wave_instr_per_launch"][C] * wi for k, wi in zip(ks, w)) for C in CLASSES + ("OTHER",)}
        members = collections.defaultdict(collections.Counter)
        for k in ks:
            for C, mem in k["class_members_static"].items():
                members[C].update(mem)
        members = {C: dict(m) for C, m in members.items()}
        instr = sum(counts.values())
        e = {"class_share": {C: round(n / instr, 5) for C, n in counts.items()}, "mix": entry_mix(counts, members), "mix_length": MIX_LEN,
             "additive_ticks_per_wave_instr": round(sum(k["additive_ticks_per_wave_instr"] * k["wave_instr_per_launch"] * wi for k, wi in zip(ks, w)) / instr, 4)}
        if entry in rates:
            e["mix_ticks_per_wave_instr_by_order"] = rates[entry]
            e["mix_ticks_per_wave_instr"] = min(rates[entry])   # the fastest order: what the port can do for this mix
        res["entries"][entry] = e
    return res


def verify(tag: str) -> list[str]:
    """every number of <tag>_valu_issue.json again, from the digests it cites and the pieces it records (no compiler needed: the static
    mix is taken as recorded)"""
    path = os.path.join(HERE, f"{tag}_valu_issue.json")
    cur = json.load(open(path))
    probe_path = os.path.join(HERE, f"{tag}_instr_probe.txt")
    probe = read_probe(probe_path)
    cls_of = probe_classes(os.path.join(HERE, f"{tag}_probe_types.txt"))
    dyn = read_digest(os.path.join(HERE, f"{tag}_pmc_valu_types.txt"))
    rates = read_mix_rates(probe_path)
    bad = []
    if cur["_probe_ticks"] != probe:
        bad.append("probe ticks differ from the probe file")
    if cur["_probe_class"] != cls_of:
        bad.append("probe classes differ from the probe's counter digest")
    by_entry = collections.defaultdict(list)
    for kname, k in cur["kernels"].items():
        by_entry[k["entry"]].append(k)
        key = next((d for d in dyn if d.startswith(kname[:48])), None)
        if key is None:
            bad.append(f"{kname}: not in the counter digest")
            continue
        c = dyn[key]
        total = c["SQ_INSTS_VALU"][1]
        counts = {C: c.get("SQ_INSTS_VALU_" + C, (0, 0.0))[1] for C in CLASSES}
        counts["OTHER"] = total - sum(counts.values())
        if any(abs(counts[C] - k["class_wave_instr_per_launch"][C]) > 0.5 for C in counts) or c["SQ_INSTS_VALU"][0] != k["launches_in_the_pass"]:
            bad.append(f"{kname}: class counts differ from the counter digest")
        prices = {cls: sum(n * probe[p] for p, n in members.items()) / sum(members.values()) for cls, members in k["class_members_static"].items()}
        if any(abs(prices[c_] - k["class_price_ticks"][c_]) > 1e-3 for c_ in prices):
            bad.append(f"{kname}: class prices do not follow from the recorded members and the probe")
        if abs(needed_ticks(counts, prices) / total - k["additive_ticks_per_wave_instr"]) > 1e-3:
            bad.append(f"{kname}: additive ticks do not follow")
    for entry, e in cur["entries"].items():
        ks = by_entry[entry]
        w = [k["launches_in_the_pass"] if len(ks) > 1 else 1 for k in ks]
        counts = {C: sum(k["class_wave_instr_per_launch"][C] * wi for k, wi in zip(ks, w)) for C in CLASSES + ("OTHER",)}
        members = collections.defaultdict(collections.Counter)
        for k in ks:
            for C, mem in k["class_members_static"].items():
                members[C].update(mem)
        if entry_mix(counts, {C: dict(m) for C, m in members.items()}) != e["mix"]:
            bad.append(f"{entry}: the mix does not follow from the class counts and the static members")
        if e.get("mix_ticks_per_wave_instr_by_order") != rates.get(entry) or e.get("mix_ticks_per_wave_instr") != min(rates.get(entry) or [None]):
            bad.append(f"{entry}: mix rates {e.get('mix_ticks_per_wave_instr_by_order')!r} are not the probe file's {rates.get(entry)!r}")
    # the generated probe source holds exactly these streams
    inc = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "csrc", "tools", "instr_probe_mixes.inc")
    if os.path.exists(inc) and f"mixes {tag}`" in open(inc).read():
        text = open(inc).read()
        for entry, e in cur["entries"].items():
            for order in range(MIX_ORDERS):
                body = text.split(f"#define MIX_BODY_{entry_name(entry)}_o{order} ")[1].split("PROBE_BODY")[0]
                have = re.findall(r'"(.+?)\\n"', body)
                if have != mix_asm(mix_sequence(e["mix"], order)):
                    bad.append(f"{entry}: instr_probe_mixes.inc does not hold order {order} of this mix")
    return bad


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "mixes":
        print("wrote", write_mixes(sys.argv[2]))
    elif len(sys.argv) == 3 and sys.argv[1] == "write":
        out = os.path.join(HERE, f"{sys.argv[2]}_valu_issue.json")
        json.dump(build(sys.argv[2]), open(out, "w"), indent=1)
        print("wrote", out)
    elif len(sys.argv) == 3 and sys.argv[1] == "verify":
        problems = verify(sys.argv[2])
        for p in problems:
            print("MISMATCH", p)
        sys.exit(1 if problems else 0)
    elif len(sys.argv) == 3 and sys.argv[1] == "show":
        r = json.load(open(os.path.join(HERE, f"{sys.argv[2]}_valu_issue.json")))
        for entry, e in r["entries"].items():
            print(f"{entry}: mix stream {e.get('mix_ticks_per_wave_instr')} ticks per wave instruction (sum of single-type prices: {e['additive_ticks_per_wave_instr']})")
            print("    classes:", e["class_share"])
            print("    stream :", e["mix"])
        for kname, k in r["kernels"].items():
            if k["static_unpriced"]:
                print(f"{kname}: unpriced (static): {k['static_unpriced']}")
    else:
        raise SystemExit(__doc__)
