#!/usr/bin/env python3
"""isa_digest.py — per-kernel digest of the gfx950 code hipcc generates for a c2d source file.

    python profiles/isa_digest.py convex-2d-gpu-collision-detection_amd/csrc/c2d_mc.hip [-D...] [--scratch] [--md]

Compiles the file device-only to assembly with the flags of the top-level Makefile (plus any -D given), then prints,
per kernel: VGPRs, SGPRs, spill counts, scratch bytes, LDS bytes, occupancy, code size, and static instruction counts
by class.  With --scratch every scratch_load / scratch_store is listed with the loop depth it executes at, taken from
LLVM's own block annotations in the assembly ("in Loop: Header=BBx_y Depth=N"): depth 0 = straight-line code outside
every loop, depth 1 = the kernel's outermost loop (for the adaptive Monte-Carlo kernels: the loop over work items).
Static counts are not dynamic counts: they say what exists in the code, the PMC passes say what runs.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include")]


def assemble(src: str, defines: list[str]) -> str:
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC] + FLAGS + defines + ["--cuda-device-only", "-S", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def demangle(names: list[str]) -> dict[str, str]:
    try:
        r = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def classify(op: str) -> str:
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_bodies(text: str) -> dict[str, list[str]]:
    """kernel symbol -> its lines (between `sym:` and the matching .Lfunc_end)."""
    lines = text.splitlines()
    kernels = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, flags=re.M)
    bodies = {}
    for k in kernels:
        start = next((i for i, ln in enumerate(lines) if ln.startswith(k + ":")), None)
        if start is None:
            continue
        end = start
        while end < len(lines) and not lines[end].startswith(".Lfunc_end") and ".amdhsa_kernel" not in lines[end]:
            end += 1
        bodies[k] = lines[start + 1:end]
    return bodies


def kernel_metadata(text: str) -> dict[str, dict[str, int]]:
    """Parse the amdhsa.kernels YAML block by hand: one dict per kernel keyed by .symbol's base name."""
    out = {}
    blocks = re.split(r"\n  - \.agpr_count:", text)
    for b in blocks[1:]:
        b = ".agpr_count:" + b
        name = re.search(r"\.name:\s+(\S+)", b)
        if not name:
            continue
        d = {}
        for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
                    "agpr_count", "max_flat_workgroup_size"):
            m = re.search(r"\.%s:\s+(\d+)" % key, b)
            if m:
                d[key] = int(m.group(1))
        out[name.group(1)] = d
    return out


def loops_of(body: list[str]):
    """[(start_line, end_line, label)] for every backward branch."""
    label_at = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = i
    loops = []
    for j, ln in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= j:
            loops.append((label_at[m.group(1)], j, m.group(1)))
    return loops


def digest(src: str, defines: list[str], show_scratch: bool, md_out: bool) -> None:
    text = assemble(src, defines)
    bodies = kernel_bodies(text)
    meta = kernel_metadata(text)
    names = demangle(list(bodies))
    for k, body in bodies.items():
        counts: dict[str, int] = {}
        n_instr = 0
        for ln in body:
            m = re.match(r"\s+([a-z_0-9]+)", ln)
            if not m or ln.lstrip().startswith((".", ";")):
                continue
            op = m.group(1)
            if not re.match(r"(v_|s_|ds_|global_|flat_|buffer_|scratch_)", op):
                continue
            counts[classify(op)] = counts.get(classify(op), 0) + 1
            n_instr += 1
        md = meta.get(k, {})
        vg = md.get("vgpr_count", 0)
        waves = 8 if vg <= 64 else (512 // (((vg + 7) // 8) * 8))
        short = re.sub(r"\(.*", "", names[k]).replace("c2d::", "")
        head = (f"{short}: vgpr {vg} (<= {min(waves, 8)} waves/SIMD by registers), sgpr {md.get('sgpr_count', '?')}, "
                f"vgpr spills {md.get('vgpr_spill_count', 0)}, sgpr spills {md.get('sgpr_spill_count', 0)}, "
                f"scratch {md.get('private_segment_fixed_size', 0)} B, LDS {md.get('group_segment_fixed_size', 0)} B, {n_instr} instructions")
        print(("### " if md_out else "") + head)
        print("    " + ", ".join(f"{c} {counts[c]}" for c in sorted(counts)))
        if show_scratch and counts.get("scratch"):
            # LLVM annotates every basic block that sits in a loop ("in Loop: Header=BBx_y Depth=N" / "Loop Header: Depth=N");
            # the depth of the block a scratch instruction belongs to is the loop depth it executes at
            depth, header = 0, ""
            for i, ln in enumerate(body):
                m = re.match(r"^\.LBB\d+_\d+:(.*)$", ln)
                if m:
                    d = re.search(r"Depth=(\d+)", m.group(1))
                    h = re.search(r"Header=(BB\d+_\d+)", m.group(1))
                    depth = int(d.group(1)) if d else 0
                    header = h.group(1) if h else (ln.split(":")[0].lstrip(".L") if d else "")
                if re.match(r"\s+scratch_", ln):
                    where = "outside every loop" if depth == 0 else f"loop depth {depth} (header {header})"
                    print(f"      line {i:5d}: {ln.strip():60s} {where}")
        print()


if __name__ == "__main__":
    args = sys.argv[1:]
    defs = [a for a in args if a.startswith("-D")]
    files = [a for a in args if not a.startswith("-")]
    if not files:
        raise SystemExit(__doc__)
    for f in files:
        digest(f, defs, "--scratch" in args, "--md" in args)
