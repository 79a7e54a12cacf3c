#!/usr/bin/env python3
"""Developer tool: look for the miscompilation that made poly_bin_move_kernel over-read (profiles/notes_r05_move_kernel_overread.md)
in gfx950 assembly: an instruction that READS the scalar condition code (s_cselect, s_cmov, s_cbranch_scc0/1) with a reaching
SCC writer (over the function's control-flow graph) that is not a comparison, a mask operation or a carry (e.g. an s_add_i32 of unrelated values).
usage: scc_scan.py FILE...   FILE.s (hipcc -S --cuda-device-only) or a built library / object (its gfx950 code objects are taken out
of the offload bundles and disassembled with llvm-objdump); prints every suspect with its function and context; exit 1 if any.
tests/test_boundary.py runs scan_library over the shipped libc2d.so."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

READERS = ("s_cselect_b32", "s_cselect_b64", "s_cmov_b32", "s_cmov_b64", "s_cbranch_scc0", "s_cbranch_scc1")
CARRY_READERS = ("s_addc_u32", "s_subb_u32")            # ... and the carry of a wider addition: only another addition's may feed them
CARRY = re.compile(r"^(s_add_u32|s_sub_u32|s_addc_u32|s_subb_u32|s_cmp_\w+|s_cmpk_\w+)$")  # (a comparison: a boolean added as the carry)
# SALU instructions that leave SCC alone
KEEPS = re.compile(r"^(s_mov|s_movk|s_cselect|s_cmov|s_mul_|s_mulk|s_load|s_buffer_load|s_store|s_waitcnt|s_nop|s_barrier|s_bfm|s_sext|s_pack|s_getpc|s_setpc|s_swappc|"
                   r"s_ff1|s_ff0|s_flbit|s_brev|s_bitset|s_cbranch|s_branch|s_endpgm|s_sleep|s_setprio|s_setreg|s_getreg|s_dcache|s_icache|s_memtime|s_memrealtime|"
                   r"s_sendmsg|s_trap|s_code_end|s_inst_prefetch|s_clause|s_set_gpr|s_rfe|s_sethalt|s_ttrace|s_atc|s_scratch|s_atomic|s_call|s_version|s_quadmask_dummy)")
# SCC writers whose SCC is a condition somebody would branch or select on
CONDITION = re.compile(r"^(s_cmp_|s_cmpk_|s_bitcmp|s_and_|s_or_|s_xor_|s_andn2_|s_orn2_|s_nand_|s_nor_|s_xnor_|s_not_|s_wqm_|s_quadmask|s_bcnt|s_.*saveexec|s_andn1|s_andn2_wrexec|"
                       r"s_min_|s_max_|s_abs|s_bfe_|s_lshr_|s_lshl_|s_ashr_|s_addc_|s_subb_|s_add_u32|s_sub_u32)")  # (add/sub_u32: the carry, as 64-bit arithmetic and divisions use it)


def scan_lines(lines, path):
    """lines: compiler assembly (labels `.LBBn_m:`, functions `name:`) or llvm-objdump --symbolize-operands output (`<Ln>:`, `addr <name>:`).
    Per function a control-flow graph of basic blocks; for every reader the SCC writers that REACH it (backwards through blocks
    that leave SCC alone, over fall-through and branch edges) are classified."""
    suspects = []
    func, blocks, labels = "?", [], {}   # a block: {"ins": [(line, op, text)], "falls": bool, "target": label or None}

    def new_block():
        blocks.append({"ins": [], "falls": True, "target": None})

    def writes_scc(op):
        return op.startswith("s_") and not KEEPS.match(op)

    def finish():
        preds = {i: [] for i in range(len(blocks))}
        for i, b in enumerate(blocks):
            if b["falls"] and i + 1 < len(blocks):
                preds[i + 1].append(i)
            if b["target"] in labels:
                preds[labels[b["target"]]].append(i)
        for i, b in enumerate(blocks):
            for k, (n, op, t) in enumerate(b["ins"]):
                if op not in READERS and op not in CARRY_READERS:
                    continue
                reaching = next(([x] for x in reversed(b["ins"][:k]) if writes_scc(x[1])), None)
                if reaching is None:  # none in this block: the last writer of every block that reaches its top
                    reaching, seen, todo = [], {i}, list(preds[i])
                    while todo:
                        j = todo.pop()
                        if j in seen:
                            continue
                        seen.add(j)
                        w = next((x for x in reversed(blocks[j]["ins"]) if writes_scc(x[1])), None)
                        if w is not None:
                            reaching.append(w)
                        else:
                            todo += preds[j]
                for m, wop, wt in reaching:
                    if not (CONDITION if op in READERS else CARRY).match(wop):
                        suspects.append((path, n, func, t, m, wt))

    new_block()
    for n, line in enumerate(lines, 1):
        t = line.split("//")[0].split(";")[0].strip()
        if not t or t.startswith((";", ".")) and not t.startswith(".LBB"):
            continue
        lab = re.match(r"^(?:(\.LBB\w+)|[0-9a-f]+ <(L\d+)>):$", t)
        if lab:
            new_block()
            labels[lab.group(1) or lab.group(2)] = len(blocks) - 1
            continue
        f = re.match(r"^(?:[0-9a-f]+ <([^>]+)>|([A-Za-z_][\w$.]*)):$", t)
        if f:
            finish()
            func, blocks, labels = f.group(1) or f.group(2), [], {}
            new_block()
            continue
        op = t.split()[0]
        blocks[-1]["ins"].append((n, op, t))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            blocks[-1]["falls"] = op.startswith("s_cbranch") or op.startswith("s_swappc")
            if op.startswith(("s_cbranch", "s_branch")):
                blocks[-1]["target"] = t.split()[-1]
            new_block()
    finish()
    return suspects


def scan(path):
    if path.endswith(".s"):
        return scan_lines(open(path), path)
    return scan_library(path)


def code_objects(path):
    """the gfx950 code objects in the clang offload bundles of a host object or shared library"""
    blob, magic, at = open(path, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__", 0
    while (at := blob.find(magic, at)) >= 0:
        (count,) = struct.unpack_from("<Q", blob, at + 24)
        o = at + 32
        for _ in range(count):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tl].decode()
            o += 24 + tl
            if "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        at += len(magic)


def scan_library(path):
    found, n_objects = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for k, co in enumerate(code_objects(path)):
            n_objects += 1
            elf = os.path.join(tmp, f"co{k}.elf")
            open(elf, "wb").write(co)
            dis = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", elf], check=True, capture_output=True, text=True).stdout
            found += scan_lines(dis.splitlines(), f"{os.path.basename(path)}[code object {k}]")
    if not n_objects:
        raise RuntimeError(f"{path}: no gfx950 code object found")
    return found


def main():
    found = []
    for p in sys.argv[1:]:
        found += scan(p)
    for path, n, func, t, m, prev in found:
        print(f"{path}:{n}: {t}\n    SCC from line {m}: {prev}\n    in {func}")
    print(f"{len(found)} suspect reads of SCC in {len(sys.argv) - 1} files")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
