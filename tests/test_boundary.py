"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every
symbol include/c2d.h declares, its host-side helpers agree with the oracle, and
the random stream is rocRAND's Philox4x32-10.  No compute entry point is called
(no GPU here)."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
PKG_DIR = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "c2d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(c2d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"libc2d.so does not export {name}"
    from c2d_amd import binding

    assert sorted(binding.EXPORTED_SYMBOLS) == names, "python mirror and header disagree"


def test_struct_layouts_match_reference_sizes(pkg):
    # utils.cu:74-106: Position 8 B, PositionWithVarAndPoseIdx 16 B, Variance 20 B, Pose 12 B, PoseCPVarAndPoseIdx 20 B
    assert pkg.SCENE_DT.itemsize == 16 and pkg.STD_DT.itemsize == 20
    assert pkg.POSE_DT.itemsize == 12 and pkg.ROW_DT.itemsize == 20
    assert pkg.SCENE_DT.names == ("x", "y", "var_idx", "pose_idx")
    assert pkg.ROW_DT.names == ("x", "y", "cp", "var_idx", "pose_idx")
    assert pkg.STD_DT.names == ("x", "y", "theta", "width", "height")
    assert pkg.POSE_DT.names == ("width", "height", "theta")


def test_version_and_status_strings(pkg):
    lib = pkg.load_library()
    assert lib.c2d_version() == 6   # 0.6: round 6 added c2d_ctx_info_sized (c2d_ctx_info the symbol writes the 0.4 layout only); 0.5: round 5 added c2d_dist_rccl_version and c2d_device_info.pci_bus_id (0.4: c2d_mc_poly_*, c2d_sat_rect_pairs_*_host, c2d_uniform_table_minstd, c2d_sqrt_f32, c2d_dist_stream_synchronize)
    assert lib.c2d_status_string(0) == b"ok"
    for st in (-1, -2, -3, -4, -5, -6):
        assert lib.c2d_status_string(st) not in (b"ok", b"unknown status")
    assert lib.c2d_status_string(-99) == b"unknown status"


def test_null_arguments_are_rejected_not_crashing(pkg):
    lib = pkg.load_library()
    assert lib.c2d_ctx_create(0, None) == -1
    assert lib.c2d_device_count(None) == -1
    assert lib.c2d_ctx_destroy(None) == 0
    assert lib.c2d_malloc(None, None, 16) == -1
    assert lib.c2d_sat_rect_pairs_verts(None, None, 10, None, None, None) == -1
    assert lib.c2d_mc_scenes(None, None, None) == -1


def test_host_stat_helpers_match_oracle(pkg, oracle):
    lib = pkg.load_library()
    rng = np.random.default_rng(0)
    for n in (1000, 2000, 20000, 120000, 4020000):
        for k in [0, 1, n // 3, n - 1, n] + list(rng.integers(0, n, 20)):
            a = lib.c2d_calc_slack(int(n), int(k))
            b = oracle.calc_slack(n, int(k))
            assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)
    bins = np.array([0, .01, .1, 1], np.float32)
    for p in [0, .005, .01, .05, .1, .5, 1, 2] + list(rng.uniform(0, 1, 50)):
        assert lib.c2d_get_bin(np.float32(p), bins.ctypes.data_as(C.POINTER(C.c_float)), 4) == oracle.get_bin(np.float32(p), bins)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc + rocRAND headers")
def test_stream_is_rocrand_philox(tmp_path, oracle):
    """The c2d streams are rocrand_state_philox4x32_10 with rocrand_init(seed, subsequence=scene, offset): offset =
    8*item for the scene sampler's layout, 4*(8*group + block) for the Monte-Carlo draws — checked by running
    rocRAND's own __host__ __device__ engine on the host."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "rocrand_stream"
    subprocess.run([hipcc, "-O1", "-x", "hip", "--cuda-host-only", "-I/opt/rocm/include",
                    os.path.join(ROOT, "oracle", "tools", "rocrand_stream.cpp"), "-o", str(exe)], check=True)
    for seed, scene, sample in [(0x0123456789ABCDEF, 0xFEDCBA9876543210, (1 << 33) - 4), (1234, 0, 0), (7, 1 << 32, 123456789)]:
        out = subprocess.run([str(exe), str(seed), str(scene), str(sample), "8"], check=True, capture_output=True, text=True).stdout
        ref = np.array([[int(v) for v in line.split()] for line in out.strip().splitlines()], np.uint32)
        assert np.array_equal(oracle.raw8(seed, scene, sample, 8), ref)
        # the Monte-Carlo loop's draw layout (groups of four samples): the same engine at offset 4 * (8 * group + block)
        out = subprocess.run([str(exe), str(seed), str(scene), str(sample), "9", "draws"], check=True, capture_output=True, text=True).stdout
        ref = np.array([[int(v) for v in line.split()] for line in out.strip().splitlines()], np.uint32)
        assert np.array_equal(oracle.draw_words(seed, scene, sample, 9), ref)
    # "rocRAND replacing curand": the five normals of a sample are rocRAND's own Box-Muller (rocrand_normal2's) of the
    # same words up to the rounding of the math functions (c2d fixes bit-reproducible log / sqrt / sincos forms and centres
    # the uniform at (x + 1/2) / 2^32 where rocRAND uses (x + 1) / 2^32; glibc / OCML differ between host and device anyway)
    out = subprocess.run([str(exe), "1234", "77", "1000", "4000", "normals"], check=True, capture_output=True, text=True).stdout
    theirs = np.array([[float(v) for v in line.split()] for line in out.strip().splitlines()], np.float64)
    ours = oracle.normals5(1234, 77, 1000, 4000).astype(np.float64)
    assert theirs.shape == ours.shape == (4000, 5)
    assert np.abs(ours - theirs).max() < 2e-5 and np.abs(ours - theirs).mean() < 2e-7
    assert abs(theirs.std() - 1.0) < 0.02


def test_headers_are_plain_c(tmp_path):
    """include/c2d.h and include/utils.h are the FFI surface: they must compile as C11 and as C++17."""
    src = tmp_path / "t.c"
    src.write_text('#include "c2d.h"\nint main(void){ float r[8]; create_rect(r, 2.f, 1.f); '
                   'return sizeof(c2d_mc_scenes_args) > 0 && sizeof(PoseCPVarAndPoseIdxIdx) == 24 && sizeof(PoseCPVarAndPoseIdx) == 20 ? 0 : 1; }\n')
    inc = "-I" + os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", inc, "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-Werror", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "t2.o")], check=True)
    # the struct sizes of utils.cu:74-104 (the program only uses header-inline code, so it links without libc2d)
    subprocess.run(["gcc", "-std=c11", inc, str(src), "-o", str(tmp_path / "t")], check=True)
    assert subprocess.run([str(tmp_path / "t")]).returncode == 0


def test_no_kernel_selects_on_a_stale_scalar_condition():
    """hipcc 7.2 compiled poly_bin_move_kernel's `min(n - tile0, 8192)` (64-bit) into an s_cselect on an SCC that an unrelated
    s_add_i32 had set: the last tile's loads ran past the batch (profiles/notes_r05_move_kernel_overread.md).  Every shipped
    build is disassembled and searched for that shape (tests/tools/scc_scan.py); the scanner is checked on the miscompiled
    sequence itself."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("scc_scan", os.path.join(ROOT, "tests", "tools", "scc_scan.py"))
    scc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scc)
    bad = """_ZN3c2d20poly_bin_move_kernelIjEEvNS_11BinMoveArgsE:
	s_sub_u32 s0, s42, s44
	s_subb_u32 s1, s43, s45
	s_lshl_b32 s33, s34, 1
	v_cmp_lt_u64_e32 vcc, s[0:1], v[0:1]
	s_add_i32 s35, s33, -1
	s_mov_b64 s[2:3], -1
	s_cbranch_vccz .LBB5_289
; %bb.29:
	s_cselect_b32 s48, s0, 0x2000
"""
    good = bad.replace("v_cmp_lt_u64_e32 vcc, s[0:1], v[0:1]\n\ts_add_i32 s35, s33, -1", "s_add_i32 s35, s33, -1\n\ts_cmpk_lt_u32 s0, 0x2000")
    assert len(scc.scan_lines(bad.splitlines(), "bad")) == 1 and scc.scan_lines(good.splitlines(), "good") == []
    # the same with the reader at the branch's TARGET (the scanner follows the control-flow graph, not the text order)
    across = bad.replace("s_cbranch_vccz .LBB5_289\n; %bb.29:\n", "s_cbranch_vccnz .LBB5_30\n\ts_endpgm\n.LBB5_30:\n")
    assert len(scc.scan_lines(across.splitlines(), "across")) == 1
    libs = [os.path.join(PKG_DIR, "lib", n) for n in ("libc2d.so", "libc2d_fmad1.so", "libc2d_fmad2.so", "libc2d_nopretest.so", "libc2d_movecheck.so")]
    libs.append(os.path.join(PKG_DIR, "lib-rehearsal", "libc2d.so"))
    for lib in libs:
        assert scc.scan_library(lib) == [], lib


def test_shipped_libraries_load_beside_pytorchs_hip_runtime():
    """A PyTorch process has loaded PyTorch's own libamdhip64.so.7 (HIP 7.0 here) before libc2d.so, and the dynamic loader binds
    libc2d.so to THAT copy: a library that needs a symbol version newer than it defines does not load there at all ("version
    `hip_7.1' not found": round 6 met it with hipStreamGetId, which is now looked up at run time, csrc/c2d_internal.hpp).  Every
    shipped build may need only versioned HIP symbols that PyTorch's runtime defines."""
    import glob
    import importlib.util

    spec = importlib.util.find_spec("torch")
    torch_hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    assert os.path.exists(torch_hip)

    def versions(path, defined):
        out = subprocess.run(["objdump", "-T", path], capture_output=True, text=True, check=True).stdout
        found = set()
        for ln in out.splitlines():
            m = re.search(r"\((hip_[0-9.]+)\)|\s(hip_[0-9.]+)\s", ln)
            if m and (("*UND*" in ln) != defined):
                found.add(m.group(1) or m.group(2))
        return found

    provides = versions(torch_hip, True)
    assert "hip_4.2" in provides and len(provides) >= 4, provides
    libs = glob.glob(os.path.join(PKG_DIR, "lib", "libc2d*.so")) + [os.path.join(PKG_DIR, "lib-rehearsal", "libc2d.so")]
    assert len(libs) >= 6
    for lib in libs:
        needs = versions(lib, False)
        assert needs and needs <= provides, (lib, sorted(needs - provides))
