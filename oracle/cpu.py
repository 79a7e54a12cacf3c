"""ctypes binding of oracle/libc2d_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# "" = the canonical arithmetic; "fmad1" / "fmad2" = the contraction-study builds (load_variant below)
_VARIANT = globals().get("_VARIANT", "")
_LIB_NAME = "libc2d_oracle%s.so" % ("_" + _VARIANT if _VARIANT else "")
_LIB_PATH = os.path.join(_HERE, _LIB_NAME)
KMAX = 16


class Position(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Pose(C.Structure):
    _fields_ = [("width", C.c_float), ("height", C.c_float), ("theta", C.c_float)]


class StdDev(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("theta", C.c_float),
                ("width", C.c_float), ("height", C.c_float)]


class Polygon(C.Structure):
    _fields_ = [("k", C.c_uint32), ("x", C.c_float * KMAX), ("y", C.c_float * KMAX)]


POLY_DT = np.dtype([("k", "<u4"), ("x", "<f4", (KMAX,)), ("y", "<f4", (KMAX,))])
POLY_POSE_DT = np.dtype([("theta", "<f4"), ("obstacle", POLY_DT)])
POSE_DT = np.dtype([("width", "<f4"), ("height", "<f4"), ("theta", "<f4")])
STD_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("theta", "<f4"), ("width", "<f4"), ("height", "<f4")])
SCENE_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("var_idx", "<f4"), ("pose_idx", "<f4")])
ROW_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("cp", "<f4"), ("var_idx", "<f4"), ("pose_idx", "<f4")])


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile if the .so is missing or stale."""
    src = os.path.join(_HERE, "c2d_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", _LIB_NAME], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def load_variant(name: str):
    """A second copy of this module bound to libc2d_oracle_<name>.so (name: "fmad1" or "fmad2"): the same
    restatement with the reference's dot products contracted the way nvcc -fmad=true might (C2D_ORACLE_FMAD)."""
    import importlib.util
    import sys

    modname = __name__ + "_" + name
    if modname in sys.modules:
        return sys.modules[modname]
    spec = importlib.util.spec_from_file_location(modname, os.path.abspath(__file__))
    m = importlib.util.module_from_spec(spec)
    m._VARIANT = name
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.c2d_oracle_num_threads.restype = C.c_int
        L.c2d_oracle_build_info.restype = C.c_char_p
        L.c2d_oracle_logf.restype = C.c_float
        L.c2d_oracle_logf.argtypes = [C.c_float]
        L.c2d_oracle_sincosf.argtypes = [C.c_float, fp, fp]
        L.c2d_oracle_sincos_u32.argtypes = [C.c_uint32, fp, fp]
        L.c2d_oracle_convex_collide.restype = C.c_int
        L.c2d_oracle_convex_collide.argtypes = [fp, fp]
        L.c2d_oracle_calc_slack.restype = C.c_float
        L.c2d_oracle_calc_slack.argtypes = [C.c_uint32, C.c_uint32]
        L.c2d_oracle_get_bin.restype = C.c_int
        L.c2d_oracle_get_bin.argtypes = [C.c_float, fp, C.c_uint32]
        L.c2d_oracle_sat_rect_pairs_verts.restype = C.c_ulonglong
        L.c2d_oracle_sat_rect_pairs_pose.restype = C.c_ulonglong
        L.c2d_oracle_sat_poly_pairs.restype = C.c_ulonglong
        L.c2d_oracle_mc_pair.restype = C.c_ulonglong
        L.c2d_oracle_mc_pair.argtypes = [C.c_float, C.c_float, C.POINTER(Position), C.POINTER(Pose),
                                         C.POINTER(StdDev), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.c2d_oracle_mc_scenes.restype = C.c_ulonglong
        L.c2d_oracle_mc_poly_pair.restype = C.c_ulonglong
        L.c2d_oracle_mc_poly_pair.argtypes = [C.POINTER(Polygon), C.POINTER(Position), C.c_float, C.POINTER(Polygon),
                                              C.POINTER(StdDev), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.c2d_oracle_mc_poly_scenes.restype = C.c_ulonglong
        L.c2d_oracle_place_polygon.argtypes = [C.POINTER(Polygon), C.c_float, C.c_float, C.c_float, fp, fp]
        L.c2d_oracle_sample_polygon.argtypes = [C.POINTER(Polygon), C.POINTER(StdDev), fp, fp, fp]
        L.c2d_oracle_mc_poly_sampled.argtypes = [C.POINTER(Polygon), C.POINTER(StdDev), C.c_uint64, C.c_uint64, C.c_uint64, fp, fp]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ct=C.c_float):
    return a.ctypes.data_as(C.POINTER(ct))


def build_info() -> str:
    """Compiler and flags of the loaded library (oracle/Makefile), for bench.py's cpu_baseline line."""
    return lib().c2d_oracle_build_info().decode()


def num_threads() -> int:
    return lib().c2d_oracle_num_threads()


def usable_cores() -> int:
    """CPU cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (a GPU box hands a 1-GPU job a 16-core share of a 128-thread host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def set_num_threads(n: int) -> int:
    lib().c2d_oracle_set_num_threads(int(n))
    return num_threads()


def logf(u):
    u = _f32(np.atleast_1d(u))
    return np.array([lib().c2d_oracle_logf(float(v)) for v in u], dtype=np.float32)


def sincosf(x):
    x = _f32(np.atleast_1d(x))
    s = np.empty_like(x)
    c = np.empty_like(x)
    L = lib()
    sv, cv = C.c_float(), C.c_float()
    for i, v in enumerate(x):
        L.c2d_oracle_sincosf(float(v), C.byref(sv), C.byref(cv))
        s[i], c[i] = sv.value, cv.value
    return s, c


def sincos_u32(y):
    y = np.atleast_1d(np.asarray(y, dtype=np.uint32))
    s = np.empty(y.shape, np.float32)
    c = np.empty(y.shape, np.float32)
    L = lib()
    sv, cv = C.c_float(), C.c_float()
    for i, v in enumerate(y):
        L.c2d_oracle_sincos_u32(int(v), C.byref(sv), C.byref(cv))
        s[i], c[i] = sv.value, cv.value
    return s, c


def philox(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, np.uint32)
    lib().c2d_oracle_philox4x32_10(_ptr(ctr, C.c_uint32), _ptr(key, C.c_uint32), _ptr(out, C.c_uint32))
    return out


def raw8(seed, scene_id, sample_begin, n):
    out = np.empty((n, 8), np.uint32)
    L = lib()
    for i in range(n):
        L.c2d_oracle_raw8(C.c_uint64(seed), C.c_uint64(scene_id), C.c_uint64(sample_begin + i),
                          out[i].ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def draw_words(seed, scene_id, sample_begin, n):
    """The six raw words of each sample in draw order (radius1, angle1, radius2, angle2, radius3, angle3):
    the group-of-four draw layout of the Monte-Carlo loop (c2d_oracle.c, c2d_oracle_draw_words)."""
    out = np.empty((n, 6), np.uint32)
    L = lib()
    for i in range(n):
        L.c2d_oracle_draw_words(C.c_uint64(seed), C.c_uint64(scene_id), C.c_uint64(sample_begin + i),
                                out[i].ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def normals5(seed, scene_id, sample_begin, n):
    out = np.empty((n, 5), np.float32)
    L = lib()
    for i in range(n):
        L.c2d_oracle_normals5(C.c_uint64(seed), C.c_uint64(scene_id), C.c_uint64(sample_begin + i),
                              out[i].ctypes.data_as(C.POINTER(C.c_float)))
    return out


def create_rect(w, h):
    r = np.empty(8, np.float32)
    lib().c2d_oracle_create_rect(_ptr(r), C.c_float(w), C.c_float(h))
    return r


def rot_trans_rectangle(r, dx, dy, dt):
    r = _f32(r).copy()
    lib().c2d_oracle_rot_trans_rectangle(_ptr(r), C.c_float(dx), C.c_float(dy), C.c_float(dt))
    return r


def convex_collide(r1, r2) -> int:
    r1, r2 = _f32(r1), _f32(r2)
    return lib().c2d_oracle_convex_collide(_ptr(r1), _ptr(r2))


def rects_from_poses(cx, cy, w, h, theta):
    """-> float32 [8][n] planes"""
    cx, cy, w, h, theta = map(_f32, (cx, cy, w, h, theta))
    n = cx.shape[0]
    out = np.empty((8, n), np.float32)
    arr = (C.POINTER(C.c_float) * 8)(*[_ptr(out[k]) for k in range(8)])
    lib().c2d_oracle_rects_from_poses(_ptr(cx), _ptr(cy), _ptr(w), _ptr(h), _ptr(theta), C.c_size_t(n), arr)
    return out


def sat_rect_pairs_verts(planes):
    """planes: float32 [16][n] -> (uint8 [n], count)"""
    planes = _f32(planes)
    assert planes.shape[0] == 16
    n = planes.shape[1]
    out = np.empty(n, np.uint8)
    arr = (C.POINTER(C.c_float) * 16)(*[_ptr(planes[k]) for k in range(16)])
    cnt = lib().c2d_oracle_sat_rect_pairs_verts(arr, C.c_size_t(n), _ptr(out, C.c_uint8))
    return out, int(cnt)


def sat_rect_pairs_pose(pp):
    """pp: float32 [10][n] (cx,cy,w,h,theta for rect 1 then rect 2)"""
    pp = _f32(pp)
    assert pp.shape[0] == 10
    n = pp.shape[1]
    out = np.empty(n, np.uint8)
    arr = (C.POINTER(C.c_float) * 10)(*[_ptr(pp[k]) for k in range(10)])
    cnt = lib().c2d_oracle_sat_rect_pairs_pose(arr, C.c_size_t(n), _ptr(out, C.c_uint8))
    return out, int(cnt)


def sat_poly_pairs(vx, vy, k):
    """vx, vy: float32 [2][rows][n] (rows = vertex rows per polygon, 1..KMAX); k: uint8 [2][n]"""
    vx, vy = _f32(vx), _f32(vy)
    k = np.ascontiguousarray(k, dtype=np.uint8)
    n = vx.shape[-1]
    rows = vx.shape[1]
    assert vx.shape == vy.shape == (2, rows, n) and 1 <= rows <= KMAX and k.shape == (2, n)
    out = np.empty(n, np.uint8)
    L = lib()
    L.c2d_oracle_sat_poly_pairs_rows.restype = C.c_ulonglong
    cnt = L.c2d_oracle_sat_poly_pairs_rows(_ptr(vx), _ptr(vy), _ptr(k, C.c_uint8), C.c_size_t(n), C.c_int(rows),
                                           _ptr(out, C.c_uint8))
    if cnt == 2**64 - 1:
        raise ValueError("vertex count outside 1..rows")
    return out, int(cnt)


def calc_slack(n, k) -> float:
    return lib().c2d_oracle_calc_slack(int(n), int(k))


def get_bin(p, bins) -> int:
    bins = _f32(bins)
    return lib().c2d_oracle_get_bin(C.c_float(p), _ptr(bins), len(bins))


def mc_pair(robot_w, robot_h, pos, pose, sd, seed, scene_id, sample_begin, n_samples) -> int:
    return int(lib().c2d_oracle_mc_pair(robot_w, robot_h, C.byref(Position(*pos)), C.byref(Pose(*pose)),
                                        C.byref(StdDev(*sd)), seed, scene_id, sample_begin, n_samples))


def polygon(xs, ys=None) -> Polygon:
    """A Polygon from vertex coordinates (xs, ys) or from one POLY_DT record."""
    if ys is None:
        rec = xs
        xs, ys = rec["x"][:int(rec["k"])], rec["y"][:int(rec["k"])]
    xs, ys = _f32(xs), _f32(ys)
    assert 1 <= len(xs) == len(ys) <= KMAX
    p = Polygon()
    p.k = len(xs)
    for i in range(len(xs)):
        p.x[i], p.y[i] = float(xs[i]), float(ys[i])
    return p


def _as_polygon(p) -> Polygon:
    return p if isinstance(p, Polygon) else polygon(*p) if isinstance(p, tuple) else polygon(p)


def place_polygon(poly, dx, dy, dt):
    """robot placement: rot_trans_rectangle per vertex -> (x[k], y[k])"""
    poly = _as_polygon(poly)
    ox, oy = np.empty(poly.k, np.float32), np.empty(poly.k, np.float32)
    lib().c2d_oracle_place_polygon(C.byref(poly), dx, dy, dt, _ptr(ox), _ptr(oy))
    return ox, oy


def sample_polygon(poly, sd, normals5):
    poly = _as_polygon(poly)
    n5 = _f32(normals5)
    ox, oy = np.empty(poly.k, np.float32), np.empty(poly.k, np.float32)
    lib().c2d_oracle_sample_polygon(C.byref(poly), C.byref(StdDev(*sd)), _ptr(n5), _ptr(ox), _ptr(oy))
    return ox, oy


def mc_poly_sampled(obstacle, sd, seed, scene_id, sample):
    obstacle = _as_polygon(obstacle)
    ox, oy = np.empty(obstacle.k, np.float32), np.empty(obstacle.k, np.float32)
    lib().c2d_oracle_mc_poly_sampled(C.byref(obstacle), C.byref(StdDev(*sd)), seed, scene_id, sample, _ptr(ox), _ptr(oy))
    return ox, oy


def mc_poly_pair(robot, pos, robot_theta, obstacle, sd, seed, scene_id, sample_begin, n_samples) -> int:
    r = int(lib().c2d_oracle_mc_poly_pair(C.byref(_as_polygon(robot)), C.byref(Position(*pos)), robot_theta, C.byref(_as_polygon(obstacle)),
                                          C.byref(StdDev(*sd)), seed, scene_id, sample_begin, n_samples))
    if r == 2 ** 64 - 1:
        raise ValueError("polygon vertex count outside 1..KMAX")
    return r


def mc_poly_scenes(robot, poly_poses, std_devs, scenes, bins, acc, max_samples, seed, scene_id_base=0, schedule=(0, 0, 0)):
    """robot: Polygon; poly_poses: POLY_POSE_DT[np]; std_devs: STD_DT[nv]; scenes: SCENE_DT[n].
    -> hits u32[n], n_used u32[n], rows ROW_DT[n], total samples"""
    poly_poses = np.ascontiguousarray(poly_poses, dtype=POLY_POSE_DT)
    std_devs = np.ascontiguousarray(std_devs, dtype=STD_DT)
    scenes = np.ascontiguousarray(scenes, dtype=SCENE_DT)
    bins, acc = _f32(bins), _f32(acc)
    assert len(acc) == len(bins) - 1
    n = scenes.shape[0]
    hits, used, rows = np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, ROW_DT)
    robot = _as_polygon(robot)
    total = lib().c2d_oracle_mc_poly_scenes(
        C.byref(robot), C.c_void_p(poly_poses.ctypes.data), C.c_uint32(len(poly_poses)), C.c_void_p(std_devs.ctypes.data),
        C.c_uint32(len(std_devs)), C.c_void_p(scenes.ctypes.data), C.c_size_t(n), _ptr(bins), _ptr(acc), C.c_uint32(len(bins)),
        C.c_uint32(max_samples), C.c_uint64(seed), C.c_uint64(scene_id_base), C.c_uint32(schedule[0]), C.c_uint32(schedule[1]),
        C.c_uint32(schedule[2]), C.c_void_p(hits.ctypes.data), C.c_void_p(used.ctypes.data), C.c_void_p(rows.ctypes.data))
    return hits, used, rows, int(total)


def mc_sampled_rect(pose, sd, seed, scene_id, sample):
    out = np.empty(8, np.float32)
    lib().c2d_oracle_mc_sampled_rect(C.byref(Pose(*pose)), C.byref(StdDev(*sd)), C.c_uint64(seed),
                                     C.c_uint64(scene_id), C.c_uint64(sample), _ptr(out))
    return out


def mc_scenes(poses, std_devs, scenes, robot_w, robot_h, bins, acc, max_samples, seed, scene_id_base=0, schedule=(0, 0, 0)):
    """poses: POSE_DT[np]; std_devs: STD_DT[nv]; scenes: SCENE_DT[n].
    -> hits u32[n], n_used u32[n], rows ROW_DT[n], total samples"""
    poses = np.ascontiguousarray(poses, dtype=POSE_DT)
    std_devs = np.ascontiguousarray(std_devs, dtype=STD_DT)
    scenes = np.ascontiguousarray(scenes, dtype=SCENE_DT)
    bins, acc = _f32(bins), _f32(acc)
    assert len(acc) == len(bins) - 1
    n = scenes.shape[0]
    hits = np.empty(n, np.uint32)
    used = np.empty(n, np.uint32)
    rows = np.empty(n, ROW_DT)
    total = lib().c2d_oracle_mc_scenes(
        C.c_void_p(poses.ctypes.data), C.c_uint32(len(poses)), C.c_void_p(std_devs.ctypes.data),
        C.c_uint32(len(std_devs)), C.c_void_p(scenes.ctypes.data), C.c_size_t(n), C.c_float(robot_w),
        C.c_float(robot_h), _ptr(bins), _ptr(acc), C.c_uint32(len(bins)), C.c_uint32(max_samples),
        C.c_uint64(seed), C.c_uint64(scene_id_base), C.c_uint32(schedule[0]), C.c_uint32(schedule[1]), C.c_uint32(schedule[2]),
        _ptr(hits, C.c_uint32), _ptr(used, C.c_uint32),
        C.c_void_p(rows.ctypes.data))
    return hits, used, rows, int(total)


def sample_scenes(poses, std_devs, robot_w, robot_h, spread, seed, scene_id_base, n):
    poses = np.ascontiguousarray(poses, dtype=POSE_DT)
    std_devs = np.ascontiguousarray(std_devs, dtype=STD_DT)
    scenes = np.empty(n, SCENE_DT)
    lib().c2d_oracle_sample_scenes(
        C.c_void_p(poses.ctypes.data), C.c_uint32(len(poses)), C.c_void_p(std_devs.ctypes.data),
        C.c_uint32(len(std_devs)), C.c_float(robot_w), C.c_float(robot_h), C.c_float(spread),
        C.c_uint64(seed), C.c_uint64(scene_id_base), C.c_size_t(n), C.c_void_p(scenes.ctypes.data))
    return scenes


def math_eval(fn: int, in_bits):
    """Canonical math over an array of raw 32-bit inputs (fn as include/c2d.h C2D_MATH_*)."""
    in_bits = np.ascontiguousarray(in_bits, dtype=np.uint32)
    out0 = np.empty(in_bits.shape, np.float32)
    out1 = np.empty(in_bits.shape, np.float32)
    lib().c2d_oracle_math_eval(C.c_int(fn), _ptr(in_bits, C.c_uint32), C.c_size_t(in_bits.size), _ptr(out0), _ptr(out1))
    return out0, out1
