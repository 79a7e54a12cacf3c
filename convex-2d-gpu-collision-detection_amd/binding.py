"""ctypes mirror of include/c2d.h.

Every method maps one-to-one onto a C-ABI entry point; argument names and
meaning follow the header (which cites the reference lines each call replaces).
Device pointers are plain integers, so buffers may come from ``Engine.malloc``
or from any other allocator on the same device (e.g. ``torch.Tensor.data_ptr()``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
KMAX = 16

POSE_DT = np.dtype([("width", "<f4"), ("height", "<f4"), ("theta", "<f4")])
STD_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("theta", "<f4"), ("width", "<f4"), ("height", "<f4")])
SCENE_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("var_idx", "<f4"), ("pose_idx", "<f4")])
ROW_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("cp", "<f4"), ("var_idx", "<f4"), ("pose_idx", "<f4")])


# polygon Monte-Carlo (include/c2d.h, "Monte-Carlo collision probability for convex polygons")
POLY_DT = np.dtype([("k", "<u4"), ("x", "<f4", (KMAX,)), ("y", "<f4", (KMAX,))])       # c2d_polygon
POLY_POSE_DT = np.dtype([("theta", "<f4"), ("obstacle", POLY_DT)])                       # c2d_poly_pose


class C2DError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        self.status = status
        super().__init__(f"{what}: status {status}" + (f" ({detail})" if detail else ""))


class _Position(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class _Pose(C.Structure):
    _fields_ = [("width", C.c_float), ("height", C.c_float), ("theta", C.c_float)]


class _StdDev(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("theta", C.c_float), ("width", C.c_float), ("height", C.c_float)]


class _DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("device", C.c_int), ("compute_units", C.c_int),
                ("wavefront_size", C.c_int), ("lds_bytes_per_cu", C.c_int), ("hbm_bytes", C.c_size_t), ("pci_bus_id", C.c_char * 32)]


class _PolyBin(C.Structure):
    _fields_ = [("rows_a", C.c_uint32), ("rows_b", C.c_uint32), ("n", C.c_size_t), ("stride", C.c_size_t),
                ("d_ax", C.c_void_p), ("d_ay", C.c_void_p), ("d_bx", C.c_void_p), ("d_by", C.c_void_p),
                ("d_ka", C.c_void_p), ("d_kb", C.c_void_p), ("d_out", C.c_void_p)]


class _McScenesArgs(C.Structure):
    _fields_ = [
        ("d_poses", C.c_void_p), ("num_poses", C.c_uint32),
        ("d_std_devs", C.c_void_p), ("num_std_devs", C.c_uint32),
        ("d_scenes", C.c_void_p), ("n_scenes", C.c_size_t),
        ("robot_w", C.c_float), ("robot_h", C.c_float),
        ("accuracy_bins", C.POINTER(C.c_float)), ("bin_accuracy", C.POINTER(C.c_float)),
        ("n_accuracy_bins", C.c_uint32), ("max_samples", C.c_uint32),
        ("seed", C.c_uint64), ("scene_id_base", C.c_uint64),
        ("schedule_small_batch", C.c_uint32), ("schedule_large_batch", C.c_uint32), ("schedule_switch_at", C.c_uint32),
        ("d_hits", C.c_void_p), ("d_n_used", C.c_void_p), ("d_rows", C.c_void_p),
        ("total_samples", C.POINTER(C.c_uint64)), ("iterations", C.POINTER(C.c_uint32)),
    ]


class _Polygon(C.Structure):
    _fields_ = [("k", C.c_uint32), ("x", C.c_float * KMAX), ("y", C.c_float * KMAX)]


class _McPolyScenesArgs(C.Structure):
    _fields_ = [("base", _McScenesArgs), ("robot", C.POINTER(_Polygon)), ("d_poly_poses", C.c_void_p), ("num_poly_poses", C.c_uint32)]


def make_polygon(xs, ys=None) -> _Polygon:
    """A c2d_polygon from vertex coordinates (xs, ys), or from one POLY_DT record."""
    if isinstance(xs, _Polygon):
        return xs
    if ys is None:
        rec = xs
        k = int(rec["k"])
        p = _Polygon()
        p.k = k  # (as given: the library validates it)
        for i in range(KMAX):
            p.x[i], p.y[i] = float(rec["x"][i]), float(rec["y"][i])
        return p
    xs, ys = np.asarray(xs, np.float32), np.asarray(ys, np.float32)
    if not (len(xs) == len(ys) <= KMAX):
        raise ValueError("a polygon has at most %d vertices" % KMAX)
    p = _Polygon()
    p.k = len(xs)
    for i in range(len(xs)):
        p.x[i], p.y[i] = float(xs[i]), float(ys[i])
    return p


def library_path() -> str:
    """lib/libc2d.so next to this file; C2D_LIBRARY overrides it (to try another build of the same C-ABI)."""
    return os.environ.get("C2D_LIBRARY") or os.path.join(_HERE, "lib", "libc2d.so")


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); also the list of symbols the header declares
_SIGNATURES = {
    "c2d_version": (C.c_int, []),
    "c2d_status_string": (C.c_char_p, [C.c_int]),
    "c2d_last_error": (C.c_char_p, [C.c_void_p]),
    "c2d_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "c2d_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "c2d_ctx_destroy": (C.c_int, [C.c_void_p]),
    "c2d_ctx_info": (C.c_int, [C.c_void_p, C.POINTER(_DeviceInfo)]),   # (the 0.4 layout: kept for old binaries, not called here)
    "c2d_ctx_info_sized": (C.c_int, [C.c_void_p, C.POINTER(_DeviceInfo), C.c_size_t]),
    "c2d_ctx_check_async": (C.c_int, [C.c_void_p]),
    "c2d_malloc": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "c2d_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "c2d_malloc_host": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "c2d_free_host": (C.c_int, [C.c_void_p, C.c_void_p]),
    "c2d_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "c2d_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "c2d_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "c2d_stream_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "c2d_stream_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "c2d_stream_synchronize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "c2d_rects_from_poses": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p]),
    "c2d_sat_rect_pairs_verts": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_sat_rect_pairs_verts_mask": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_sat_rect_pairs_aos": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_sat_rect_pairs_pose": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_sat_rect_pairs_verts_host": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "c2d_sat_rect_pairs_pose_host": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "c2d_sat_poly_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_sat_poly_pairs_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_poly_bins_create": (C.c_int, [C.c_void_p, C.POINTER(_PolyBin), C.c_size_t, C.POINTER(C.c_void_p)]),
    "c2d_poly_bins_from_padded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "c2d_poly_bins_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "c2d_poly_bins_size": (C.c_size_t, [C.c_void_p]),
    "c2d_poly_bins_pairs": (C.c_size_t, [C.c_void_p]),
    "c2d_poly_bins_bytes": (C.c_size_t, [C.c_void_p]),
    "c2d_poly_bins_get": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(_PolyBin)]),
    "c2d_sat_poly_pairs_binned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_poly_bins_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_philox_normals": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_math_eval": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c2d_mc_pair": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.POINTER(_Position), C.POINTER(_Pose), C.POINTER(_StdDev),
                              C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "c2d_mc_scenes": (C.c_int, [C.c_void_p, C.POINTER(_McScenesArgs), C.c_void_p]),
    "c2d_mc_poly_pair": (C.c_int, [C.c_void_p, C.POINTER(_Polygon), C.POINTER(_Position), C.c_float, C.POINTER(_Polygon), C.POINTER(_StdDev),
                                   C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "c2d_mc_poly_scenes": (C.c_int, [C.c_void_p, C.POINTER(_McPolyScenesArgs), C.c_void_p]),
    "c2d_sample_scenes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                    C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p]),
    "c2d_uniform_table_minstd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint64, C.c_void_p]),
    "c2d_sqrt_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "c2d_dist_unique_id": (C.c_int, [C.c_void_p]),
    "c2d_dist_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "c2d_dist_init_file": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_double, C.POINTER(C.c_void_p)]),
    "c2d_dist_rank": (C.c_int, [C.c_void_p]),
    "c2d_dist_world_size": (C.c_int, [C.c_void_p]),
    "c2d_dist_transport": (C.c_char_p, [C.c_void_p]),
    "c2d_dist_rccl_version": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "c2d_dist_all_reduce_sum_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "c2d_dist_broadcast_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "c2d_dist_barrier": (C.c_int, [C.c_void_p, C.c_void_p]),
    "c2d_dist_stream_synchronize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "c2d_dist_timed_out": (C.c_int, [C.c_void_p]),
    "c2d_dist_destroy": (C.c_int, [C.c_void_p]),
    "c2d_calc_slack": (C.c_float, [C.c_uint32, C.c_uint32]),
    "c2d_get_bin": (C.c_int, [C.c_float, C.POINTER(C.c_float), C.c_uint32]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _open_library(path: str) -> C.CDLL:
    if not os.path.exists(path):
        raise C2DError(-3, "libc2d.so not built", f"run `make lib` or __graft_entry__.build(); expected {path}")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def load_library() -> C.CDLL:
    """dlopen lib/libc2d.so and type every entry point.  Raises if the HIP
    library has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        _lib = _open_library(library_path())
    return _lib


class DeviceArray:
    """A device allocation owned by an Engine, with numpy-like metadata."""

    def __init__(self, eng: "Engine", ptr: int, shape, dtype):
        self.eng, self.ptr, self.shape, self.dtype = eng, ptr, tuple(shape), np.dtype(dtype)

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    def row(self, i: int) -> int:
        """device pointer of sub-array [i] of a C-contiguous array"""
        inner = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        return self.ptr + i * inner

    def get(self, stream: int = 0) -> np.ndarray:
        return self.eng.to_host(self, stream)

    def free(self):
        if self.ptr:
            self.eng.free(self.ptr)
            self.ptr = 0


DIST_ID_BYTES = 128


class Dist:
    """One c2d_dist: this rank's end of the RCCL communicator that sums the hit counters
    (include/c2d.h, multi-GPU block).  Collective calls: every rank must make them."""

    def __init__(self, eng: "Engine", handle):
        self.eng, self.h = eng, handle

    @property
    def rank(self) -> int:
        return int(self.eng.lib.c2d_dist_rank(self.h))

    @property
    def world_size(self) -> int:
        return int(self.eng.lib.c2d_dist_world_size(self.h))

    @property
    def transport(self) -> str:
        return self.eng.lib.c2d_dist_transport(self.h).decode()

    def all_reduce_sum_u64(self, buf, count: int, stream: int = 0):
        self.eng._check(self.eng.lib.c2d_dist_all_reduce_sum_u64(self.h, C.c_void_p(_ptr_of(buf)), count, C.c_void_p(stream)),
                        "c2d_dist_all_reduce_sum_u64")

    def broadcast_u64(self, buf, count: int, root: int = 0, stream: int = 0):
        self.eng._check(self.eng.lib.c2d_dist_broadcast_u64(self.h, C.c_void_p(_ptr_of(buf)), count, root, C.c_void_p(stream)),
                        "c2d_dist_broadcast_u64")

    def barrier(self, stream: int = 0):
        self.eng._check(self.eng.lib.c2d_dist_barrier(self.h, C.c_void_p(stream)), "c2d_dist_barrier")

    def synchronize(self, stream: int = 0):
        """wait for the collectives queued on `stream` under the watchdog (c2d_dist_stream_synchronize)"""
        self.eng._check(self.eng.lib.c2d_dist_stream_synchronize(self.h, C.c_void_p(stream)), "c2d_dist_stream_synchronize")

    @property
    def timed_out(self) -> bool:
        return bool(self.eng.lib.c2d_dist_timed_out(self.h))

    def close(self):
        if self.h:
            self.eng.lib.c2d_dist_destroy(self.h)
            self.h = None


class PolyBins:
    """One c2d_poly_bins handle: the launch table of a binned polygon batch."""

    def __init__(self, eng: "Engine", handle):
        self.eng, self.h = eng, handle

    def __len__(self) -> int:
        return int(self.eng.lib.c2d_poly_bins_size(self.h))

    @property
    def pairs(self) -> int:
        return int(self.eng.lib.c2d_poly_bins_pairs(self.h))

    @property
    def bytes(self) -> int:
        return int(self.eng.lib.c2d_poly_bins_bytes(self.h))

    def get(self, i: int) -> dict:
        b = _PolyBin()
        self.eng._check(self.eng.lib.c2d_poly_bins_get(self.h, i, C.byref(b)), "c2d_poly_bins_get")
        return {"rows_a": b.rows_a, "rows_b": b.rows_b, "n": b.n, "stride": b.stride, "ax": b.d_ax or 0, "ay": b.d_ay or 0, "bx": b.d_bx or 0,
                "by": b.d_by or 0, "ka": b.d_ka or 0, "kb": b.d_kb or 0, "out": b.d_out or 0}

    def results(self, out, stream: int = 0):
        """results in the order of the padded input (handles made by poly_bins_from_padded)"""
        self.eng._check(self.eng.lib.c2d_poly_bins_results(self.eng.h, self.h, _ptr_of(out), C.c_void_p(stream)), "c2d_poly_bins_results")

    def close(self):
        if self.h:
            self.eng.lib.c2d_poly_bins_destroy(self.eng.h, self.h)
            self.h = None


def _ptr_of(x) -> int:
    if isinstance(x, DeviceArray):
        return x.ptr
    if x is None:
        return 0
    return int(x)


class Engine:
    """One c2d_ctx (one device)."""

    def __init__(self, device: int = 0, lib_path: Optional[str] = None):
        # lib_path: another build of the same C-ABI (validation builds such as lib/libc2d_fmad1.so)
        self.lib = _open_library(lib_path) if lib_path else load_library()
        h = C.c_void_p()
        st = self.lib.c2d_ctx_create(device, C.byref(h))
        if st != 0:
            raise C2DError(st, "c2d_ctx_create", self.lib.c2d_status_string(st).decode())
        self.h = h
        self.device = device

    # -- plumbing -----------------------------------------------------------
    def _check(self, st: int, what: str):
        if st != 0:
            raise C2DError(st, what, self.lib.c2d_last_error(self.h).decode() or self.lib.c2d_status_string(st).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.c2d_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        di = _DeviceInfo()
        self._check(self.lib.c2d_ctx_info_sized(self.h, C.byref(di), C.sizeof(di)), "c2d_ctx_info_sized")   # told how large THIS mirror of the struct is
        return {"name": di.name.decode(), "arch": di.arch.decode(), "device": di.device, "compute_units": di.compute_units,
                "wavefront_size": di.wavefront_size, "lds_bytes_per_cu": di.lds_bytes_per_cu, "hbm_bytes": di.hbm_bytes,
                "pci_bus_id": di.pci_bus_id.decode()}

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self.lib.c2d_malloc(self.h, C.byref(p), nbytes), "c2d_malloc")
        return p.value or 0

    def free(self, ptr: int):
        self._check(self.lib.c2d_free(self.h, C.c_void_p(ptr)), "c2d_free")

    def memset(self, ptr, value: int, nbytes: int, stream: int = 0):
        self._check(self.lib.c2d_memset(self.h, C.c_void_p(_ptr_of(ptr)), value, nbytes, C.c_void_p(stream)), "c2d_memset")

    def synchronize(self, stream: int = 0):
        self._check(self.lib.c2d_stream_synchronize(self.h, C.c_void_p(stream)), "c2d_stream_synchronize")

    # -- multi-GPU ----------------------------------------------------------
    def dist_rccl_version(self):
        """(ncclGetVersion code, path of the librccl that c2d_dist loaded) — c2d_dist_rccl_version"""
        v = C.c_int(0)
        buf = C.create_string_buffer(1024)
        st = self.lib.c2d_dist_rccl_version(C.byref(v), buf, 1024)
        if st != 0:
            raise C2DError(st, "c2d_dist_rccl_version", self.lib.c2d_status_string(st).decode())
        return int(v.value), buf.value.decode()

    def dist_unique_id(self) -> bytes:
        buf = C.create_string_buffer(DIST_ID_BYTES)
        st = self.lib.c2d_dist_unique_id(buf)
        if st != 0:
            raise C2DError(st, "c2d_dist_unique_id", self.lib.c2d_status_string(st).decode())
        return buf.raw

    def dist_init(self, rank: int, world_size: int, unique_id: bytes) -> Dist:
        if len(unique_id) != DIST_ID_BYTES:
            raise ValueError("the communicator id has 128 bytes")
        h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, DIST_ID_BYTES)
        self._check(self.lib.c2d_dist_init(self.h, rank, world_size, buf, C.byref(h)), "c2d_dist_init")
        return Dist(self, h)

    def dist_init_file(self, rank: int, world_size: int, path: str, timeout_s: float = 300.0) -> Dist:
        h = C.c_void_p()
        self._check(self.lib.c2d_dist_init_file(self.h, rank, world_size, path.encode(), timeout_s, C.byref(h)), "c2d_dist_init_file")
        return Dist(self, h)

    def check_async(self):
        """Raise if a kernel of this ctx reported an argument error since the last check (c2d_ctx_check_async)."""
        self._check(self.lib.c2d_ctx_check_async(self.h), "c2d_ctx_check_async")

    def stream_create(self) -> int:
        s = C.c_void_p()
        self._check(self.lib.c2d_stream_create(self.h, C.byref(s)), "c2d_stream_create")
        return s.value or 0

    def stream_destroy(self, stream: int):
        self._check(self.lib.c2d_stream_destroy(self.h, C.c_void_p(stream)), "c2d_stream_destroy")

    def empty(self, shape, dtype) -> DeviceArray:
        if isinstance(shape, int):
            shape = (shape,)
        a = DeviceArray(self, 0, shape, dtype)
        a.ptr = self.malloc(max(a.nbytes, 1))
        return a

    def zeros(self, shape, dtype, stream: int = 0) -> DeviceArray:
        a = self.empty(shape, dtype)
        self.memset(a.ptr, 0, max(a.nbytes, 1), stream)
        return a

    def to_device(self, host: np.ndarray, stream: int = 0) -> DeviceArray:
        host = np.ascontiguousarray(host)
        a = self.empty(host.shape, host.dtype)
        if host.nbytes:
            self._check(self.lib.c2d_memcpy_h2d(self.h, C.c_void_p(a.ptr), C.c_void_p(host.ctypes.data), host.nbytes,
                                                C.c_void_p(stream)), "c2d_memcpy_h2d")
            self.synchronize(stream)
        return a

    def to_host(self, a: DeviceArray, stream: int = 0) -> np.ndarray:
        out = np.empty(a.shape, a.dtype)
        if out.nbytes:
            self._check(self.lib.c2d_memcpy_d2h(self.h, C.c_void_p(out.ctypes.data), C.c_void_p(a.ptr), out.nbytes,
                                                C.c_void_p(stream)), "c2d_memcpy_d2h")
            self.synchronize(stream)
        return out

    def read(self, ptr: int, shape, dtype, stream: int = 0) -> np.ndarray:
        return self.to_host(DeviceArray(self, ptr, shape, dtype), stream)

    # -- geometry -------------------------------------------------------------
    def rects_from_poses(self, cx, cy, w, h, theta, n: int, out_planes: Sequence, stream: int = 0):
        arr = (C.c_void_p * 8)(*[_ptr_of(p) for p in out_planes])
        self._check(self.lib.c2d_rects_from_poses(self.h, _ptr_of(cx), _ptr_of(cy), _ptr_of(w), _ptr_of(h), _ptr_of(theta),
                                                  n, arr, C.c_void_p(stream)), "c2d_rects_from_poses")

    def sat_rect_pairs_verts(self, planes: Sequence, n: int, out, count=None, stream: int = 0):
        if len(planes) != 16:
            raise ValueError("need 16 vertex planes")
        arr = (C.c_void_p * 16)(*[_ptr_of(p) for p in planes])
        self._check(self.lib.c2d_sat_rect_pairs_verts(self.h, arr, n, _ptr_of(out), _ptr_of(count), C.c_void_p(stream)),
                    "c2d_sat_rect_pairs_verts")

    def sat_rect_pairs_aos(self, r1, r2, n: int, out, count=None, stream: int = 0):
        self._check(self.lib.c2d_sat_rect_pairs_aos(self.h, _ptr_of(r1), _ptr_of(r2), n, _ptr_of(out), _ptr_of(count),
                                                    C.c_void_p(stream)), "c2d_sat_rect_pairs_aos")

    def sat_rect_pairs_verts_mask(self, planes: Sequence, n: int, mask, count=None, stream: int = 0):
        if len(planes) != 16:
            raise ValueError("need 16 vertex planes")
        arr = (C.c_void_p * 16)(*[_ptr_of(p) for p in planes])
        self._check(self.lib.c2d_sat_rect_pairs_verts_mask(self.h, arr, n, _ptr_of(mask), _ptr_of(count), C.c_void_p(stream)),
                    "c2d_sat_rect_pairs_verts_mask")

    def sat_rect_pairs_pose(self, planes: Sequence, n: int, out, count=None, stream: int = 0):
        if len(planes) != 10:
            raise ValueError("need 10 pose planes")
        arr = (C.c_void_p * 10)(*[_ptr_of(p) for p in planes])
        self._check(self.lib.c2d_sat_rect_pairs_pose(self.h, arr, n, _ptr_of(out), _ptr_of(count), C.c_void_p(stream)),
                    "c2d_sat_rect_pairs_pose")

    def sat_rect_pairs_host(self, planes, out: np.ndarray, fmt: str = "verts") -> int:
        """c2d_sat_rect_pairs_verts_host / _pose_host: planes = 16 (10) host pointers or 1-D float32 arrays of n elements each, out = host u8[n]
        (numpy arrays or the arrays of host_empty()); returns the number of colliding pairs"""
        want = 16 if fmt == "verts" else 10
        if len(planes) != want:
            raise ValueError("need %d planes" % want)
        n = out.shape[0]
        ptrs = [(p.ctypes.data if isinstance(p, np.ndarray) else int(p)) for p in planes]
        arr = (C.c_void_p * want)(*ptrs)
        cnt = C.c_ulonglong(0)
        fn = self.lib.c2d_sat_rect_pairs_verts_host if fmt == "verts" else self.lib.c2d_sat_rect_pairs_pose_host
        self._check(fn(self.h, arr, n, C.c_void_p(out.ctypes.data), C.byref(cnt)), "c2d_sat_rect_pairs_%s_host" % fmt)
        return int(cnt.value)

    def host_empty(self, shape, dtype) -> np.ndarray:
        """a numpy array in page-locked host memory (c2d_malloc_host); free it with host_free(array)"""
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        p = C.c_void_p()
        self._check(self.lib.c2d_malloc_host(self.h, C.byref(p), max(nbytes, 1)), "c2d_malloc_host")
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, a: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        if p:
            self._check(self.lib.c2d_free_host(self.h, C.c_void_p(p)), "c2d_free_host")

    def sat_poly_pairs_rows(self, vx, vy, k, n: int, rows: int, out, count=None, stream: int = 0):
        self._check(self.lib.c2d_sat_poly_pairs_rows(self.h, _ptr_of(vx), _ptr_of(vy), _ptr_of(k), n, rows, _ptr_of(out), _ptr_of(count),
                                                     C.c_void_p(stream)), "c2d_sat_poly_pairs_rows")

    def sat_poly_pairs(self, vx, vy, k, n: int, out, count=None, stream: int = 0):
        self._check(self.lib.c2d_sat_poly_pairs(self.h, _ptr_of(vx), _ptr_of(vy), _ptr_of(k), n, _ptr_of(out), _ptr_of(count),
                                                C.c_void_p(stream)), "c2d_sat_poly_pairs")

    # -- binned polygon batches (include/c2d.h "binned polygon batches") ---------------
    def poly_bins_create(self, bins) -> "PolyBins":
        """bins: sequence of dicts with rows_a, rows_b, n, ax, ay, bx, by, out and optionally ka, kb, stride
        (device pointers or DeviceArrays)."""
        arr = (_PolyBin * max(len(bins), 1))()
        for i, b in enumerate(bins):
            arr[i] = _PolyBin(b["rows_a"], b["rows_b"], b["n"], b.get("stride", 0), _ptr_of(b["ax"]), _ptr_of(b["ay"]), _ptr_of(b["bx"]),
                              _ptr_of(b["by"]), _ptr_of(b.get("ka")), _ptr_of(b.get("kb")), _ptr_of(b["out"]))
        h = C.c_void_p()
        self._check(self.lib.c2d_poly_bins_create(self.h, arr, len(bins), C.byref(h)), "c2d_poly_bins_create")
        return PolyBins(self, h)

    def poly_bins_from_padded(self, vx, vy, k, n: int, rows: int, granularity: int, stream: int = 0) -> "PolyBins":
        h = C.c_void_p()
        st = self.lib.c2d_poly_bins_from_padded(self.h, _ptr_of(vx), _ptr_of(vy), _ptr_of(k), n, rows, granularity, C.byref(h), C.c_void_p(stream))
        if st != 0:
            if h.value:
                self.lib.c2d_poly_bins_destroy(self.h, h)
            self._check(st, "c2d_poly_bins_from_padded")
        return PolyBins(self, h)

    def sat_poly_pairs_binned(self, bins: "PolyBins", count=None, stream: int = 0):
        self._check(self.lib.c2d_sat_poly_pairs_binned(self.h, bins.h, _ptr_of(count), C.c_void_p(stream)), "c2d_sat_poly_pairs_binned")

    # -- random stream / Monte-Carlo ------------------------------------------------
    def philox_normals(self, seed: int, scene_id: int, sample_begin: int, n: int, normals, raw=None, stream: int = 0):
        self._check(self.lib.c2d_philox_normals(self.h, seed, scene_id, sample_begin, n, _ptr_of(normals), _ptr_of(raw),
                                                C.c_void_p(stream)), "c2d_philox_normals")

    MATH_LOG, MATH_SINCOS, MATH_SINCOS_U32, MATH_SQRT, MATH_BOX_MULLER = range(5)

    def math_eval(self, fn: int, in_bits, n: int, out0, out1=None, stream: int = 0):
        self._check(self.lib.c2d_math_eval(self.h, fn, _ptr_of(in_bits), n, _ptr_of(out0), _ptr_of(out1), C.c_void_p(stream)),
                    "c2d_math_eval")

    def mc_pair(self, robot_w, robot_h, pos, pose, std_dev, seed, scene_id, sample_begin, n_samples, hits, stream: int = 0):
        self._check(self.lib.c2d_mc_pair(self.h, robot_w, robot_h, C.byref(_Position(*pos)), C.byref(_Pose(*pose)),
                                         C.byref(_StdDev(*std_dev)), seed, scene_id, sample_begin, n_samples,
                                         _ptr_of(hits), C.c_void_p(stream)), "c2d_mc_pair")

    def mc_scenes(self, poses, num_poses, std_devs, num_std_devs, scenes, n_scenes, robot_w, robot_h, accuracy_bins,
                  bin_accuracy, max_samples, seed, scene_id_base, hits, n_used, rows=None, stream: int = 0,
                  schedule=(0, 0, 0)):
        bins = np.ascontiguousarray(accuracy_bins, dtype=np.float32)
        acc = np.ascontiguousarray(bin_accuracy, dtype=np.float32)
        if len(acc) != len(bins) - 1:
            raise ValueError("bin_accuracy must have len(accuracy_bins) - 1 entries")
        total, iters = C.c_uint64(0), C.c_uint32(0)
        a = _McScenesArgs(_ptr_of(poses), num_poses, _ptr_of(std_devs), num_std_devs, _ptr_of(scenes), n_scenes, robot_w,
                          robot_h, bins.ctypes.data_as(C.POINTER(C.c_float)), acc.ctypes.data_as(C.POINTER(C.c_float)),
                          len(bins), max_samples, seed, scene_id_base, schedule[0], schedule[1], schedule[2],
                          _ptr_of(hits), _ptr_of(n_used), _ptr_of(rows),
                          C.pointer(total), C.pointer(iters))
        self._check(self.lib.c2d_mc_scenes(self.h, C.byref(a), C.c_void_p(stream)), "c2d_mc_scenes")
        return int(total.value), int(iters.value)

    def mc_scenes_async(self, poses, num_poses, std_devs, num_std_devs, scenes, n_scenes, robot_w, robot_h, accuracy_bins,
                        bin_accuracy, max_samples, seed, scene_id_base, hits, n_used, rows=None, stream: int = 0,
                        schedule=(0, 0, 0)):
        """c2d_mc_scenes without host outputs: the whole adaptive loop is only enqueued (no synchronisation)."""
        bins = np.ascontiguousarray(accuracy_bins, dtype=np.float32)
        acc = np.ascontiguousarray(bin_accuracy, dtype=np.float32)
        if len(acc) != len(bins) - 1:
            raise ValueError("bin_accuracy must have len(accuracy_bins) - 1 entries")
        a = _McScenesArgs(_ptr_of(poses), num_poses, _ptr_of(std_devs), num_std_devs, _ptr_of(scenes), n_scenes, robot_w,
                          robot_h, bins.ctypes.data_as(C.POINTER(C.c_float)), acc.ctypes.data_as(C.POINTER(C.c_float)),
                          len(bins), max_samples, seed, scene_id_base, schedule[0], schedule[1], schedule[2],
                          _ptr_of(hits), _ptr_of(n_used), _ptr_of(rows), None, None)
        self._check(self.lib.c2d_mc_scenes(self.h, C.byref(a), C.c_void_p(stream)), "c2d_mc_scenes")

    def mc_poly_pair(self, robot, pos, robot_theta, obstacle, std_dev, seed, scene_id, sample_begin, n_samples, hits, stream: int = 0):
        """robot, obstacle: (xs, ys) tuples, POLY_DT records or make_polygon() results"""
        r = make_polygon(*robot) if isinstance(robot, tuple) else make_polygon(robot)
        o = make_polygon(*obstacle) if isinstance(obstacle, tuple) else make_polygon(obstacle)
        self._check(self.lib.c2d_mc_poly_pair(self.h, C.byref(r), C.byref(_Position(*pos)), robot_theta, C.byref(o), C.byref(_StdDev(*std_dev)),
                                              seed, scene_id, sample_begin, n_samples, _ptr_of(hits), C.c_void_p(stream)), "c2d_mc_poly_pair")

    def mc_poly_scenes(self, robot, poly_poses, num_poly_poses, std_devs, num_std_devs, scenes, n_scenes, accuracy_bins, bin_accuracy, max_samples,
                       seed, scene_id_base, hits, n_used, rows=None, stream: int = 0, schedule=(0, 0, 0), host_outputs: bool = True):
        """c2d_mc_poly_scenes; with host_outputs=False the loop is only enqueued (no synchronisation) and None is returned"""
        bins = np.ascontiguousarray(accuracy_bins, dtype=np.float32)
        acc = np.ascontiguousarray(bin_accuracy, dtype=np.float32)
        if len(acc) != len(bins) - 1:
            raise ValueError("bin_accuracy must have len(accuracy_bins) - 1 entries")
        total, iters = C.c_uint64(0), C.c_uint32(0)
        base = _McScenesArgs(None, 0, _ptr_of(std_devs), num_std_devs, _ptr_of(scenes), n_scenes, 0.0, 0.0,
                             bins.ctypes.data_as(C.POINTER(C.c_float)), acc.ctypes.data_as(C.POINTER(C.c_float)),
                             len(bins), max_samples, seed, scene_id_base, schedule[0], schedule[1], schedule[2],
                             _ptr_of(hits), _ptr_of(n_used), _ptr_of(rows),
                             C.pointer(total) if host_outputs else None, C.pointer(iters) if host_outputs else None)
        r = None if robot is None else (make_polygon(*robot) if isinstance(robot, tuple) else make_polygon(robot))
        a = _McPolyScenesArgs(base, C.pointer(r) if r is not None else None, _ptr_of(poly_poses), num_poly_poses)
        self._check(self.lib.c2d_mc_poly_scenes(self.h, C.byref(a), C.c_void_p(stream)), "c2d_mc_poly_scenes")
        return (int(total.value), int(iters.value)) if host_outputs else None

    def sample_scenes(self, poses, num_poses, std_devs, num_std_devs, robot_w, robot_h, spread, seed, scene_id_base,
                      n_scenes, scenes, stream: int = 0):
        self._check(self.lib.c2d_sample_scenes(self.h, _ptr_of(poses), num_poses, _ptr_of(std_devs), num_std_devs, robot_w,
                                               robot_h, spread, seed, scene_id_base, n_scenes, _ptr_of(scenes),
                                               C.c_void_p(stream)), "c2d_sample_scenes")

    def uniform_table_minstd(self, out, rows: int, dims: int, lo, hi, first_draw: int = 0, stream: int = 0):
        lo_, hi_ = np.ascontiguousarray(lo, dtype=np.float32), np.ascontiguousarray(hi, dtype=np.float32)
        self._check(self.lib.c2d_uniform_table_minstd(self.h, _ptr_of(out), rows, dims, lo_.ctypes.data_as(C.POINTER(C.c_float)),
                                                      hi_.ctypes.data_as(C.POINTER(C.c_float)), first_draw, C.c_void_p(stream)), "c2d_uniform_table_minstd")

    def sqrt_f32(self, src, dst, n: int, stream: int = 0):
        self._check(self.lib.c2d_sqrt_f32(self.h, _ptr_of(src), _ptr_of(dst), n, C.c_void_p(stream)), "c2d_sqrt_f32")

    def calc_slack(self, n: int, k: int) -> float:
        return float(self.lib.c2d_calc_slack(n, k))

    def get_bin(self, p: float, bins) -> int:
        b = np.ascontiguousarray(bins, dtype=np.float32)
        return int(self.lib.c2d_get_bin(p, b.ctypes.data_as(C.POINTER(C.c_float)), len(b)))
