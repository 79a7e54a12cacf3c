"""convex-2d-gpu-collision-detection_amd — MI355X-native batched 2D SAT collision engine.

The product is ``lib/libc2d.so`` (hand-written HIP for gfx950 behind the C-ABI of
``include/c2d.h``) plus the C++ host drivers in ``csrc/``.  This Python package is
only the thin ctypes mirror of that C-ABI used by the test-suite and ``bench.py``;
it holds no compute of its own and raises if the HIP library is missing — there is
no CPU fallback.

The directory name contains hyphens (it is the reference's repository name), so
load it with ``__graft_entry__.load_package()`` rather than a plain ``import``.
"""
from .binding import (  # noqa: F401
    C2DError,
    Engine,
    Dist,
    PolyBins,
    DeviceArray,
    KMAX,
    POSE_DT,
    STD_DT,
    SCENE_DT,
    ROW_DT,
    POLY_DT,
    POLY_POSE_DT,
    make_polygon,
    library_path,
    load_library,
)
