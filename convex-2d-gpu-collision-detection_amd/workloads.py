"""Synthetic inputs for the BASELINE.json configurations (host-side numpy only).

Shapes and ranges follow SURVEY.md §8(d): rectangle sizes are drawn from the
reference's default pose range min_pose/max_pose = [0.1, 5] (generate_dataset.cu:56-57),
centres from U(-8, 8), angles from U(0, 2*pi).
"""
from __future__ import annotations

import numpy as np

KMAX = 16

# BASELINE config 3: one fixed robot + one Gaussian-pose obstacle
# (robot size = reference defaults, compute_collision_probability.cu:39-40).
MC_PAIR_SCENE = {
    "robot_w": 4.07, "robot_h": 1.74,
    "pos": (3.0, 1.0),
    "pose": (2.0, 1.0, 0.6),                  # width, height, theta
    "std_dev": (0.3, 0.3, 0.2, 0.0, 0.0),     # x, y, theta, width, height
}

DEFAULT_BINS = (0.0, 0.01, 0.1, 1.0)          # generate_dataset.cu:58
DEFAULT_BIN_ACCURACY = (1e-4, 1e-3, 1e-2)     # generate_dataset.cu:59


def random_obb_pose_planes(n: int, seed: int = 0x5A7, extent: float = 8.0) -> np.ndarray:
    """float32 [10][n]: (cx, cy, w, h, theta) of rectangle 1 then rectangle 2."""
    rng = np.random.Generator(np.random.Philox(seed))
    out = np.empty((10, n), np.float32)
    for r in range(2):
        out[5 * r + 0] = rng.uniform(-extent, extent, n)
        out[5 * r + 1] = rng.uniform(-extent, extent, n)
        out[5 * r + 2] = rng.uniform(0.1, 5.0, n)
        out[5 * r + 3] = rng.uniform(0.1, 5.0, n)
        out[5 * r + 4] = rng.uniform(0.0, 2.0 * np.pi, n)
    return out


def random_convex_polygons(n: int, seed: int = 0xC0FFEE, kmin: int = 3, kmax: int = KMAX, extent: float = 8.0, rows: int = KMAX):
    """BASELINE config 5 input: vx, vy float32 [2][rows][n] (rows = KMAX unless a tighter layout is asked for),
    k uint8 [2][n].  Each polygon has K ~ U{kmin..kmax} vertices at sorted random angles on an
    ellipse (hence convex, counter-clockwise), randomly rotated and placed."""
    assert kmax <= rows <= KMAX
    rng = np.random.Generator(np.random.Philox(seed))
    vx = np.zeros((2, rows, n), np.float32)
    vy = np.zeros((2, rows, n), np.float32)
    k = rng.integers(kmin, kmax + 1, size=(2, n)).astype(np.uint8)
    for p in range(2):
        mask = np.arange(rows)[:, None] >= k[p][None, :]
        ang = rng.uniform(0.0, 2.0 * np.pi, size=(rows, n))
        ang[mask] = np.inf                      # unused slots sort to the end
        ang = np.sort(ang, axis=0)              # the K used angles ascend: counter-clockwise
        ang[mask] = 0.0
        a = rng.uniform(0.3, 2.5, n)
        b = rng.uniform(0.3, 2.5, n)
        rot = rng.uniform(0.0, 2.0 * np.pi, n)
        cx = rng.uniform(-extent, extent, n)
        cy = rng.uniform(-extent, extent, n)
        x = a[None, :] * np.cos(ang)
        y = b[None, :] * np.sin(ang)
        c, s = np.cos(rot)[None, :], np.sin(rot)[None, :]
        vx[p] = (c * x - s * y + cx[None, :]).astype(np.float32)
        vy[p] = (s * x + c * y + cy[None, :]).astype(np.float32)
        vx[p][mask] = 0.0
        vy[p][mask] = 0.0
    return vx, vy, k


def random_tables(num_poses: int, num_variances: int, seed: int = 7, shape_variance: bool = False):
    """Pose and StdDev tables as generate_dataset builds them (generate_dataset.cu:282-332):
    variances ~ U(0, 0.3) per dimension (width/height forced to 0 unless
    shape_variance), std_dev = sqrt(variance); poses ~ U([0.1,0.1,0],[5,5,2pi])."""
    from .binding import POSE_DT, STD_DT

    rng = np.random.Generator(np.random.Philox(seed))
    var = rng.uniform(0.0, 0.3, size=(num_variances, 5)).astype(np.float32)
    if not shape_variance:
        var[:, 3:] = 0.0
    sd = np.sqrt(var).astype(np.float32)
    poses = np.empty((num_poses, 3), np.float32)
    poses[:, 0] = rng.uniform(0.1, 5.0, num_poses)
    poses[:, 1] = rng.uniform(0.1, 5.0, num_poses)
    poses[:, 2] = rng.uniform(0.0, 2.0 * np.pi, num_poses)
    return poses.view(POSE_DT).reshape(-1), sd.view(STD_DT).reshape(-1), var


def touching_pose_pairs(n: int, seed: int, scale: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """float32 [10][n] pose pairs (cx, cy, w, h, theta twice) built to TOUCH: the second rectangle's centre sits along one of the
    four frame directions of the pair at the distance where that direction's gap is zero, moved by 0 or a few parts in
    1e-7 .. 1e-3 either way, and anywhere along the direction's normal.  A third of the pairs have parallel or perpendicular
    frames.  The workload for the closed-form pair test (c2d_sat.hip pose_pair_closed_form, c2d_mc.hip model_gap): most of
    these pairs are decided inside or right at its margin."""
    rng = np.random.Generator(np.random.Philox(seed))
    F = np.float32
    w1, h1, w2, h2 = [(rng.uniform(0.2, 4, n) * scale).astype(F) for _ in range(4)]
    t1 = rng.uniform(-3.2, 3.2, n).astype(F)
    t2 = np.where(rng.random(n) < 0.3, t1 + rng.choice([0.0, np.pi / 2, np.pi], n), rng.uniform(-3.2, 3.2, n)).astype(F)
    x1, y1 = [((rng.uniform(-1, 1, n) + offset) * scale).astype(F) for _ in range(2)]
    s1, c1, s2, c2 = np.sin(t1.astype(np.float64)), np.cos(t1.astype(np.float64)), np.sin(t2.astype(np.float64)), np.cos(t2.astype(np.float64))
    which = rng.integers(0, 4, n)
    ex = np.choose(which, [c1, -s1, c2, -s2])
    ey = np.choose(which, [s1, c1, s2, c2])
    hw, hh, hx, hy = [np.abs(v.astype(np.float64)) / 2 for v in (w1, h1, w2, h2)]
    ext = hw * np.abs(ex * c1 + ey * s1) + hh * np.abs(-ex * s1 + ey * c1) + hx * np.abs(ex * c2 + ey * s2) + hy * np.abs(-ex * s2 + ey * c2)
    rel = rng.choice([0.0, 1e-7, -1e-7, 3e-7, -3e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3], n)
    dist = ext * (1 + rel)
    sign = rng.choice([-1.0, 1.0], n)
    lateral = rng.uniform(-0.3, 0.3, n) * scale
    x2 = (x1 + sign * dist * ex - lateral * ey).astype(F)
    y2 = (y1 + sign * dist * ey + lateral * ex).astype(F)
    return np.stack([x1, y1, w1, h1, t1, x2, y2, w2, h2, t2]).astype(F)


def inject_non_finite(values: np.ndarray, seed: int, frac: float = 0.5, axis_items: int = -1) -> np.ndarray:
    """Copy of `values` (float32 [planes][n], item index last) in which about `frac` of the items get one to three of
    their coordinates replaced by NaN, +inf, -inf or +-3e38 (large enough for products to overflow).  Used by the tests
    that pin the behaviour on non-finite vertices (include/c2d.h, "non-finite inputs")."""
    rng = np.random.Generator(np.random.Philox(seed))
    out = np.array(values, dtype=np.float32, copy=True)
    flat = out.reshape(-1, out.shape[axis_items])
    planes, n = flat.shape
    junk = np.array([np.nan, np.inf, -np.inf, 3e38, -3e38], np.float32)
    victims = np.flatnonzero(rng.random(n) < frac)
    for rep in range(3):
        sel = victims[rng.random(victims.size) < (1.0 if rep == 0 else 0.4)]
        flat[rng.integers(0, planes, sel.size), sel] = rng.choice(junk, sel.size)
    return out


def convex_polygon(k: int, rng, a: float = 1.0, b: float = 1.0, rot: float = 0.0, clockwise: bool = False):
    """k vertices at sorted random angles on an ellipse with half axes a, b about the origin, rotated by rot: float32 xs, ys"""
    ang = np.sort(rng.uniform(0.0, 2.0 * np.pi, k))
    if clockwise:
        ang = ang[::-1]
    x, y = a * np.cos(ang), b * np.sin(ang)
    c, s = np.cos(rot), np.sin(rot)
    return (c * x - s * y).astype(np.float32), (s * x + c * y).astype(np.float32)


def rect_polygon(w: float, h: float):
    """create_rect (utils.cu:119-130) as a 4-gon: the same floats in the same order"""
    w, h = np.float32(w), np.float32(h)
    two = np.float32(2)
    return (np.array([-w / two, w / two, w / two, -w / two], np.float32), np.array([-h / two, -h / two, h / two, h / two], np.float32))


def near_regular_polygon(k: int, rng, a: float, b: float, rot: float = 0.0):
    """k vertices on an ellipse at angles 2 pi (i + jitter_i) / k, |jitter| < 0.3: a full-bodied convex polygon"""
    ang = 2.0 * np.pi * (np.arange(k) + rng.uniform(-0.3, 0.3, k)) / k
    x, y = a * np.cos(ang), b * np.sin(ang)
    c, s = np.cos(rot), np.sin(rot)
    return (c * x - s * y).astype(np.float32), (s * x + c * y).astype(np.float32)


# The polygon counterpart of MC_PAIR_SCENE: a robot polygon about the size of the reference's robot (4.07 x 1.74), an obstacle
# about the size of config 3's (2 x 1), the same pose noise; placed so that about half of the samples collide.
def mc_poly_pair_scene(ka: int = 7, kb: int = 5, seed: int = 11, pos=(2.8, 1.0)):
    rng = np.random.Generator(np.random.Philox(seed))
    return {
        "robot": near_regular_polygon(ka, rng, 4.07 / 2 * 1.15, 1.74 / 2 * 1.15),
        "pos": pos, "theta": 0.6,
        "obstacle": near_regular_polygon(kb, rng, 1.15, 0.6, rot=0.3),
        "std_dev": (0.3, 0.3, 0.2, 0.0, 0.0),
    }


def random_poly_tables(num_poses: int, num_variances: int, seed: int = 7, kmin: int = 3, kmax: int = KMAX, shape_variance: bool = False):
    """The polygon counterpart of random_tables: POLY_POSE_DT[num_poses] — robot rotation ~ U(0, 2 pi) and an obstacle polygon with
    K ~ U{kmin..kmax} vertices on an ellipse with half axes ~ U(0.05, 2.5) (the reference's obstacle sizes 0.1 .. 5,
    generate_dataset.cu:56-57) — and STD_DT[num_variances] as random_tables draws it (relative width / height deviations
    ~ sqrt(U(0, 0.02)) when shape_variance)."""
    from .binding import POLY_POSE_DT, STD_DT

    rng = np.random.Generator(np.random.Philox(seed))
    var = rng.uniform(0.0, 0.3, size=(num_variances, 5)).astype(np.float32)
    var[:, 3:] = rng.uniform(0.0, 0.02, size=(num_variances, 2)) if shape_variance else 0.0
    sd = np.sqrt(var).astype(np.float32)
    poses = np.zeros(num_poses, POLY_POSE_DT)
    poses["theta"] = rng.uniform(0.0, 2.0 * np.pi, num_poses)
    ks = rng.integers(kmin, kmax + 1, num_poses)
    for i in range(num_poses):
        x, y = convex_polygon(int(ks[i]), rng, rng.uniform(0.05, 2.5), rng.uniform(0.05, 2.5), rng.uniform(0, 2 * np.pi), clockwise=bool(rng.integers(0, 2)))
        poses["obstacle"]["k"][i] = ks[i]
        poses["obstacle"]["x"][i, :ks[i]] = x
        poses["obstacle"]["y"][i, :ks[i]] = y
    return poses, sd.view(STD_DT).reshape(-1)


def random_poly_scenes(n: int, poly_poses, std_devs, robot_radius: float, seed: int = 3, spread: float = 4.0):
    """SCENE_DT[n] rows for a polygon dataset, in the spirit of generate_dataset.cu:207-219: table indices uniform, the robot on a
    ring around the obstacle at about touching distance, shifted by a normal draw of the scene's position noise."""
    from .binding import SCENE_DT

    rng = np.random.Generator(np.random.Philox(seed))
    rows = np.zeros(n, SCENE_DT)
    pi_, vi = rng.integers(0, len(poly_poses), n), rng.integers(0, len(std_devs), n)
    ob = poly_poses["obstacle"]
    rad = np.sqrt(ob["x"] ** 2 + ob["y"] ** 2).max(axis=1)[pi_]
    sdm = (std_devs["x"][vi] + std_devs["y"][vi]) / 2
    dist = 0.7 * (rad + robot_radius) + sdm + rng.normal(0, 1, n) * sdm * spread / 4
    th = rng.uniform(0, 2 * np.pi, n)
    rows["x"], rows["y"] = dist * np.cos(th), dist * np.sin(th)
    rows["var_idx"], rows["pose_idx"] = vi, pi_
    return rows
