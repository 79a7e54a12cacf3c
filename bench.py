#!/usr/bin/env python3
"""bench.py — headline benchmark of the c2d hot path on MI355X.

Metric (BASELINE.json): rectangle-pair SAT tests per second, whole job, inputs
resident in HBM, on BASELINE config 2: 10^7 random OBB pairs per GPU as 16 SoA
vertex planes -> u8 booleans (65 algorithmic bytes per pair).  One "step" = one
pass of c2d_sat_rect_pairs_verts over the rank's 10^7 pairs.  With N > 1 every
rank owns its own 10^7 pairs (weak scaling, no data-path collective) and one
RCCL all-reduce of the colliding-pair count closes the timed region.

The same JSON line also carries
  roofline     — HBM roofline of the SAT kernel (HIP events on its stream),
  cpu_baseline — the CPU oracle (OpenMP port of the reference arithmetic) timed on
                 a bounded sample of the same workload on this host, rank 0, N = 1,
  mc           — Monte-Carlo samples/s of BASELINE config 3 (1 scene, 10^8
                 samples per GPU, sample ranges sharded over ranks, one all-reduce
                 of the hit count), with its VALU-roofline note.

Launch: `python bench.py [--gpus N --steps K --warmup W]`.  With N > 1 and no launcher
environment (RANK / WORLD_SIZE unset) the process starts the N ranks itself — before it
imports torch or touches a GPU — as children of `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ...`, relays rank 0's JSON line and exits with
the children's status; launched under torch.distributed.run directly it is one of the ranks.

The reduce of each leg is the product's own: c2d_dist_all_reduce_sum_u64 of libc2d.so
(ncclAllReduce of RCCL over xGMI, include/c2d.h); torch.distributed provides rendezvous,
barriers and the max-over-ranks of the timings.  If the c2d communicator cannot be created, or
RCCL counts another number of ranks than the launcher, every rank exits non-zero: a scaling
curve of some other reduce is never reported (`--reduce torch` asks for torch's all_reduce
explicitly).

Parity in the same run (rank 0, N = 1): ALL booleans of the config-2 and config-5 batches are
compared with the CPU oracle, and the first scenes of the config-4 shard with the oracle's
adaptive loop (hits and sample counts); each of those legs carries its own `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PAIR = 65            # 16 planes x 4 B read + 1 B written
FP32_VALU_PEAK_TFLOPS = 157.3  # spec, FMA counted as 2
VALU_PEAK_TLANE = FP32_VALU_PEAK_TFLOPS / 2  # 10^12 VALU lane-instructions/s (one FMA lane = 2 flop)
KMAX = 16


def measured_counts():
    """PMC-measured per-unit instruction counts and traffic of the current kernels on the bench workloads
    (profiles/measured_counts.json, written from the rocprofv3 --pmc passes named in its `source` fields).
    They are properties of (kernel build, workload): a leg whose workload differs from the recorded one
    gets no VALU roofline instead of a wrong one."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "measured_counts.json")))
    except Exception:
        return {}


NOMINAL_GHZ = 2.4


def held_clock(roofline: dict, c: dict) -> None:
    """The VALU peak of MI355X_MICROARCH.md is priced at the nominal 2.4 GHz; under a dense VALU load the chip holds less.  Where the
    clock build of the kernel has recorded the clock its waves held (tests/tools/mc_clock.py -> measured_counts.json), the roofline
    also carries the peak and the fraction at THAT clock."""
    ghz = c.get("held_clock_ghz")
    if not ghz:
        return
    peak = roofline["peak"] * ghz / NOMINAL_GHZ
    roofline["held_clock_ghz"] = ghz
    roofline["peak_at_held_clock"] = round(peak, 2)
    roofline["frac_at_held_clock"] = round(roofline["achieved"] / peak, 4)
    roofline["held_clock_source"] = "recorded, not measured in this run: %s" % c.get("held_clock_source")


SIMDS = 256 * 4


def issue_weighted(roofline: dict, c: dict, wave_instr_per_s: float) -> None:
    """The 2-ticks-per-instruction VALU peak is not a roof any real mix reaches, and single-type instruction prices do not add
    (profiles/valu_issue.py).  Where the collection has measured the rate of this leg's OWN instruction mix as a dependency-free
    stream (csrc/tools/instr_probe, `mix <entry>` lines) and the clock build the clock the kernel holds, the roofline also carries
    the fraction of the issue ticks the chip had that this mix needs: wave instructions/s x ticks per wave instruction of the mix
    / (1024 SIMDs x held clock)."""
    ticks, ghz = c.get("issue_ticks_per_wave_instr"), c.get("held_clock_ghz")
    if not ticks or not ghz:
        return
    roofline["issue_ticks_per_wave_instr"] = ticks
    roofline["frac_issue_weighted"] = round(wave_instr_per_s * ticks / (SIMDS * ghz * 1e9), 4)
    roofline["issue_weighted_source"] = ("recorded, not measured in this run: the leg's instruction mix (per-type PMC counts x static mix of the loop code) as a "
                                         "dependency-free stream in csrc/tools/instr_probe, %s; clock: held_clock_ghz" % c.get("source"))


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh processes.  This parent has not
    imported torch and never touches a GPU; it relays rank 0's JSON line and the children's exit status."""
    import socket
    import subprocess

    # Starting the ranks is a fork + exec.  That is only safe from a process that has not initialised the GPU — this one has not, of
    # its own — but a profiler / tool library preloaded into it (rocprofv3 puts its own into every process it starts) has, before
    # main, and an exec from such a process is what this pool's hosts forbid.  Same rule as the drivers' --gpus N.
    for var in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_LIBRARY", "LD_PRELOAD"):
        val = os.environ.get(var, "")
        if val and (var != "LD_PRELOAD" or any(nm in val.lower() for nm in ("rocprof", "roctracer", "roctx", "libhsa", "libamdhip"))):
            sys.stderr.write(f"[bench] --gpus {n_gpus} refused: a profiler / tool library is preloaded into this process ({var}={val}); the "
                             "launcher would have to exec its ranks from a process that holds the GPU.  Profile ONE rank: `rocprofv3 ... -- "
                             "python3 bench.py` (N = 1), or one rank of a torch.distributed.run launch started by hand.\n")
            return 2

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # The port was free a moment ago; if something took it before the launcher bound it (its rendezvous then fails with EADDRINUSE
    # before any rank has started), pick another one.  Nothing else is ever retried.
    import threading

    limit_s = float(os.environ.get("C2D_BENCH_LAUNCH_TIMEOUT_S", "3000"))
    returncode, out_text = 1, ""
    for _attempt in range(3):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        # (its own session: on a timeout the launcher AND its ranks are signalled as one process group — SIGKILL to the launcher
        # alone cannot be forwarded and would leave the ranks holding their GPUs)
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
        # The ranks' diagnostics are relayed AS THEY COME (a run that hangs in the rendezvous or in RCCL's start-up must not be
        # silent), and scanned: the port retry below applies only while no rank has said anything of its own yet.
        seen = {"addr_in_use": False, "rank_lines": 0, "stdout": []}

        def relay(pipe=child.stderr, seen=seen):
            for ln in pipe:
                sys.stderr.write(ln)
                sys.stderr.flush()
                if "[bench]" in ln:
                    seen["rank_lines"] += 1
                elif "EADDRINUSE" in ln and seen["rank_lines"] == 0:
                    seen["addr_in_use"] = True

        def collect(pipe=child.stdout, seen=seen):
            seen["stdout"].append(pipe.read())

        threads = [threading.Thread(target=relay, daemon=True), threading.Thread(target=collect, daemon=True)]
        for t in threads:
            t.start()
        import signal

        def forward(signum, _frame, child=child):  # a signal to this parent (a `timeout` wrapper, Ctrl-C) reaches the ranks' own session too
            try:
                os.killpg(child.pid, signum)
            except (ProcessLookupError, PermissionError):
                pass

        previous = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT)}
        timed_out = False
        try:
            returncode = child.wait(timeout=limit_s)
        except subprocess.TimeoutExpired:
            timed_out = True
            for sig in (signal.SIGTERM, signal.SIGKILL):  # SIGKILL goes to the group even when the launcher left on SIGTERM: a rank may not have
                try:
                    os.killpg(child.pid, sig)  # (the session leader's pid is the group's id)
                except (ProcessLookupError, PermissionError):
                    break
                try:
                    child.wait(timeout=10.0)
                except subprocess.TimeoutExpired:
                    pass
            returncode = child.wait() or 1
            sys.stderr.write(f"[bench] the {n_gpus} ranks did not finish within {limit_s:.0f} s ($C2D_BENCH_LAUNCH_TIMEOUT_S): "
                             "the launcher's process group was terminated\n")
        for sg, h in previous.items():
            signal.signal(sg, h)
        for t in threads:
            t.join(timeout=10)
        out_text = "".join(seen["stdout"])
        if returncode == 0 or timed_out or not seen["addr_in_use"] or seen["rank_lines"] or '"metric"' in out_text:
            break  # (never a second set of ranks after a timeout)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    line = None
    for ln in out_text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    if returncode != 0:
        return returncode
    return 0 if line is not None else 1


def torch_random_convex_polygons(torch, dev, n, seed, kmin=3, kmax=KMAX, extent=8.0, rows=KMAX):
    """Device-side twin of workloads.random_convex_polygons (config 5 input); rows = vertex rows per polygon of the layout."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    vx = torch.zeros((2, rows, n), dtype=torch.float32, device=dev)
    vy = torch.zeros_like(vx)
    k = torch.randint(kmin, kmax + 1, (2, n), generator=g, device=dev, dtype=torch.int32)
    for p in range(2):
        mask = torch.arange(rows, device=dev)[:, None] >= k[p][None, :]
        ang = torch.rand((rows, n), generator=g, device=dev) * (2 * np.pi)
        ang[mask] = float("inf")
        ang, _ = torch.sort(ang, dim=0)
        ang[mask] = 0
        a = torch.rand(n, generator=g, device=dev) * 2.2 + 0.3
        b = torch.rand(n, generator=g, device=dev) * 2.2 + 0.3
        rot = torch.rand(n, generator=g, device=dev) * (2 * np.pi)
        cx = (torch.rand(n, generator=g, device=dev) * 2 - 1) * extent
        cy = (torch.rand(n, generator=g, device=dev) * 2 - 1) * extent
        x, y = a[None] * torch.cos(ang), b[None] * torch.sin(ang)
        c, s = torch.cos(rot)[None], torch.sin(rot)[None]
        vx[p] = c * x - s * y + cx[None]
        vy[p] = s * x + c * y + cy[None]
        vx[p][mask] = 0
        vy[p][mask] = 0
    return vx, vy, k.to(torch.uint8)


class Run:
    """What the legs of one bench run share: setup() puts the rank, the device, the engine, the stream and the reduce on it, every leg reads what it
    needs from it at its top and leaves what later legs and the result line read at its bottom (the data flow between the legs is those lines)."""


def setup(args, R) -> None:
    """rank, device, engine, stream, the reduce (torch.distributed + libc2d's communicator) and the helpers every leg uses, onto R"""
    # Exactly one line may reach stdout (the JSON result).  Libraries write there too — RCCL prints a
    # version banner to stdout when its communicator is created — so fd 1 is pointed at stderr for the
    # whole run and the result is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from __graft_entry__ import load_package

    pkg = load_package()
    import importlib

    wl = importlib.import_module("c2d_amd.workloads")
    shd = importlib.import_module("c2d_amd.sharding")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    # --share-device is the rehearsal of the N > 1 code path on a box with fewer GPUs than ranks: RCCL cannot put two ranks
    # on one device, so the ranks load the rehearsal build of the library (a sum through files; tests only, `make lib-rehearsal`)
    lib_path = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "lib-rehearsal", "libc2d.so") if args.share_device else None
    eng = pkg.Engine(local_rank, lib_path=lib_path)  # raises if libc2d.so is missing or the device is not gfx950
    dev_info = eng.info()
    rccl_version, rccl_library = None, None
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    counts = measured_counts()

    # ---- the reduce: libc2d's communicator (RCCL), created now — RCCL takes seconds for that, and an idle GPU in
    # front of the timed region would run its first steps through the post-idle clock ramp
    cdist, reduce_impl = None, "none (single rank)"
    if use_dist:
        reduce_impl = "torch.distributed all_reduce (%s)" % args.backend
        if args.reduce == "c2d":
            # rank 0 creates the id; it is broadcast whether or not that worked (an all-zero id = "could not"), so that a
            # failure on one rank never leaves the others waiting in a collective
            raw = bytes(pkg.binding.DIST_ID_BYTES)
            if rank == 0:
                try:
                    raw = eng.dist_unique_id()
                except Exception as e:  # noqa: BLE001
                    print(f"[bench] rank 0: c2d_dist_unique_id failed ({e})", file=sys.stderr)
            if args.backend == "nccl":
                id_t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
                dist.broadcast(id_t, 0)
                uid = id_t.cpu().numpy().tobytes()
            else:
                box = [raw]
                dist.broadcast_object_list(box, 0)
                uid = box[0]
            ok = 0
            if any(uid):
                try:
                    cdist = eng.dist_init(rank, world, uid)  # under libc2d's watchdog: C2D_ERR_DIST if a peer never arrives
                    ok = 1 if cdist.world_size == world else 0
                    if not ok:
                        print(f"[bench] rank {rank}: RCCL counts {cdist.world_size} ranks in the communicator, the launcher {world}", file=sys.stderr)
                except Exception as e:  # noqa: BLE001
                    print(f"[bench] rank {rank}: c2d_dist_init failed ({e})", file=sys.stderr)
            # every rank learns whether every rank succeeded, so that all of them leave together
            flag = torch.tensor([ok], dtype=torch.int32, device=dev if args.backend == "nccl" else None)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                print(f"[bench] rank {rank}: the product reduce (c2d_dist over RCCL) is not available on every rank; not measuring with "
                      "another one (--reduce torch asks for torch.distributed explicitly)", file=sys.stderr)
                dist.destroy_process_group()
                os._exit(3)  # (a helper thread of the watchdog may still sit inside RCCL)
            reduce_impl = "libc2d c2d_dist_all_reduce_sum_u64 (%s, %d ranks in the communicator)" % (cdist.transport, cdist.world_size)
            rccl_version, rccl_library = eng.dist_rccl_version()
        elif args.backend == "nccl":
            rccl_version, rccl_library = int("%d%02d%02d" % tuple(torch.cuda.nccl.version()[:3])), "torch.distributed (bundled librccl)"
    # one line per rank on stderr: which card, which RCCL — so that a scaling record can be read without guessing
    print("[bench] rank %d of %d: device %d (%s, PCI %s), host pid %d, reduce: %s, RCCL %s from %s" % (
        rank, world, local_rank, dev_info["name"], dev_info["pci_bus_id"] or "?", os.getpid(), reduce_impl, rccl_version, rccl_library), file=sys.stderr, flush=True)

    def all_reduce_sum(t):
        """The one collective of each leg: sum of 64-bit counters over ranks, in place (t: int64 device tensor)."""
        if not use_dist:
            return
        if cdist is not None:
            cdist.all_reduce_sum_u64(t.data_ptr(), t.numel(), stream=sh)
        elif args.backend == "nccl":
            with torch.cuda.stream(stream):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:  # rehearsal backend: reduce a host copy
            torch.cuda.synchronize()
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)

    if use_dist:
        warm = torch.zeros(1, dtype=torch.int64, device=dev)
        all_reduce_sum(warm)
        torch.cuda.synchronize()  # (two communicators live in this process — c2d's and torch's: never a collective of each in flight at once)
        dist.barrier()
        torch.cuda.synchronize()
    # ---- what makes an N > 1 line readable: every rank's own kernel time, and the closing reduce by itself ----
    on_device = args.backend == "nccl"

    def rank_spread(kernel_ms, units, roofline=None, work=None):
        """Every rank's own kernel time (c2d_amd.sharding.kernel_time_spread: one all_gather AFTER the timed region): returns
        ({min, median, max, ...} of kernel_ms over ranks, all ranks' units / the slowest rank's time) and writes `kernel_ms_ranks` and
        `frac_slowest_rank` into the roofline — `frac` is rank 0's, a curve is bounded by the slowest rank's."""
        spread, kernels_only, frac_slowest = shd.kernel_time_spread(kernel_ms, units, roofline["frac"] if roofline is not None else None, work,
                                                                    dev if on_device else None)
        if roofline is not None:
            roofline["kernel_ms_ranks"] = spread
            roofline["frac_slowest_rank"] = frac_slowest
        return spread, kernels_only

    def time_reduce(numel=1, before=None, reps=20):
        """The closing all_reduce_sum by itself, microseconds, in a SEPARATE untimed pass: its own event pair on the kernels' stream
        (opened behind one launch of `before`, the leg's step, so that the stream is as busy as where the timed region meets the
        reduce) and the host's wall time of call + synchronize.  None with a single rank and no process group."""
        if not use_dist:
            return None
        scratch = torch.zeros(numel, dtype=torch.int64, device=dev)
        dist.barrier()
        torch.cuda.synchronize()
        ev_us, host_us = [], []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if before is not None:
                before()
            h0 = time.perf_counter()
            e0.record(stream)
            all_reduce_sum(scratch)
            e1.record(stream)
            torch.cuda.synchronize()
            host_us.append((time.perf_counter() - h0) * 1e6)
            ev_us.append(e0.elapsed_time(e1) * 1e3)
        ev, ho = np.array(ev_us), np.array(host_us)
        return {"reps": reps, "counters": numel, "event_median": round(float(np.median(ev)), 1), "event_min": round(float(ev.min()), 1),
                "event_max": round(float(ev.max()), 1), "host_median": round(float(np.median(ho)), 1),
                "note": "separate untimed pass; event pair on the kernels' stream around the reduce alone, each behind one launch of the leg's "
                        "step (so it includes waiting for the slowest peer's step); host = wall time of call + synchronize, which also "
                        "waits for that launch to drain"}

    barrier_us = None
    if use_dist:  # what one dist.barrier() + synchronize costs the host here: each timed region contains one
        b = []
        for _ in range(10):
            h0 = time.perf_counter()
            dist.barrier()
            torch.cuda.synchronize()
            b.append((time.perf_counter() - h0) * 1e6)
        barrier_us = round(float(np.median(b)), 1)
    R.barrier_us, R.rank_spread, R.time_reduce = barrier_us, rank_spread, time_reduce
    # what the later legs and the result line read
    R.all_reduce_sum, R.args, R.cdist, R.counts, R.dev, R.dev_info, R.dist, R.eng, R.pkg, R.rank = all_reduce_sum, args, cdist, counts, dev, dev_info, dist, eng, pkg, rank
    R.rccl_library, R.rccl_version, R.real_stdout, R.reduce_impl, R.sh, R.shd, R.stream, R.torch, R.use_dist, R.wl = rccl_library, rccl_version, real_stdout, reduce_impl, sh, shd, stream, torch, use_dist, wl
    R.world, = world,


def leg_pairs(R) -> None:
    """workload: config 2, resident in HBM"""
    # state the legs before this one left on R
    all_reduce_sum, args, counts, dev, dist, eng, rank, sh, shd, stream = R.all_reduce_sum, R.args, R.counts, R.dev, R.dist, R.eng, R.rank, R.sh, R.shd, R.stream
    torch, use_dist, world = R.torch, R.use_dist, R.world
    # ---- workload: config 2, resident in HBM -----------------------------------------
    n = args.pairs
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5A7 + rank)
    pose = torch.empty((10, n), dtype=torch.float32, device=dev)
    for r in range(2):
        pose[5 * r + 0].uniform_(-8.0, 8.0, generator=gen)
        pose[5 * r + 1].uniform_(-8.0, 8.0, generator=gen)
        pose[5 * r + 2].uniform_(0.1, 5.0, generator=gen)
        pose[5 * r + 3].uniform_(0.1, 5.0, generator=gen)
        pose[5 * r + 4].uniform_(0.0, 2.0 * np.pi, generator=gen)
    planes = torch.empty((16, n), dtype=torch.float32, device=dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    row = lambda t, k: t.data_ptr() + k * t.stride(0) * t.element_size()  # noqa: E731
    for r in range(2):
        eng.rects_from_poses(*[row(pose, 5 * r + k) for k in range(5)], n, [row(planes, 8 * r + k) for k in range(8)], stream=sh)
    plane_ptrs = [row(planes, k) for k in range(16)]
    torch.cuda.synchronize()
    pose_ptrs = [row(pose, k) for k in range(10)]

    def step():
        eng.sat_rect_pairs_verts(plane_ptrs, n, out.data_ptr(), count.data_ptr(), stream=sh)

    def barrier():
        if use_dist:
            dist.barrier()

    def prewarm(fn):
        w0 = time.perf_counter()
        while (time.perf_counter() - w0) * 1e3 < args.prewarm_ms:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()

    def step_distribution(fn, k):
        """Per-step durations (ms) of k further steps, each bracketed by its own pair of events, in a SEPARATE untimed
        pass: event markers between back-to-back 100-us kernels cost a few us of gap each, so they stay out of the
        timed region, whose mean comes from one event pair around all K steps."""
        prewarm(fn)  # the pass follows host work (reduce, read-back): without it the steps run inside the ~15 ms clock ramp after idle
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        for i in range(k):
            evs[i].record(stream)
            fn()
        evs[k].record(stream)
        torch.cuda.synchronize()
        d = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(k)])
        return {"steps": k, "median": round(float(np.median(d)), 5), "min": round(float(d.min()), 5), "max": round(float(d.max()), 5),
                "note": "separate untimed pass with an event marker between steps"}

    prewarm(step)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    count.zero_()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):  # EXACTLY K steps
        step()
    ev1.record(stream)
    total_count = count
    all_reduce_sum(count)  # the single RCCL reduce of the hit counts
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = shd.max_over_ranks(t1 - t0, dev)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch duration on the kernel's stream
    count_after_timed = int(total_count.item())
    step_ms = step_distribution(step, min(args.steps, 100))
    pairs_total = n * world * args.steps
    value = pairs_total / elapsed
    reduce_us = R.time_reduce(1, before=step)
    collide_rate = count_after_timed / (n * world * args.steps)

    achieved_gbs = BYTES_PER_PAIR * n / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    c = counts.get("sat_rect_verts.config2")
    if c and n == c.get("pairs"):
        traffic = c.get("hbm_bytes_per_launch")
        traffic_source = "recorded, not measured in this run: %s (2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes)" % c.get("source")
    roofline = {"bound": "hbm", "kernel": "sat_rect_verts_kernel<4, 64>", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": BYTES_PER_PAIR * n, "kernel_ms": round(kernel_ms, 5),
                "step_ms_distribution": step_ms}
    spread, value_kernels_only = R.rank_spread(kernel_ms, n, roofline)
    # how the timed region divides: the slowest rank's K launches, and what is left (launch gaps, the reduce, one barrier, synchronize)
    scaling_detail = {"timed_region_ms": round(elapsed * 1e3, 4), "kernels_ms_slowest_rank": round(spread["max"] * args.steps, 4),
                      "timed_region_minus_kernels_us": round((elapsed * 1e3 - spread["max"] * args.steps) * 1e3, 1),
                      "barrier_us": R.barrier_us,
                      "note": "value = all ranks' pairs / timed_region (contract: barrier + synchronize on both sides, max over ranks); "
                              "value_kernels_only = all ranks' pairs per launch / the slowest rank's average launch time by HIP events: a fixed "
                              "closing cost of the 2-ms region (reduce_us, barrier_us) lowers `value` at any N and is not a scaling loss of the kernels"}
    R.reduce_us, R.scaling_detail, R.value_kernels_only = reduce_us, scaling_detail, value_kernels_only
    # what the later legs and the result line read
    R.barrier, R.collide_rate, R.count_after_timed, R.elapsed, R.n, R.out, R.plane_ptrs, R.planes, R.pose, R.pose_ptrs = barrier, collide_rate, count_after_timed, elapsed, n, out, plane_ptrs, planes, pose, pose_ptrs
    R.prewarm, R.roofline, R.step_distribution, R.value = prewarm, roofline, step_distribution, value


def leg_mask_output(R) -> None:
    """same kernel arithmetic, bit-mask output (64.125 B/pair)"""
    # state the legs before this one left on R
    args, counts, dev, eng, n, plane_ptrs, prewarm, sh, stream, torch = R.args, R.counts, R.dev, R.eng, R.n, R.plane_ptrs, R.prewarm, R.sh, R.stream, R.torch
    # ---- same kernel arithmetic, bit-mask output (64.125 B/pair) ---------------------------------------------------
    mask_leg = None
    if not args.no_pose:
        mwords = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
        mcount = torch.zeros(1, dtype=torch.int64, device=dev)

        def mask_step():
            eng.sat_rect_pairs_verts_mask(plane_ptrs, n, mwords.data_ptr(), mcount.data_ptr(), stream=sh)

        prewarm(mask_step)
        me0, me1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        me0.record(stream)
        for _ in range(args.steps):
            mask_step()
        me1.record(stream)
        torch.cuda.synchronize()
        mms = me0.elapsed_time(me1) / args.steps
        mgbs = (64 * n + (n + 7) // 8) / (mms * 1e-3) / 1e9
        mask_leg = {"metric": "sat_pair_tests_per_s (bit-mask output, per GPU)", "value": n / (mms * 1e-3), "kernel_ms": round(mms, 5),
                    "bytes_per_pair": 64.125,
                    "roofline": {"bound": "hbm", "kernel": "sat_rect_verts_mask4_kernel", "achieved": round(mgbs, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(mgbs / HBM_PEAK_GBS, 4), "traffic": None}}
        c = counts.get("sat_rect_verts_mask.config2")
        if c and n == c.get("pairs"):
            mask_leg["roofline"]["traffic"] = c.get("hbm_bytes_per_launch")
            mask_leg["roofline"]["traffic_source"] = "recorded, not measured in this run: %s" % c.get("source")
        del mwords
    # what the later legs and the result line read
    R.mask_leg, = mask_leg,


def leg_pose_format(R) -> None:
    """secondary input format: poses (41 B/pair), reported separately"""
    # state the legs before this one left on R
    args, counts, dev, eng, n, out, pose_ptrs, prewarm, sh, step_distribution = R.args, R.counts, R.dev, R.eng, R.n, R.out, R.pose_ptrs, R.prewarm, R.sh, R.step_distribution
    stream, torch = R.stream, R.torch
    # ---- secondary input format: poses (41 B/pair), reported separately ----------
    pose_leg = None
    if not args.no_pose:
        pcount = torch.zeros(1, dtype=torch.int64, device=dev)

        def pose_step():
            eng.sat_rect_pairs_pose(pose_ptrs, n, out.data_ptr(), pcount.data_ptr(), stream=sh)

        prewarm(pose_step)
        pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        pe0.record(stream)
        for _ in range(args.steps):
            pose_step()
        pe1.record(stream)
        torch.cuda.synchronize()
        pms = pe0.elapsed_time(pe1) / args.steps
        pose_gbs = 41 * n / (pms * 1e-3) / 1e9
        pose_leg = {"metric": "sat_pair_tests_per_s (pose format, per GPU)", "value": n / (pms * 1e-3), "kernel_ms": round(pms, 5),
                    "bytes_per_pair": 41,
                    "roofline": {"bound": "hbm", "kernel": "sat_rect_pose_kernel<4, 64>", "achieved": round(pose_gbs, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(pose_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                                 "step_ms_distribution": step_distribution(pose_step, min(args.steps, 100))},
                    "note": "a pair is decided from (cx,cy,w,h,theta) by the closed-form gap of the two rectangles when it exceeds a proven "
                            "rounding margin (2 sincos + ~70 instructions), by the reference's vertex arithmetic otherwise (about one pair in "
                            "1e4 here): HBM-bound since round 3"}
        c = counts.get("sat_rect_pose.config2")
        if c and n == c.get("pairs"):
            lane = n / (pms * 1e-3) * c["valu_instr_per_pair"] / 1e12
            pose_leg["valu_roofline"] = {"bound": "valu", "achieved": round(lane, 2), "peak": VALU_PEAK_TLANE, "unit": "T VALU lane-instr/s",
                                         "frac": round(lane / VALU_PEAK_TLANE, 4), "valu_instr_per_pair": c["valu_instr_per_pair"],
                                         "instr_source": c.get("source"),
                                         "note": "peak priced at the nominal 2.4 GHz; secondary to the HBM roofline above since the closed-form "
                                                 "pair test (round 3) took the kernel from 294 to the recorded instruction count per pair"}
            pose_leg["roofline"]["traffic"] = c.get("hbm_bytes_per_launch")
            pose_leg["roofline"]["traffic_source"] = "recorded, not measured in this run: %s" % c.get("source")
    # what the later legs and the result line read
    R.pose_leg, = pose_leg,


def leg_host_resident(R) -> None:
    """the same batch starting in HOST memory (include/c2d.h c2d_sat_rect_pairs_*_host): never `value` — the link is the bound"""
    # state the legs before this one left on R
    args, eng, n, out, planes, pose, rank, world = R.args, R.eng, R.n, R.out, R.planes, R.pose, R.rank, R.world
    # ---- the same batch starting in HOST memory (include/c2d.h c2d_sat_rect_pairs_*_host): never `value` — the link is the bound ---
    host_leg = None
    if not args.no_pose and rank == 0 and world == 1:
        import ctypes as C

        pin = eng.host_empty((16, n), np.float32)
        pin[:] = planes.cpu().numpy()
        pin_pose = eng.host_empty((10, n), np.float32)
        pin_pose[:] = pose.cpu().numpy()
        h_res = eng.host_empty(n, np.uint8)
        d_tmp = eng.empty((16, n), np.float32)

        def link_copy():
            eng._check(eng.lib.c2d_memcpy_h2d(eng.h, C.c_void_p(d_tmp.ptr), C.c_void_p(pin.ctypes.data), pin.nbytes, None), "c2d_memcpy_h2d")
            eng.synchronize()

        def best_of(fn, k=4):
            b = 1e9
            for _ in range(k):
                t_ = time.perf_counter()
                fn()
                b = min(b, time.perf_counter() - t_)
            return b

        link_gbs = pin.nbytes / best_of(link_copy) / 1e9
        d_tmp.free()
        gpu_bools = out.cpu().numpy()
        host_leg = {"note": "the batch starts and ends in host memory (page-locked): upload, test, download in one synchronous call; bound by the "
                            "host-to-device link, reported as a fraction of one large page-locked copy measured in this run; never the headline value",
                    "link_h2d_GBs": round(link_gbs, 1)}
        for fmt, arr, bpp in (("verts", pin, 64), ("pose", pin_pose, 40)):
            cnt_box = {}

            def call():
                cnt_box["c"] = eng.sat_rect_pairs_host([arr[k] for k in range(arr.shape[0])], h_res, fmt)

            t_host = best_of(call)
            same = int((h_res == gpu_bools).sum())
            if same != n:
                raise SystemExit(f"PARITY FAILURE: host-resident {fmt} entry point differs from the device entry point on {n - same} of {n} pairs")
            host_leg[fmt] = {"pairs_per_s": n / t_host, "ms": round(t_host * 1e3, 3), "bytes_up_per_pair": bpp,
                             "GBs_up": round(bpp * n / t_host / 1e9, 1), "frac_of_link": round(bpp * n / t_host / 1e9 / link_gbs, 3),
                             "parity": f"booleans equal to the device entry point's on {same} of {n} pairs", "colliding": cnt_box["c"]}
        for a_ in (pin, pin_pose, h_res):
            eng.host_free(a_)
    del pose
    # what the later legs and the result line read
    R.host_leg, R.pose = host_leg, None


def leg_mc(R) -> None:
    """Monte-Carlo leg: config 3"""
    # state the legs before this one left on R
    all_reduce_sum, args, barrier, counts, dev, eng, prewarm, rank, sh, shd = R.all_reduce_sum, R.args, R.barrier, R.counts, R.dev, R.eng, R.prewarm, R.rank, R.sh, R.shd
    stream, torch, wl, world = R.stream, R.torch, R.wl, R.world
    S = mc_hits_one_step = sc = None
    # ---- Monte-Carlo leg: config 3 --------------------------------------------------------
    mc = None
    if not args.no_mc:
        sc = wl.MC_PAIR_SCENE
        hits = torch.zeros(1, dtype=torch.int64, device=dev)
        S = args.mc_samples

        def mc_step():
            # rank r evaluates samples [r*S, (r+1)*S) of stream (seed 1234, scene 0)
            eng.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, rank * S, S,
                        hits.data_ptr(), stream=sh)

        prewarm(mc_step)
        mc_step()
        torch.cuda.synchronize()
        hits.zero_()
        barrier()
        torch.cuda.synchronize()
        m0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.mc_reps):
            mc_step()
        e1.record(stream)
        all_reduce_sum(hits)
        torch.cuda.synchronize()
        barrier()
        m1 = time.perf_counter()
        mel = shd.max_over_ranks(m1 - m0, dev)
        mc_kernel_ms = e0.elapsed_time(e1) / args.mc_reps
        p = float(hits.item()) / (S * world * args.mc_reps)
        # hits of ONE step on this rank's range (every rep repeats the same samples), for the oracle comparison below
        mc_hits_one_step = int(hits.item()) // args.mc_reps if world == 1 else None
        if world == 1 and int(hits.item()) != mc_hits_one_step * args.mc_reps:
            raise SystemExit("bench: the Monte-Carlo reps of one sample range gave different hit counts")
        mc = {"metric": "mc_samples_per_s", "value": S * world * args.mc_reps / mel, "samples_per_gpu": S, "reps": args.mc_reps,
              "kernel_ms": round(mc_kernel_ms, 4), "probability": p, "scene": "config3: robot 4.07x1.74 at (3,1) th=0.6, obstacle 2x1, sigma=(.3,.3,.2,0,0)",
              "bound": "valu", "note": "~0 HBM bytes per sample; VALU/transcendental bound (DESIGN.md)"}
        mc["reduce_us"] = R.time_reduce(1, before=mc_step, reps=5)
        c = counts.get("mc_pair.config3")
        if c:  # the count belongs to THIS scene (wl.MC_PAIR_SCENE) and kernel build
            lane_ops = S / (mc_kernel_ms * 1e-3) * c["valu_instr_per_sample"] / 1e12
            mc["roofline"] = {"bound": "valu", "achieved": round(lane_ops, 2), "peak": VALU_PEAK_TLANE,
                              "unit": "T VALU lane-instr/s per GPU (peak = 157.3 TFLOP/s / 2 flop per FMA)",
                              "frac": round(lane_ops / VALU_PEAK_TLANE, 4), "valu_instr_per_sample": c["valu_instr_per_sample"],
                              "instr_source": c.get("source")}
            held_clock(mc["roofline"], c)
            issue_weighted(mc["roofline"], c, S / (mc_kernel_ms * 1e-3) * c["valu_instr_per_sample"] / 64)
        mc["kernel_ms_ranks"], mc["value_kernels_only"] = R.rank_spread(mc_kernel_ms, S, mc.get("roofline"))
    # what the later legs and the result line read
    R.S, R.mc, R.mc_hits_one_step, R.sc = S, mc, mc_hits_one_step, sc


def leg_mc_poly(R) -> None:
    """Monte-Carlo over convex polygons (README.md:3 "arbitrary convex 2D shapes"; include/c2d.h c2d_mc_poly_pair)"""
    # state the legs before this one left on R
    all_reduce_sum, args, barrier, counts, dev, eng, pkg, prewarm, rank, sh = R.all_reduce_sum, R.args, R.barrier, R.counts, R.dev, R.eng, R.pkg, R.prewarm, R.rank, R.sh
    shd, stream, torch, wl, world = R.shd, R.stream, R.torch, R.wl, R.world
    PS = mc_poly_hits_one_step = mc_poly_scenes_keep = psc = None
    # ---- Monte-Carlo over convex polygons (README.md:3 "arbitrary convex 2D shapes"; include/c2d.h c2d_mc_poly_pair) -------------
    mc_poly = None
    if not args.no_mc:
        psc = wl.mc_poly_pair_scene()  # 7-gon robot, pentagon obstacle, config 3's pose noise, p ~ 0.57
        phits = torch.zeros(1, dtype=torch.int64, device=dev)
        PS = args.mc_samples
        p_robot, p_obst = pkg.make_polygon(*psc["robot"]), pkg.make_polygon(*psc["obstacle"])

        def mc_poly_step():
            eng.mc_poly_pair(p_robot, psc["pos"], psc["theta"], p_obst, psc["std_dev"], 1234, 0, rank * PS, PS, phits.data_ptr(), stream=sh)

        prewarm(mc_poly_step)
        torch.cuda.synchronize()
        phits.zero_()
        barrier()
        torch.cuda.synchronize()
        pm0 = time.perf_counter()
        pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pe0.record(stream)
        for _ in range(args.mc_reps):
            mc_poly_step()
        pe1.record(stream)
        all_reduce_sum(phits)
        torch.cuda.synchronize()
        barrier()
        pmel = shd.max_over_ranks(time.perf_counter() - pm0, dev)
        mc_poly_hits_one_step = int(phits.item()) // args.mc_reps if world == 1 else None  # (every rep repeats the same samples)
        mc_poly_ms = pe0.elapsed_time(pe1) / args.mc_reps
        mc_poly = {"metric": "mc_poly_samples_per_s", "value": PS * world * args.mc_reps / pmel, "samples_per_gpu": PS, "reps": args.mc_reps,
                   "kernel_ms": round(mc_poly_ms, 4), "probability": float(phits.item()) / (PS * world * args.mc_reps),
                   "scene": "7-gon robot about 4.7 x 2.0 at (2.8, 1.0) th=0.6, pentagon obstacle about 2.3 x 1.2, sigma=(.3,.3,.2,0,0)",
                   "bound": "valu", "note": "~0 HBM bytes per sample; one sample per lane, the interval test on all ka + kb true normals (DESIGN.md §5)"}
        c = counts.get("mc_poly_pair.bench")
        if c:
            lane_ops = PS / (mc_poly_ms * 1e-3) * c["valu_instr_per_sample"] / 1e12
            mc_poly["roofline"] = {"bound": "valu", "kernel": "mc_poly_pair_kernel", "achieved": round(lane_ops, 2), "peak": VALU_PEAK_TLANE,
                                   "unit": "T VALU lane-instr/s per GPU", "frac": round(lane_ops / VALU_PEAK_TLANE, 4),
                                   "valu_instr_per_sample": c["valu_instr_per_sample"], "instr_source": c.get("source")}
            held_clock(mc_poly["roofline"], c)
            issue_weighted(mc_poly["roofline"], c, PS / (mc_poly_ms * 1e-3) * c["valu_instr_per_sample"] / 64)
        mc_poly["kernel_ms_ranks"], mc_poly["value_kernels_only"] = R.rank_spread(mc_poly_ms, PS, mc_poly.get("roofline"))
        mc_poly["reduce_us"] = R.time_reduce(1, before=mc_poly_step, reps=5)
        # the adaptive loop over a dataset of polygon scenes (c2d_mc_poly_scenes): random obstacle polygons of 3..16 vertices, a 9-gon
        # robot, the stop rule of config 4.  Scenes shard over ranks like config 4's (scene_id_base = the shard's first scene).
        PN = args.poly_scenes
        if PN > 0:
            pp_tab, ps_tab = wl.random_poly_tables(4096, 4096, seed=7)
            p_scn = wl.random_poly_scenes(PN * world, pp_tab, ps_tab, 2.3, seed=8)[rank * PN:(rank + 1) * PN]
            p_rob9 = wl.mc_poly_pair_scene(9, 5)["robot"]
            d_pp, d_ps, d_pscn = eng.to_device(pp_tab), eng.to_device(ps_tab), eng.to_device(p_scn)
            d_ph, d_pu = eng.zeros(PN, np.uint32), eng.zeros(PN, np.uint32)

            def poly_scenes_step():
                return eng.mc_poly_scenes(p_rob9, d_pp, len(pp_tab), d_ps, len(ps_tab), d_pscn, PN, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 120_000, 11,
                                          rank * PN, d_ph, d_pu, None, stream=sh)

            poly_scenes_step()  # warm (clocks, allocations of the schedule's lists)
            torch.cuda.synchronize()
            barrier()
            pq0 = time.perf_counter()
            p_total, p_iters = poly_scenes_step()
            torch.cuda.synchronize()
            pq_own_ms = (time.perf_counter() - pq0) * 1e3  # this rank's own call (synchronous: it returns the totals), before the barrier
            barrier()
            pqel = shd.max_over_ranks(time.perf_counter() - pq0, dev)
            p_tot_t = torch.tensor([p_total], dtype=torch.int64, device=dev)
            all_reduce_sum(p_tot_t)
            torch.cuda.synchronize()
            mc_poly["scenes"] = {"metric": "mc_poly_scenes_per_s", "value": PN * world / pqel, "scenes_per_gpu": PN, "max_samples": 120_000, "seconds": round(pqel, 4),
                                 "drawn_samples_per_s": int(p_tot_t.item()) / pqel, "mean_samples_per_scene": int(p_tot_t.item()) / (PN * world), "steps": p_iters,
                                 "workload": "4096 random obstacle polygons (3..16 vertices) x 4096 standard deviations, 9-gon robot, adaptive stopping as config 4"}
            cq = counts.get("mc_poly_scenes.bench")
            if cq and cq.get("scenes") == PN and cq.get("max_samples") == 120_000:
                lane_ops = p_total / pqel * cq["valu_instr_per_sample"] / 1e12   # (this rank's drawn samples over the slowest rank's time)
                mc_poly["scenes"]["roofline"] = {"bound": "valu", "kernel": "mc_poly_scenes_advance_kernel (all schedule steps)", "achieved": round(lane_ops, 2),
                                                 "peak": VALU_PEAK_TLANE, "unit": "T VALU lane-instr/s per GPU", "frac": round(lane_ops / VALU_PEAK_TLANE, 4),
                                                 "valu_instr_per_sample": cq["valu_instr_per_sample"], "instr_source": cq.get("source"),
                                                 "note": "instructions per DRAWN sample, as for config 4"}
                held_clock(mc_poly["scenes"]["roofline"], cq)
                issue_weighted(mc_poly["scenes"]["roofline"], cq, p_total / pqel * cq["valu_instr_per_sample"] / 64)
            mc_poly["scenes"]["call_ms_ranks"], mc_poly["scenes"]["value_kernels_only"] = R.rank_spread(pq_own_ms, PN, mc_poly["scenes"].get("roofline"), work=p_total)
            mc_poly["scenes"]["call_ms_note"] = "host wall time of each rank's own synchronous c2d_mc_poly_scenes call (the call returns the totals, so there is no event pair)"
            mc_poly_scenes_keep = (pp_tab, ps_tab, p_scn, p_rob9, d_ph.get(), d_pu.get(), rank * PN)
            for a_ in (d_pp, d_ps, d_pscn, d_ph, d_pu):
                a_.free()
    # what the later legs and the result line read
    R.PS, R.mc_poly, R.mc_poly_hits_one_step, R.mc_poly_scenes_keep, R.psc = PS, mc_poly, mc_poly_hits_one_step, mc_poly_scenes_keep, psc


def leg_scenes(R) -> None:
    """config 4: adaptive Monte-Carlo over many scenes"""
    # state the legs before this one left on R
    all_reduce_sum, args, barrier, counts, dev, eng, pkg, rank, sh, shd = R.all_reduce_sum, R.args, R.barrier, R.counts, R.dev, R.eng, R.pkg, R.rank, R.sh, R.shd
    stream, torch, wl, world = R.stream, R.torch, R.wl, R.world
    # ---- config 4: adaptive Monte-Carlo over many scenes -------------------------------------
    scenes_leg, scenes_keep = None, None
    if args.scenes > 0:
        ns = args.scenes
        tp, ts, _ = wl.random_tables(65536, 65536, seed=7)
        d_p, d_s = eng.to_device(tp), eng.to_device(ts)
        d_sc = eng.empty(ns, pkg.SCENE_DT)
        base = rank * ns  # scene ids (hence random streams) are global: rank r owns [r*ns, (r+1)*ns)
        eng.sample_scenes(d_p, 65536, d_s, 65536, 4.07, 1.74, 4.0, 7, base, ns, d_sc, stream=sh)
        # per-scene outputs as torch tensors (int32 holds the u32 counts: at most max_samples + one batch), so that the leg's
        # totals are summed on the device, on the kernels' stream, without a host pass inside the timed region
        t_h, t_u = torch.zeros(ns, dtype=torch.int32, device=dev), torch.zeros(ns, dtype=torch.int32, device=dev)
        hsum = torch.zeros(2, dtype=torch.int64, device=dev)
        # untimed warm-up: the same call on the first 100 000 data points (kernel code resident, clocks up), then the outputs cleared
        warm_n = min(ns, 100_000)
        eng.mc_scenes_async(d_p, 65536, d_s, 65536, d_sc, warm_n, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY,
                            args.scenes_max_samples, 11, base, t_h.data_ptr(), t_u.data_ptr(), None, stream=sh)
        with torch.cuda.stream(stream):
            t_h.zero_()
            t_u.zero_()
        torch.cuda.synchronize()
        barrier()
        s0 = time.perf_counter()
        se0, se1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        se0.record(stream)
        eng.mc_scenes_async(d_p, 65536, d_s, 65536, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY,
                            args.scenes_max_samples, 11, base, t_h.data_ptr(), t_u.data_ptr(), None, stream=sh)
        se1.record(stream)
        with torch.cuda.stream(stream):
            hsum[0] = t_h.sum(dtype=torch.int64)
            hsum[1] = t_u.sum(dtype=torch.int64)
            local_total = hsum[1].clone()
        all_reduce_sum(hsum)  # hit and sample totals: the one collective
        torch.cuda.synchronize()
        barrier()
        sel = shd.max_over_ranks(time.perf_counter() - s0, dev)
        loop_ms = se0.elapsed_time(se1)
        local_total = int(local_total.item())
        steps_run = int(torch.unique(t_u).numel())  # distinct stop points seen (a lower bound of the schedule steps with work)
        scenes_leg = {"metric": "mc_data_points_per_s", "value": ns * world / sel, "data_points_per_s": ns * world / sel,
                      "data_points_per_gpu": ns, "max_samples": args.scenes_max_samples, "distinct_stop_points": steps_run,
                      "seconds": round(sel, 4), "device_loop_ms": round(loop_ms, 3),
                      "drawn_samples_per_s": float(hsum[1].item()) / sel, "mean_samples_per_point": float(hsum[1].item()) / (ns * world),
                      "pooled_hit_fraction": float(hsum[0].item()) / float(hsum[1].item()),
                      "workload": "config4: scenes drawn by the generate_dataset formula from 65536-entry tables, adaptive stopping",
                      "note": "a data point = one (scene, obstacle instance) row of the dataset, sampled until its stop rule passes; "
                              "drawn samples include those the kernels dismiss from their radius word alone (DESIGN.md §5)"}
        c = counts.get("mc_scenes.config4")
        if c and ns == c.get("data_points") and args.scenes_max_samples == c.get("max_samples"):
            lane = local_total / (loop_ms * 1e-3) * c["valu_instr_per_sample"] / 1e12
            scenes_leg["roofline"] = {"bound": "valu", "kernel": "mc_scenes_advance_kernel (all schedule steps)", "achieved": round(lane, 2),
                                      "peak": VALU_PEAK_TLANE, "unit": "T VALU lane-instr/s per GPU", "frac": round(lane / VALU_PEAK_TLANE, 4),
                                      "valu_instr_per_sample": c["valu_instr_per_sample"], "instr_source": c.get("source"),
                                      "note": "instructions per DRAWN sample: most samples of this workload are certain misses decided from "
                                              "their radius word, four words per Philox block (DESIGN.md §5); ~0 HBM bytes per sample"}
            held_clock(scenes_leg["roofline"], c)
            issue_weighted(scenes_leg["roofline"], c, local_total / (loop_ms * 1e-3) * c["valu_instr_per_sample"] / 64)
            if c.get("evaluated_fraction"):
                scenes_leg["evaluated_samples_per_s"] = float(hsum[1].item()) / sel * c["evaluated_fraction"]
                scenes_leg["evaluated_fraction"] = c["evaluated_fraction"]
                scenes_leg["evaluated_note"] = ("samples that reach the full evaluation (second Box-Muller pair, rotation, closed-form test; vertices and the "
                                               "SAT's own arithmetic for a thin result); recorded: %s" % c.get("evaluated_source"))
        scenes_leg["device_loop_ms_ranks"], scenes_leg["value_kernels_only"] = R.rank_spread(loop_ms, ns, scenes_leg.get("roofline"), work=local_total)
        scenes_leg["reduce_us"] = R.time_reduce(2, reps=10)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            scenes_keep = {"tables": (tp, ts), "scenes": d_sc.get(), "hits": t_h.cpu().numpy().view(np.uint32), "used": t_u.cpu().numpy().view(np.uint32),
                           "base": base}
        # fixed-samples mode: max_samples = 1000, i.e. exactly one 1000-sample step per data point, no adaptivity —
        # the per-data-point floor of the loop (scene set-up, one wave per data point, one decide step)
        fixed_n = 1000
        with torch.cuda.stream(stream):
            t_h.zero_()
            t_u.zero_()
        torch.cuda.synchronize()
        barrier()
        f0 = time.perf_counter()
        fe0, fe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fe0.record(stream)
        eng.mc_scenes_async(d_p, 65536, d_s, 65536, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY,
                            fixed_n, 11, base, t_h.data_ptr(), t_u.data_ptr(), None, stream=sh)
        fe1.record(stream)
        with torch.cuda.stream(stream):
            fsum = torch.stack([t_h.sum(dtype=torch.int64), t_u.sum(dtype=torch.int64)])
        all_reduce_sum(fsum)
        torch.cuda.synchronize()
        barrier()
        fel = shd.max_over_ranks(time.perf_counter() - f0, dev)
        if int(t_u.min().item()) != fixed_n or int(t_u.max().item()) != fixed_n:
            raise SystemExit("bench: fixed-samples mode drew other than %d samples for some data point" % fixed_n)
        scenes_leg["fixed_samples"] = {"samples_per_point": fixed_n, "seconds": round(fel, 5), "data_points_per_s": ns * world / fel,
                                       "samples_per_s": float(fsum[1].item()) / fel, "pooled_hit_fraction": float(fsum[0].item()) / float(fsum[1].item()),
                                       "note": "every data point sampled exactly 1000 times (max_samples = 1000: one schedule step, no stop rule at work)"}
        fixed_ms = fe0.elapsed_time(fe1)
        scenes_leg["fixed_samples"]["device_loop_ms"] = round(fixed_ms, 4)
        scenes_leg["fixed_samples"]["device_loop_ms_ranks"], scenes_leg["fixed_samples"]["value_kernels_only"] = R.rank_spread(fixed_ms, ns)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            scenes_fixed_hits = t_h[:2000].cpu().numpy().view(np.uint32)
            scenes_keep["fixed_hits"] = scenes_fixed_hits
        for a_ in (d_p, d_s, d_sc):
            a_.free()
        del t_h, t_u
    # what the later legs and the result line read
    R.scenes_keep, R.scenes_leg = scenes_keep, scenes_leg


def leg_poly(R) -> None:
    """config 5: convex polygons K <= 16"""
    # state the legs before this one left on R
    all_reduce_sum, args, barrier, counts, dev, eng, prewarm, rank, sh, shd = R.all_reduce_sum, R.args, R.barrier, R.counts, R.dev, R.eng, R.prewarm, R.rank, R.sh, R.shd
    step_distribution, stream, torch, world = R.step_distribution, R.stream, R.torch, R.world
    # ---- config 5: convex polygons K <= 16 ---------------------------------------------------------
    poly_leg, poly_keep = None, None
    if args.poly_pairs > 0:
        npoly = args.poly_pairs
        vx, vy, kk = torch_random_convex_polygons(torch, dev, npoly, seed=0xC0FFEE + rank)
        pout = torch.empty(npoly, dtype=torch.uint8, device=dev)
        pcnt = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()

        def poly_step():
            eng.sat_poly_pairs(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), npoly, pout.data_ptr(), pcnt.data_ptr(), stream=sh)

        prewarm(poly_step)
        torch.cuda.synchronize()
        pcnt.zero_()
        barrier()
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        preps = args.poly_reps
        q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        q0.record(stream)
        for _ in range(preps):
            poly_step()
        q1.record(stream)
        all_reduce_sum(pcnt)
        torch.cuda.synchronize()
        barrier()
        pel = shd.max_over_ranks(time.perf_counter() - p0, dev)
        eng.check_async()
        poly_ms = q0.elapsed_time(q1) / preps
        poly_collide = float(pcnt.item()) / (npoly * world * preps)
        exact_bytes = int(kk.to(torch.int64).sum().item()) * 8 + 3 * npoly
        padded_gbs = 259 * npoly / (poly_ms * 1e-3) / 1e9
        poly_leg = {"metric": "poly_pair_tests_per_s", "value": npoly * world * preps / pel, "pairs_per_gpu": npoly, "reps": preps,
                    "ms_per_pass": pel / preps * 1e3, "kernel_ms": round(poly_ms, 5), "collide_rate": poly_collide,
                    "roofline": {"bound": "hbm", "kernel": "sat_poly_kernel<16, 5, true>", "achieved": round(padded_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(padded_gbs / HBM_PEAK_GBS, 4),
                                 "algorithmic_bytes_per_launch": 259 * npoly,
                                 "bytes_note": "259 B/pair = the padded layout f32[2][16][n] x 2 + 2 count bytes + 1 result byte (SURVEY.md §8d); every "
                                               "row below a wave's largest vertex count is read",
                                 "exact_GBs": round(exact_bytes / (poly_ms * 1e-3) / 1e9, 1), "exact_frac": round(exact_bytes / (poly_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "traffic": None, "step_ms_distribution": step_distribution(poly_step, preps)},
                    "workload": "config5: K ~ U{3..16} convex polygons, SoA [2][16][n], true normals"}
        _, poly_leg["value_kernels_only"] = R.rank_spread(poly_ms, npoly, poly_leg["roofline"])
        poly_leg["reduce_us"] = R.time_reduce(1, before=poly_step, reps=10)
        c = counts.get("sat_poly.config5")
        if c and npoly == c.get("pairs"):
            lane = npoly / (poly_ms * 1e-3) * c["valu_instr_per_pair"] / 1e12
            poly_leg["valu_roofline"] = {"bound": "valu", "achieved": round(lane, 2), "peak": VALU_PEAK_TLANE, "unit": "T VALU lane-instr/s",
                                         "frac": round(lane / VALU_PEAK_TLANE, 4), "valu_instr_per_pair": c["valu_instr_per_pair"],
                                         "instr_source": c.get("source")}
            if c.get("hbm_bytes_per_launch"):
                poly_leg["roofline"]["traffic"] = c["hbm_bytes_per_launch"]
                poly_leg["roofline"]["traffic_source"] = "recorded, not measured in this run: %s" % c.get("source")
        # ---- the same 1e7 pairs as a BINNED batch (include/c2d.h "binned polygon batches"): one bin per (ka, kb), so the bytes
        # that move are the vertices (+ one result byte).  The binning pass is a one-off conversion, timed separately.
        torch.cuda.synchronize()
        eng.poly_bins_from_padded(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), npoly, KMAX, args.poly_bin_granularity, stream=sh).close()  # allocator warm
        tb0 = time.perf_counter()
        bins = eng.poly_bins_from_padded(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), npoly, KMAX, args.poly_bin_granularity, stream=sh)
        bin_ms = (time.perf_counter() - tb0) * 1e3
        bcnt = torch.zeros(1, dtype=torch.int64, device=dev)

        def binned_step():
            eng.sat_poly_pairs_binned(bins, bcnt.data_ptr(), stream=sh)

        prewarm(binned_step)
        torch.cuda.synchronize()
        bcnt.zero_()
        barrier()
        torch.cuda.synchronize()
        b0 = time.perf_counter()
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record(stream)
        for _ in range(preps):
            binned_step()
        w1.record(stream)
        all_reduce_sum(bcnt)
        torch.cuda.synchronize()
        barrier()
        bel = shd.max_over_ranks(time.perf_counter() - b0, dev)
        eng.check_async()
        binned_ms = w0.elapsed_time(w1) / preps
        bout = torch.empty(npoly, dtype=torch.uint8, device=dev)
        bins.results(bout.data_ptr(), stream=sh)
        torch.cuda.synchronize()
        same = int((bout == pout).sum().item())
        if same != npoly:
            raise SystemExit(f"PARITY FAILURE: the binned polygon path differs from the padded one on {npoly - same} of {npoly} pairs")
        moved = bins.bytes
        exact_gbs = exact_bytes / (binned_ms * 1e-3) / 1e9
        poly_leg["binned"] = {"metric": "poly_pair_tests_per_s (binned batch, one bin per (ka, kb))" if args.poly_bin_granularity == 1 else
                                        f"poly_pair_tests_per_s (binned batch, sizes rounded up to {args.poly_bin_granularity} rows)",
                              "value": npoly * world * preps / bel, "kernel_ms": round(binned_ms, 5), "ms_per_pass": bel / preps * 1e3, "bins": len(bins),
                              "collide_rate": float(bcnt.item()) / (npoly * world * preps), "bytes_moved_per_pair": moved / npoly,
                              "roofline": {"bound": "hbm", "kernel": "sat_poly_binned_kernel", "achieved": round(exact_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(exact_gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": exact_bytes,
                                           "bytes_note": "EXACT bytes of the workload: 8 B per real vertex + 2 count bytes + 1 result byte per pair (SURVEY.md §8d); "
                                                         "the bins themselves move %.1f B/pair" % (moved / npoly),
                                           "traffic": None, "step_ms_distribution": step_distribution(binned_step, preps)},
                              "binning_pass_ms": round(bin_ms, 3),
                              "binning_note": "c2d_poly_bins_from_padded, second call (wall time incl. the allocation of the bins' block): one-off conversion of "
                                              "the padded batch (a stable counting sort that moves every vertex once), not part of a test; results are "
                                              "returned in the padded order by c2d_poly_bins_results",
                              "parity": f"booleans equal to the padded entry point's on {same} of {npoly} pairs"}
        _, poly_leg["binned"]["value_kernels_only"] = R.rank_spread(binned_ms, npoly, poly_leg["binned"]["roofline"])
        poly_leg["binned"]["reduce_us"] = R.time_reduce(1, before=binned_step, reps=10)
        c = counts.get("sat_poly_binned.config5")
        if c and npoly == c.get("pairs") and args.poly_bin_granularity == 1:
            lane = npoly / (binned_ms * 1e-3) * c["valu_instr_per_pair"] / 1e12
            poly_leg["binned"]["valu_roofline"] = {"bound": "valu", "achieved": round(lane, 2), "peak": VALU_PEAK_TLANE, "unit": "T VALU lane-instr/s",
                                                   "frac": round(lane / VALU_PEAK_TLANE, 4), "valu_instr_per_pair": c["valu_instr_per_pair"],
                                                   "instr_source": c.get("source")}
            if c.get("hbm_bytes_per_launch"):
                poly_leg["binned"]["roofline"]["traffic"] = c["hbm_bytes_per_launch"]
                poly_leg["binned"]["roofline"]["traffic_source"] = "recorded, not measured in this run: %s" % c.get("source")
        bins.close()
        del bout
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            poly_keep = (vx.cpu().numpy(), vy.cpu().numpy(), kk.cpu().numpy(), pout.cpu().numpy())
        del vx, vy, kk

        # the same entry point on a tight layout of small polygons (triangles and quadrilaterals, 4 vertex rows: 67 B/pair)
        vx4, vy4, kk4 = torch_random_convex_polygons(torch, dev, npoly, seed=0xBEEF + rank, kmin=3, kmax=4, rows=4)
        torch.cuda.synchronize()

        def small_step():
            eng.sat_poly_pairs_rows(vx4.data_ptr(), vy4.data_ptr(), kk4.data_ptr(), npoly, 4, pout.data_ptr(), None, stream=sh)

        prewarm(small_step)
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        for _ in range(preps):
            small_step()
        r1.record(stream)
        torch.cuda.synchronize()
        eng.check_async()
        sms = r0.elapsed_time(r1) / preps
        sgbs = 67 * npoly / (sms * 1e-3) / 1e9
        poly_leg["small_polygons"] = {"metric": "poly_pair_tests_per_s (K ~ U{3..4}, 4-row layout, per GPU)", "value": npoly / (sms * 1e-3),
                                      "kernel_ms": round(sms, 5), "collide_rate": float(pout.to(torch.int64).sum().item()) / npoly,
                                      "roofline": {"bound": "hbm", "kernel": "sat_poly4_kernel", "achieved": round(sgbs, 1),
                                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sgbs / HBM_PEAK_GBS, 4),
                                                   "algorithmic_bytes_per_launch": 67 * npoly, "traffic": None}}
        c4 = counts.get("sat_poly4.small_polygons", {})
        if c4.get("pairs") == npoly and c4.get("hbm_bytes_per_launch"):
            poly_leg["small_polygons"]["roofline"]["traffic"] = c4["hbm_bytes_per_launch"]
            poly_leg["small_polygons"]["roofline"]["traffic_source"] = "recorded, not measured in this run: %s" % c4.get("source")
        del vx4, vy4, kk4, pout
    # what the later legs and the result line read
    R.poly_keep, R.poly_leg = poly_keep, poly_leg


def cpu_baselines_and_parity(R) -> None:
    """CPU baseline + full-size parity: oracle port on this host, rank 0, N = 1 only"""
    # state the legs before this one left on R
    PS, S, args, count_after_timed, mc, mc_hits_one_step, mc_poly, mc_poly_hits_one_step, mc_poly_scenes_keep, n = R.PS, R.S, R.args, R.count_after_timed, R.mc, R.mc_hits_one_step, R.mc_poly, R.mc_poly_hits_one_step, R.mc_poly_scenes_keep, R.n
    out, planes, poly_keep, poly_leg, psc, rank, sc, scenes_keep, scenes_leg, wl = R.out, R.planes, R.poly_keep, R.poly_leg, R.psc, R.rank, R.sc, R.scenes_keep, R.scenes_leg, R.wl
    world, = R.world,
    # ---- CPU baseline + full-size parity: oracle port on this host, rank 0, N = 1 only ------------------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu as oracle  # checker / reported baseline only

        oracle.set_num_threads(oracle.usable_cores())  # the box's CPU share, not the host's thread count
        budget = args.cpu_seconds

        # config 2: every boolean of the batch the timed steps produced
        host_planes = planes.cpu().numpy()
        gpu_out = out.cpu().numpy()
        ref, ref_cnt = oracle.sat_rect_pairs_verts(host_planes)  # warm-up + parity check
        equal = int((ref == gpu_out).sum())
        parity = {"config2": f"booleans equal on {equal} of {n}"}
        if equal != n or ref_cnt * args.steps != count_after_timed:
            raise SystemExit(f"PARITY FAILURE: GPU booleans differ from the CPU oracle ({parity['config2']}; count {count_after_timed} vs {ref_cnt} x {args.steps})")
        reps, c0 = 0, time.perf_counter()
        while True:
            oracle.sat_rect_pairs_verts(host_planes)
            reps += 1
            if time.perf_counter() - c0 >= budget:
                break
        cel = time.perf_counter() - c0
        cpu_baseline = {"value": n * reps / cel, "unit": "pair_tests/s", "cores": oracle.num_threads(), "kind": "port", "flags": oracle.build_info(),
                        "sample": f"all {n} pairs of the workload x {reps} passes ({cel:.1f} s), OpenMP", "parity": parity["config2"]}
        del host_planes
        if mc is not None:
            # consecutive sample ranges of the same stream; the range boundaries are chosen so that the walk passes through
            # done == S exactly, where the oracle's running hit count must EQUAL the GPU's for one mc_step (config 3 at its
            # stated size, hit for hit: ccp.cu:135-139 with utils.cu:144-184 per sample)
            ms, done, h, h_at_s = 4_000_000, 0, 0, None
            c0 = time.perf_counter()
            while True:
                m = min(ms, S - done) if done < S else ms
                h += oracle.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, done, m)
                done += m
                if done == S:
                    h_at_s = h
                if done >= S and time.perf_counter() - c0 >= budget:  # (never stops short of S: the parity check needs all of it)
                    break
            cel = time.perf_counter() - c0
            mc["parity"] = f"hits equal on {S} of {S} samples ({mc_hits_one_step} hits)"
            if h_at_s != mc_hits_one_step:
                raise SystemExit(f"PARITY FAILURE: Monte-Carlo hit count over the first {S} samples: GPU {mc_hits_one_step}, CPU oracle {h_at_s}")
            mc["cpu_baseline"] = {"value": done / cel, "unit": "samples/s", "cores": oracle.num_threads(), "kind": "port", "flags": oracle.build_info(),
                                  "sample": f"first {done} samples of the same stream ({cel:.1f} s), OpenMP", "probability": h / done,
                                  "parity": mc["parity"]}
        if mc_poly is not None:  # the polygon Monte-Carlo leg: all PS samples of one step, hit for hit, then the timed walk
            done, h, h_at_s = 0, 0, None
            c0 = time.perf_counter()
            while True:
                m = min(4_000_000, PS - done) if done < PS else 4_000_000
                h += oracle.mc_poly_pair(psc["robot"], psc["pos"], psc["theta"], psc["obstacle"], psc["std_dev"], 1234, 0, done, m)
                done += m
                if done == PS:
                    h_at_s = h
                if done >= PS and time.perf_counter() - c0 >= budget:
                    break
            cel = time.perf_counter() - c0
            mc_poly["parity"] = f"hits equal on {PS} of {PS} samples ({mc_poly_hits_one_step} hits)"
            if h_at_s != mc_poly_hits_one_step:
                raise SystemExit(f"PARITY FAILURE: polygon Monte-Carlo hit count over the first {PS} samples: GPU {mc_poly_hits_one_step}, CPU oracle {h_at_s}")
            mc_poly["cpu_baseline"] = {"value": done / cel, "unit": "samples/s", "cores": oracle.num_threads(), "kind": "port", "flags": oracle.build_info(),
                                       "sample": f"first {done} samples of the same stream ({cel:.1f} s), OpenMP", "probability": h / done,
                                       "parity": mc_poly["parity"]}
        if mc_poly is not None and "scenes" in mc_poly:  # its adaptive dataset: the first scenes of the shard, hits and stop points, then the oracle's rate
            pp_tab, ps_tab, p_scn, p_rob9, g_h, g_u, base0 = mc_poly_scenes_keep
            chk, smp = 0, 0
            c0 = time.perf_counter()
            while chk < len(p_scn):
                m = min(500, len(p_scn) - chk)
                rh, ru, _, rt = oracle.mc_poly_scenes(p_rob9, pp_tab, ps_tab, p_scn[chk:chk + m], wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 120_000, 11, base0 + chk)
                if not (np.array_equal(rh, g_h[chk:chk + m]) and np.array_equal(ru, g_u[chk:chk + m])):
                    raise SystemExit(f"PARITY FAILURE: adaptive polygon scenes {chk}..{chk + m}: hit or sample counts differ from the CPU oracle")
                chk += m
                smp += int(rt)
                if time.perf_counter() - c0 >= budget:
                    break
            cel = time.perf_counter() - c0
            mc_poly["scenes"]["parity"] = f"hits and sample counts equal on {chk} of {chk} scenes checked"
            mc_poly["scenes"]["cpu_baseline"] = {"value": chk / cel, "unit": "scenes/s", "samples_per_s": smp / cel, "cores": oracle.num_threads(), "kind": "port", "flags": oracle.build_info(),
                                                 "sample": f"first {chk} scenes of the shard ({smp} samples, {cel:.1f} s), OpenMP over scenes", "parity": mc_poly["scenes"]["parity"]}
        if poly_keep is not None:  # config 5: every boolean of the 16-row batch
            hvx, hvy, hk, hout = poly_keep
            ref, ref_cnt = oracle.sat_poly_pairs(hvx, hvy, hk)
            equal = int((ref == hout).sum())
            poly_leg["parity"] = f"booleans equal on {equal} of {len(ref)}"
            if equal != len(ref):
                raise SystemExit("PARITY FAILURE: GPU polygon booleans differ from the CPU oracle (%s)" % poly_leg["parity"])
            reps, c0 = 0, time.perf_counter()
            while True:
                oracle.sat_poly_pairs(hvx, hvy, hk)
                reps += 1
                if time.perf_counter() - c0 >= budget:
                    break
            cel = time.perf_counter() - c0
            poly_leg["cpu_baseline"] = {"value": len(ref) * reps / cel, "unit": "pair_tests/s", "cores": oracle.num_threads(), "kind": "port", "flags": oracle.build_info(),
                                        "sample": f"all {len(ref)} pairs of the workload x {reps} passes ({cel:.1f} s), OpenMP", "parity": poly_leg["parity"]}
            poly_keep = None
        if scenes_keep is not None:  # config 4: the oracle's adaptive loop on the first scenes of the shard, chunk by chunk
            tp, ts = scenes_keep["tables"]
            chunk, done, cpu_samples, bad = 2000, 0, 0, 0
            c0 = time.perf_counter()
            while done < len(scenes_keep["scenes"]):
                m = min(chunk, len(scenes_keep["scenes"]) - done)
                rh, ru, _, rt = oracle.mc_scenes(tp, ts, scenes_keep["scenes"][done:done + m], 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY,
                                                 args.scenes_max_samples, 11, scenes_keep["base"] + done)
                bad += int((rh != scenes_keep["hits"][done:done + m]).sum()) + int((ru != scenes_keep["used"][done:done + m]).sum())
                cpu_samples += rt
                done += m
                if time.perf_counter() - c0 >= budget:
                    break
            cel = time.perf_counter() - c0
            scenes_leg["parity"] = (f"hits and sample counts equal on {done} of {done} data points checked" if not bad
                                    else f"{bad} mismatching counts in the first {done} data points")
            if bad:
                raise SystemExit("PARITY FAILURE: adaptive Monte-Carlo results differ from the CPU oracle (%s)" % scenes_leg["parity"])
            scenes_leg["cpu_baseline"] = {"value": done / cel, "unit": "data_points/s", "samples_per_s": cpu_samples / cel, "cores": oracle.num_threads(),
                                          "kind": "port", "flags": oracle.build_info(), "sample": f"first {done} data points of the shard ({cpu_samples} samples, {cel:.1f} s), OpenMP over scenes",
                                          "parity": scenes_leg["parity"]}
            if "fixed_hits" in scenes_keep:  # the fixed-samples sub-leg: its first 2000 data points against the oracle
                fh = scenes_keep["fixed_hits"]
                rh, ru, _, _ = oracle.mc_scenes(tp, ts, scenes_keep["scenes"][:len(fh)], 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY,
                                                1000, 11, scenes_keep["base"])
                fbad = int((rh != fh).sum()) + int((ru != 1000).sum())
                scenes_leg["fixed_samples"]["parity"] = f"hits equal on {len(fh) - fbad} of {len(fh)} data points checked"
                if fbad:
                    raise SystemExit("PARITY FAILURE: fixed-samples Monte-Carlo hits differ from the CPU oracle")
            scenes_keep = None
    # what the later legs and the result line read
    R.cpu_baseline, = cpu_baseline,


def emit(R) -> None:
    # state the legs before this one left on R
    args, cdist, collide_rate, cpu_baseline, dev_info, dist, elapsed, eng, host_leg, mask_leg = R.args, R.cdist, R.collide_rate, R.cpu_baseline, R.dev_info, R.dist, R.elapsed, R.eng, R.host_leg, R.mask_leg
    mc, mc_poly, n, poly_leg, pose_leg, rank, rccl_library, rccl_version, real_stdout, reduce_impl = R.mc, R.mc_poly, R.n, R.poly_leg, R.pose_leg, R.rank, R.rccl_library, R.rccl_version, R.real_stdout, R.reduce_impl
    roofline, scenes_leg, use_dist, value, world = R.roofline, R.scenes_leg, R.use_dist, R.value, R.world
    if rank == 0:
        line = {
            "metric": "sat_pair_tests_per_s", "value": value, "unit": "pair_tests/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_kernels_only": R.value_kernels_only, "reduce_us": R.reduce_us, "scaling_detail": R.scaling_detail,
            "config": {"workload": ("config2: %s random OBB pairs per GPU, 16 SoA vertex planes -> u8 booleans, single SAT overlap kernel"
                                    % ("1e7" if n == 10_000_000 else str(n))),
                       "pairs_per_gpu": n, "bytes_per_pair": BYTES_PER_PAIR, "collide_rate": round(collide_rate, 5),
                       "parallelism": f"pairs sharded over {world} GPU(s), one process per GPU, no data-path collective, one sum of the hit count per leg",
                       "reduce": reduce_impl,
                       "ranks_in_reduce": (cdist.world_size if cdist is not None else (dist.get_world_size() if use_dist else 1)),
                       "rccl_version": rccl_version, "rccl_library": rccl_library, "torch_backend": args.backend if use_dist else None},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "mask_output": mask_leg, "pose_format": pose_leg, "host_resident": host_leg, "mc": mc, "mc_poly": mc_poly, "scenes": scenes_leg, "poly": poly_leg,
            "device": dev_info["name"], "pci_bus_id": dev_info["pci_bus_id"],
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if cdist is not None:
        cdist.close()
    eng.close()
    if use_dist:
        dist.destroy_process_group()

def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pairs", type=int, default=10_000_000, help="rectangle pairs per GPU (config 2: 1e7)")
    ap.add_argument("--mc-samples", type=int, default=100_000_000, help="MC samples per GPU (config 3: 1e8)")
    ap.add_argument("--mc-reps", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed device wake-up before the W warm-up steps: the first ~15 ms of load after idle run "
                         "up to 12 %% slower while the power manager ramps clocks (profiles/r01a trace)")
    ap.add_argument("--poly-scenes", type=int, default=200_000, help="scenes per GPU of the adaptive polygon Monte-Carlo sub-leg (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for rendezvous / barriers; nccl (= RCCL over xGMI) is the measured path, gloo only "
                         "rehearses the N > 1 code path on a box with fewer GPUs than ranks (together with --share-device; the "
                         "c2d reduce then uses its file rehearsal transport, since RCCL refuses two ranks on one device)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: initialise the process group and run the collectives even with one rank")
    ap.add_argument("--reduce", default="c2d", choices=["c2d", "torch"],
                    help="who sums the hit counters over ranks: libc2d's c2d_dist (RCCL, default) or torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mc", action="store_true")
    ap.add_argument("--no-pose", action="store_true")
    ap.add_argument("--scenes", type=int, default=4_000_000,
                    help="config 4 data points per GPU (1e6 scenes x 32 obstacle instances / 8 GPUs); 0 = skip the leg")
    ap.add_argument("--scenes-max-samples", type=int, default=120_000)
    ap.add_argument("--poly-pairs", type=int, default=10_000_000, help="config 5 polygon pairs per GPU; 0 = skip the leg")
    ap.add_argument("--poly-reps", type=int, default=20)
    ap.add_argument("--poly-bin-granularity", type=int, default=1,
                    help="binned polygon leg: polygon sizes rounded up to this many rows per bin (1 = one bin per (ka, kb): the exact bytes)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args.gpus))  # nothing GPU-related has been imported yet

    R = Run()
    setup(args, R)
    for leg in (leg_pairs, leg_mask_output, leg_pose_format, leg_host_resident, leg_mc, leg_mc_poly, leg_scenes, leg_poly, cpu_baselines_and_parity, emit):
        leg(R)


if __name__ == "__main__":
    main()

