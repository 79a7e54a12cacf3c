#!/bin/bash
# Round 6 — bench.py's N > 1 line with the fields that make a scaling curve readable (roofline.kernel_ms_ranks, frac_slowest_rank,
# reduce_us, value_kernels_only, scaling_detail), at the largest rank count the one-GPU box allows: FIVE ranks sharing GPU 0 (the
# pool's process guard lets six processes hold one card open; torch's launcher is the sixth), gloo rendezvous, the rehearsal build
# of the library (file transport) — and the one-rank line with the real RCCL (--force-dist) beside it.  bash profiles/r06_rehearse.sh
set -o pipefail
R=$PWD
O=$R/gpurun_out/r06_rehearsal
mkdir -p $O
N=5
echo "== bench.py --gpus $N, shared device"
timeout -k 10 900 python3 bench.py --gpus $N --share-device --backend gloo --steps 20 --warmup 5 > $O/bench_n${N}.json 2> $O/bench_n${N}.err; echo "rc=$?"
echo "== bench.py, one rank, real RCCL in the reduce (--force-dist)"
timeout -k 10 600 python3 bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "rc=$?"
python3 - <<PY
import json
for f in ("bench_n$N", "bench_force_dist"):
    j = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f, "n_gpus", j["n_gpus"], "value %.4g" % j["value"], "kernels_only %.4g" % j["value_kernels_only"], "kernel_ms_ranks", r["kernel_ms_ranks"],
          "frac", r["frac"], "frac_slowest_rank", r["frac_slowest_rank"], "reduce_us", {k: j["reduce_us"][k] for k in ("event_median", "host_median")},
          "scaling_detail", {k: v for k, v in j["scaling_detail"].items() if k != "note"})
PY
echo "== done"
