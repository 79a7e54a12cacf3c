#!/bin/bash
# Round 5 — the N > 1 paths of bench.py and of the drivers at the largest rank count the one-GPU box allows.  The pool lets at
# most six processes hold one card open (a run with eight was killed by its process guard: gpurun_out/r5c_call.log), and the
# N = 8 case is the round-end driver's alone.  bench.py: FIVE ranks share GPU 0 (torch's launcher is the sixth process), gloo
# rendezvous; drivers: SIX ranks (their launcher never opens the card); the rehearsal build of the library (file transport) in
# front of the product one.  bash profiles/r05_rehearse.sh
set -o pipefail
R=$PWD
O=$R/gpurun_out/r05_rehearsal
mkdir -p $O
N=5
echo "== bench.py --gpus $N, shared device"
timeout -k 10 900 python3 bench.py --gpus $N --share-device --backend gloo --steps 20 --warmup 5 > $O/bench_n${N}.json 2> $O/bench_n${N}.err; echo "rc=$?"
echo "== bench.py, one rank, same flags"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "rc=$?"
export LD_LIBRARY_PATH=$R/convex-2d-gpu-collision-detection_amd/lib-rehearsal:$LD_LIBRARY_PATH
export C2D_SHARE_DEVICE=1
N=6
B=$R/convex-2d-gpu-collision-detection_amd/bin
D=/tmp/c2d_r05_rehearsal
rm -rf $D && mkdir -p $D
echo "== generate_dataset --gpus $N"
timeout -k 10 600 $B/generate_dataset --data_dir $D/gen$N -n 12 -b 100000 --num_poses 65536 --num_variances 65536 --seed 7 --gpus $N > $O/generate_dataset_n${N}.json 2> $O/generate_dataset_n${N}.err; echo "rc=$?"
timeout -k 10 600 $B/generate_dataset --data_dir $D/gen1 -n 12 -b 100000 --num_poses 65536 --num_variances 65536 --seed 7 > $O/generate_dataset_n1.json 2> $O/generate_dataset_n1.err; echo "rc=$?"
echo "== compute_collision_probability --gpus $N (dataset mode on generate_dataset's scenes, then config 3 sharded)"
mkdir -p $D/in && python3 - <<PY
import numpy as np, glob
for k, f in enumerate(sorted(glob.glob("$D/gen1/[0-9]*.npy"), key=lambda p: int(p.split("/")[-1][:-4]))):
    a = np.load(f)                       # rows: x, y, cp, var_idx, pose_idx  ->  input rows: x, y, var_idx, pose_idx
    np.save("$D/in/%d.npy" % k, np.ascontiguousarray(a[:, [0, 1, 3, 4]]))
PY
for W in $N 1; do
  mkdir -p $D/ccp$W/meta && cp $D/gen1/poses.npy $D/gen1/variances.npy $D/ccp$W/ && cp $D/gen1/meta/*.npy $D/ccp$W/meta/
  G=""; [ $W -gt 1 ] && G="--gpus $W"
  timeout -k 10 600 $B/compute_collision_probability --data_in $D/in --data_out $D/ccp$W --seed 7 $G > $O/ccp_n${W}.json 2> $O/ccp_n${W}.err; echo "rc=$?"
  timeout -k 10 600 $B/compute_collision_probability --pair_samples 600000000 --seed 1234 $G > $O/ccp_pair_n${W}.json 2> $O/ccp_pair_n${W}.err; echo "rc=$?"
done
python3 - <<PY
import numpy as np, glob, json
same = 0
fs = sorted(glob.glob("$D/ccp1/[0-9]*.npy"))
for f in fs:
    same += int(np.array_equal(np.load(f).view(np.uint32), np.load(f.replace("/ccp1/", "/ccp$N/")).view(np.uint32)))
print("compute_collision_probability: %d of %d batch files identical between 1 and $N ranks" % (same, len(fs)))
fs = sorted(glob.glob("$D/gen1/[0-9]*.npy"))
same = sum(int(np.array_equal(np.load(f).view(np.uint32), np.load(f.replace("/gen1/", "/gen$N/")).view(np.uint32))) for f in fs)
print("generate_dataset: %d of %d batch files identical between 1 and $N ranks" % (same, len(fs)))
PY
echo "== done"
