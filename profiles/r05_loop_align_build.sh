#!/bin/bash
# experiment: device code of one .hip with every loop header aligned to 2^$2 bytes;  build.sh <file.hip> <log2 align> <out.o> [inner]
set -e
SRC=$1; A=$2; OUT=$3; MODE=${4:-all}
LL=/opt/rocm/lib/llvm/bin
B=$(basename $SRC .hip); D=$(dirname $OUT)
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Iinclude"
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o $D/$B.s $SRC 2>/dev/null
python3 - $D/$B.s $A $MODE <<'PY'
import re, sys
path, a, mode = sys.argv[1], sys.argv[2], sys.argv[3]
lines = open(path).read().split("\n")
out, n = [], 0
for i, ln in enumerate(lines):
    if re.match(r"^\.LBB\d+_\d+:", ln):
        j, note = i + 1, ln
        while j < len(lines) and re.match(r"^\s*;", lines[j]):
            note += lines[j]
            j += 1
        if ("Inner Loop Header" in note) if mode == "inner" else ("Loop Header" in note):
            out.append("\t.p2align\t%s" % a)
            n += 1
    out.append(ln)
open(path, "w").write("\n".join(out))
print("aligned", n, "loop headers in", path)
PY
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $D/$B.s -o $D/$B.dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $D/$B.out $D/$B.dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$D/$B.out -output=$D/$B.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $D/$B.hipfb -c $SRC -o $OUT
