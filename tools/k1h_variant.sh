#!/bin/bash
# tools/k1h_variant.sh <name> <K1H_EXP list> — timing-experiment build of K1h (gen_k1h.py: K1H_EXP=noload,nopass,... — results are WRONG,
# only the clock is of interest) linked against the in-tree objects into tools/lib_k1h_<name>.so (bench.py --lib ...).
set -e
NAME=$1; EXP=$2
C=$(cd $(dirname $0)/../ntcard_amd/csrc && pwd)
T=/tmp/k1hv_$NAME; mkdir -p $T
cp $C/*.hpp $C/ntc_sketch_k1h.hip $T/
mkdir -p $T/../include 2>/dev/null || true
K1H_EXP=$EXP python3 $C/gen_k1h.py $T/ntc_k1h_gen.inc > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$T -I$C -c $T/ntc_sketch_k1h.hip -o $T/ntc_sketch_k1h.o
OBJS=$(ls $C/build/*.o | grep -v ntc_sketch_k1h.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $(dirname $0)/lib_k1h_$NAME.so $OBJS $T/ntc_sketch_k1h.o -ldl
echo built tools/lib_k1h_$NAME.so
