#!/bin/bash
# tools/k1h_variant.sh <name> <K1H_EXP list> — timing-experiment build of K1h (gen_k1h.py: K1H_EXP=noload,nopass,... — results are WRONG,
# only the clock is of interest; `timers` keeps the results and adds section clocks) linked against the in-tree objects into
# tools/lib_k1h_<name>.so (bench.py --lib ...).  K1H_CXXFLAGS adds compiler flags (-DK1H_STATIC_PRIO=1).
set -e
NAME=$1; EXP=$2
C=$(cd $(dirname $0)/../ntcard_amd/csrc && pwd)
T=/tmp/k1hv_$NAME; mkdir -p $T
cp $C/*.hpp $C/ntc_sketch_k1h.hip $C/ntc_sketch_k1h_body.hip $T/
K1H_EXP=$EXP python3 $C/gen_k1h.py $T/ntc_k1h_gen.inc > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$T -I$C -I$C/../../include ${K1H_CXXFLAGS:-}"
for p in 0 1 2 3; do /opt/rocm/bin/hipcc $FLAGS -DK1H_PART=$p -c $T/ntc_sketch_k1h_body.hip -o $T/ntc_sketch_k1h_p$p.o & done
/opt/rocm/bin/hipcc $FLAGS -c $T/ntc_sketch_k1h.hip -o $T/ntc_sketch_k1h.o
wait
OBJS=$(ls $C/build/*.o | grep -v ntc_sketch_k1h)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $(dirname $0)/lib_k1h_$NAME.so $OBJS $T/ntc_sketch_k1h.o $T/ntc_sketch_k1h_p0.o $T/ntc_sketch_k1h_p1.o $T/ntc_sketch_k1h_p2.o $T/ntc_sketch_k1h_p3.o -ldl
echo built tools/lib_k1h_$NAME.so
