#!/bin/bash
# tools/ab_build.sh <name> [extra hipcc flags] — A/B build of the library's hand-written sources with extra flags (K1h's generated bodies are taken from the
# in-tree objects: tools/k1h_variant.sh builds variants of those), written to tools/lib_<name>.so (bench.py --lib tools/lib_<name>.so runs it; the product has no override).
set -e
NAME=$1; shift
C=$(cd $(dirname $0)/../ntcard_amd/csrc && pwd)
T=/tmp/ab_$NAME; mkdir -p $T
for f in ntc_kernels ntc_sketch_hf ntc_sketch_k1h ntc_apply ntc_engine; do
  /opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$C -c $C/$f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $(dirname $0)/lib_$NAME.so $T/*.o $C/build/ntc_estimator.o $C/build/ntc_sketch_k1h_p0.o $C/build/ntc_sketch_k1h_p1.o $C/build/ntc_sketch_k1h_p2.o $C/build/ntc_sketch_k1h_p3.o -ldl
echo built tools/lib_$NAME.so
