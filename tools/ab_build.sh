#!/bin/bash
# tools/ab_build.sh <name> [extra hipcc flags, e.g. -DNTC_BS_TIMERS] — A/B build of the whole library with extra
# flags, written to tools/lib_<name>.so (bench.py --lib tools/lib_<name>.so runs it; the product has no override).
set -e
NAME=$1; shift
C=$(cd $(dirname $0)/../ntcard_amd/csrc && pwd)
T=/tmp/ab_$NAME; mkdir -p $T
for f in ntc_kernels ntc_sketch_hf ntc_sketch_bs ntc_apply ntc_engine; do
  /opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$C -c $C/$f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $(dirname $0)/lib_$NAME.so $T/*.o $C/build/ntc_estimator.o
echo built tools/lib_$NAME.so
