#!/bin/bash
# tools/ab_build.sh <name> <hf_source.hip> [extra hipcc flags] — A/B build: a full libntcard_hip with one
# alternative sketch_hf source, written to tools/lib_<name>.so (select it with NTCARD_HIP_LIB=...).
set -e
NAME=$1; SRC=$2; shift 2
C=$(dirname $0)/../ntcard_amd/csrc
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$C -c $SRC -o /tmp/ab_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $(dirname $0)/lib_$NAME.so $C/build/ntc_kernels.o $C/build/ntc_sketch_fast.o /tmp/ab_$NAME.o $C/build/ntc_engine.o $C/build/ntc_estimator.o
echo built tools/lib_$NAME.so
