#!/bin/bash
# Diagnostics: build an A/B copy of the library with extra -D flags for ONE source file (the other objects are reused
# from graphsage_amd/_C).  Loaded with GS_LIB=<path>.   bash benchmarks/probes/build_variant.sh <name> <file.hip> <flags...>
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; SRC=$2; shift 2
OUT=$R/benchmarks/probes/_lib
mkdir -p $OUT
python -m graphsage_amd.build > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-finite-math-only \
  "$@" -c $R/graphsage_amd/csrc/$SRC -o $OUT/$NAME.o
OBJS=$(ls $R/graphsage_amd/_C/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libgs_$NAME.so $OBJS $OUT/$NAME.o
rm -f $OUT/$NAME.o
echo $OUT/libgs_$NAME.so
