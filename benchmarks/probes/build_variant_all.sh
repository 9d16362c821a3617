#!/bin/bash
# Diagnostics: an A/B copy of the whole library with extra -D flags for EVERY source file.  bash benchmarks/probes/build_variant_all.sh <name> <flags...>
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
T=$(mktemp -d /tmp/gs_var.XXXX)
for f in $R/graphsage_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-finite-math-only "$@" -c $f -o $T/$(basename ${f%.hip}).o &
done
wait
mkdir -p $R/benchmarks/probes/_lib
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/benchmarks/probes/_lib/libgs_$NAME.so $T/*.o
rm -rf $T
echo $R/benchmarks/probes/_lib/libgs_$NAME.so
