#!/bin/bash
# Diagnostics: build the library of a COMMIT (default HEAD) for a same-call A/B against the working tree:
#   bash benchmarks/probes/build_head.sh [rev]  ->  benchmarks/probes/_lib/libgs_head.so   (load with GS_LIB=<path>; same ABI only)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
REV=${1:-HEAD}
T=$(mktemp -d /tmp/gs_head.XXXX)
git -C $R archive $REV graphsage_amd/csrc include | tar -x -C $T
mkdir -p $R/benchmarks/probes/_lib
for f in $T/graphsage_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-finite-math-only -c $f -o ${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/benchmarks/probes/_lib/libgs_head.so $T/graphsage_amd/csrc/*.o
rm -rf $T
echo $R/benchmarks/probes/_lib/libgs_head.so
