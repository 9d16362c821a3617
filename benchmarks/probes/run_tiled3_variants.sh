#!/bin/bash
# Round 6: what bounds a stage of sage_tiled3_fwd_kernel -- the kernel's time with one component compiled out (wrong values).
#   bash benchmarks/probes/run_tiled3_variants.sh          (build here)  |  ... run   (on the GPU box)
R=$(cd "$(dirname "$0")/../.." && pwd)
V="base: nog:-DF3_DIAG_NOGA,-DF3_DIAG_NOGB nomfma:-DF3_DIAG_NOMFMA nocut:-DGS_DIAG_SPLIT_NOCUT noldsr:-DF3_DIAG_NOLDSR noldsw:-DF3_DIAG_NOLDSW nocutnomfma:-DGS_DIAG_SPLIT_NOCUT,-DF3_DIAG_NOMFMA mfmaonly:-DF3_DIAG_NOGA,-DF3_DIAG_NOGB,-DF3_DIAG_NOLDSR,-DF3_DIAG_NOLDSW,-DGS_DIAG_SPLIT_NOCUT"
if [ "$1" != "run" ]; then
  for v in $V; do n=${v%%:*}; f=$(echo ${v#*:} | tr ',' ' '); bash $R/benchmarks/probes/build_variant.sh t3_$n gs_split.hip $f > /dev/null || exit 1; done
  ls $R/benchmarks/probes/_lib/ | grep t3_
  exit 0
fi
cd $R
for v in $V; do n=${v%%:*}
  echo "== $n: $(GS_LIB=$R/benchmarks/probes/_lib/libgs_t3_$n.so timeout 200 python benchmarks/micro_split.py 2>/dev/null | grep -E 'split_alone' | tr -d ' \n')"
done
