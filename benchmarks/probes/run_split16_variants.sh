# bash benchmarks/probes/run_split16_variants.sh  (on the GPU box; the variant libraries are built beforehand with build_variant.sh)
for v in base nodma nomfma noldsr noepi mfmaonly mfmaonly_noepi base; do
  lib=benchmarks/probes/_lib/libgs_s16_$v.so
  [ -f $lib ] || continue
  echo "== $v: $(GS_LIB=$lib timeout 100 python -m benchmarks.micro_split pool16 2>/dev/null | tail -1)"
done
