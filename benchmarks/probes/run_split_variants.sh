for v in nomfma nog nostore mfmaonly nothing; do
  echo "== $v"; GS_LIB=benchmarks/probes/_lib/libgs_tiled_$v.so timeout 100 python benchmarks/micro_split.py pool 2>/dev/null | grep "split_tiled"
done
