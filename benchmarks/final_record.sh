#!/bin/bash
# The whole record of a round in ONE gpurun call: GPU suite + driver command + kernel traces + PMC passes (round.sh), MFMA utilisation
# of the max-pool step, then -- with the digest-stamped summaries copied into profiles/ ON THE BOX -- the driver's command once more,
# so that its JSON line carries non-stale profile-sourced fields (<tag>_f_driver_bench.json).  Everything lands in gpurun_out/<outdir>/
# and in gpurun_out/<outdir>/profiles/ (= what to copy into profiles/ here).
#   [TAG=r06] bash benchmarks/final_record.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r06}
O=$R/gpurun_out/$1
mkdir -p $O/profiles
cd $R
TAG=$TAG bash benchmarks/round.sh $1 tests pmc 2>&1 | tail -40
TAG=$TAG bash benchmarks/mfma_maxpool.sh $1/mfma 2>&1 | tail -12
P=$O/profiles
cp $O/${TAG}_bench_kernel_stats.md $O/${TAG}_maxpool_kernel_stats.md $O/${TAG}_unsup_kernel_stats.md $O/${TAG}_gcn_kernel_stats.md $O/${TAG}_rmat_kernel_stats.md $P/
cp $O/pmc/k2_pmc.json $P/k2_pmc_deg492_b512_25x10_f602.json
cp $O/pmc/step_traffic.json $P/${TAG}_step_traffic.json
cp $O/pmc/mfma_util.md $P/${TAG}_mfma_util.md
cp $O/pmc/bench.json $P/${TAG}_bench.json
cp $O/${TAG}_driver_bench.json $P/${TAG}_a_driver_bench.json
{ echo "## GS_POOL_F16=1 (two fp16 pieces, the default)"; cat $O/mfma/${TAG}_mfma_util_maxpool_f16_1.md; echo; echo "## GS_POOL_F16=0 (three bf16 pieces)"; cat $O/mfma/${TAG}_mfma_util_maxpool_f16_0.md; } > $P/${TAG}_mfma_util_maxpool.md
python - "$O/${TAG}_aux_dominant.json" "$P/${TAG}_aux_dominant.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for v in d["configs"].values():
    v["source"] = "profiles/" + v["source"].split("/")[-1]
json.dump(d, open(sys.argv[2], "w"), indent=1)
PY
cp $P/* $R/profiles/
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $P/${TAG}_f_driver_bench.json 2> $O/f_driver.err
python - "$P/${TAG}_f_driver_bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("final driver command: %.2f us/step wall, events %.2f, value %.4g" % (d["ms_per_step"] * 1e3, d["ms_per_step_events"]["ms_per_step_median"] * 1e3, d["value"]))
print({k: r.get(k) for k in ("frac", "frac_algorithmic", "traffic_stale", "in_step_frac_algorithmic", "in_step_frac_counter", "wasted_traffic_ratio", "in_step_counter_stale")})
print({k: (round(v["ms_per_step"] * 1e3, 1), v.get("roofline", {}).get("dominant_kernel_stale")) for k, v in d["aux"].items() if "ms_per_step" in v})
print(d.get("roofline_step_kernel", {}).get("mfma"))
PY
