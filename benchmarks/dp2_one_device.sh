#!/bin/bash
# bench.py --gpus 2 with BOTH ranks on one MI355X (GS_DIST_BACKEND=gloo bootstrap: RCCL refuses two ranks per device): the multi-rank
# bench path end to end -- a functional record of the N > 1 path, not a scaling number.   bash benchmarks/dp2_one_device.sh <outdir> [name "ENV.. -- bench args"]...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-dp2}; shift
mkdir -p $O
cd $R
run() { # name, "env -- args"
  name=$1; envs=${2%%--*}; args=${2#*--}
  env $envs GS_FAULT_DUMP_S=80 GS_DIST_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 295$((10 + RANDOM % 80)) bench.py --gpus 2 --warmup 8 --no-cpu-baseline --no-aux $args 2>$O/$name.err | tail -1 > $O/$name.json
  python - $O/$name.json $name $O/$name.err <<'P'
import json, sys, re
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print(sys.argv[2] + ":", json.dumps({"n_gpus": d["n_gpus"], "ms_per_step": d["ms_per_step"], "value": d["value"], "allreduce": c.get("allreduce"), "parallelism": c.get("parallelism"), "global_batch": c.get("global_batch"), "loss_after": c.get("loss_after"), "dp_schedule": d.get("dp_schedule")}))
except Exception as ex:
    msg = [l for l in open(sys.argv[3]) if "RuntimeError" in l or "Error:" in l]
    print(sys.argv[2], "FAILED:", (msg[-1].strip()[:300] if msg else repr(ex)))
P
}
if [ $# -eq 0 ]; then
  run peer "GS_DP_PEER_PUSH=1 -- --steps 64"
  run eager "GS_DP_NATIVE=0 -- --steps 64"
else
  while [ $# -gt 1 ]; do run "$1" "$2"; shift 2; done
fi
