# 2 ranks on ONE GPU through gloo (RCCL refuses two ranks per device): validates the N>1 code path of bench.py.
# Bounded: every rank dumps its Python stack after 50 s and the whole run is killed after 100 s.
cd ${GRAFT_REPO_ROOT:-.}
GS_FAULT_DUMP_S=50 GS_DIST_BACKEND=gloo timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 6 --nodes 20000 --avg_degree 10 --feat_dim 64 2>&1 | grep -v "^\[W\|Warning\|warn" | tail -60
# the unsupervised (configs[3]) and RMAT (configs[4]) data-parallel paths, same bounds
GS_FAULT_DUMP_S=50 GS_DIST_BACKEND=gloo timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 20 --warmup 6 --nodes 20000 --avg_degree 10 --feat_dim 64 --unsupervised 2>&1 | grep -v "^\[W\|Warning\|warn" | tail -3 | cut -c1-400
GS_FAULT_DUMP_S=50 GS_DIST_BACKEND=gloo timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --steps 20 --warmup 6 --workload rmat --nodes 20000 --rmat-edges 400000 --feat_dim 64 2>&1 | grep -v "^\[W\|Warning\|warn" | tail -2 | cut -c1-300
