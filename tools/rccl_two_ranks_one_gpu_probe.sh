#!/bin/bash
# Can two RCCL ranks share one GPU? (No: "Duplicate GPU detected" -- kept as the record of why the >1-rank
# GPU tests exchange over gloo.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline
