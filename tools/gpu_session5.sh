#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sharded_torch.py -m gpu -q > gpurun_out/pytest_torch.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_torch.log
timeout 600 python bench.py --torch-path --no-cpu-baseline > gpurun_out/bench_torchpath.log 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1
