#!/bin/bash
# interactions/s of the three bit-exact all-pairs kernels by body count (one GPU): the table behind strict_kernel_choice()
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "N one_thread_per_body pc<8,8> pc<16,4> default"
for nb in ${SIZES:-512 1000 2048 4096 8192 10000 16384 24576 32768 49152 65536 81920 98304 114688 131072 163840 196608 262144}; do
  line="$nb"
  for k in 1 8 16 0; do
    v=$(python bench.py --mode strict --strict-kernel $k --bodies $nb --steps 10 --warmup 2 --no-traffic --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e' % d['value'])")
    line="$line $v"
  done
  echo $line
done
} | tee gpurun_out/${TAG:-r02}_strict_kernel_sweep.txt
