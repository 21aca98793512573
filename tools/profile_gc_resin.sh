#!/bin/bash
# Kernel trace of GraphConstructionResIN(hidden_dim=40) forward + backward in fp32 (wide fused kernels only)
# and in bf16 storage -> gpurun_out/<tag>_gc_resin_{fp32,bf16}_kernel_stats.md
TAG=${1:-r04}
ROOT=$(pwd); OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in fp32w fused; do
  rm -rf /tmp/p_gc
  rocprofv3 --kernel-trace -d /tmp/p_gc -o k -- python $ROOT/tools/bench_gc_resin.py --mode $m --steps 5 > /dev/null 2>&1
  n=$([ $m = fp32w ] && echo fp32 || echo bf16)
  python $ROOT/tools/rocpd_summary.py /tmp/p_gc/k_results.db > $ROOT/$OUT/${TAG}_gc_resin_${n}_kernel_stats.md
done
