#!/bin/bash
# Kernel trace of the headline run only -> gpurun_out/<tag>_kernel_stats.md and <tag>_timeline.md
TAG=${1:-step}
ROOT=$(pwd); OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/p_k
rocprofv3 --kernel-trace -d /tmp/p_k -o k -- python $ROOT/bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1
python $ROOT/tools/rocpd_summary.py /tmp/p_k/k_results.db > $ROOT/$OUT/${TAG}_kernel_stats.md
(cd $ROOT/tools && python rocpd_timeline.py /tmp/p_k/k_results.db --mark "${MARK:-no_minmax}" > $ROOT/$OUT/${TAG}_timeline.md)
