#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_* (run on the GPU box from the repo root):
#   tools/profile_round.sh r03 gpurun_out/prof_r03
# kernel trace of the headline run, FETCH_SIZE / WRITE_SIZE in separate --pmc passes (TCC slots),
# SQ counters of the backward kernels, cfg5 kernel trace, pipe utilisation of the cfg5 / DBSCAN kernels.
set -u
TAG=${1:-r03}
OUT=${2:-gpurun_out/prof_$TAG}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python $ROOT/bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 1"
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
cd /tmp
rocprofv3 --kernel-trace -d /tmp/p_k -o k -- $B > /dev/null 2>&1
python $ROOT/tools/rocpd_summary.py /tmp/p_k/k_results.db > $ROOT/$OUT/${TAG}_bench_cfg3_bf16_kernel_stats.md
(cd $ROOT/tools && python rocpd_timeline.py /tmp/p_k/k_results.db --mark no_minmax > $ROOT/$OUT/${TAG}_step_timeline.md)
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_f -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_w -o w -- $B > /dev/null 2>&1
python $ROOT/tools/make_traffic_json.py /tmp/p_f/f_results.db /tmp/p_w/w_results.db 64000000 \
  "bench.py cfg3 bf16 (32 events, E=64e6), $TAG, FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes." \
  > $ROOT/$OUT/${TAG}_hbm_traffic_bf16.json
python $ROOT/tools/rocpd_pmc_summary.py /tmp/p_f/f_results.db > $ROOT/$OUT/${TAG}_pmc_fetch_bf16.md
python $ROOT/tools/rocpd_pmc_summary.py /tmp/p_w/w_results.db > $ROOT/$OUT/${TAG}_pmc_write_bf16.md
rocprofv3 --pmc $SQ -d /tmp/p_s -o s -- $B > /dev/null 2>&1
python $ROOT/tools/rocpd_pmc_summary.py /tmp/p_s/s_results.db > $ROOT/$OUT/${TAG}_pmc_sq_bf16.md
python $ROOT/tools/make_pipe_json.py /tmp/p_s/s_results.db "bench.py cfg3 bf16, SQ counters, $TAG" > $ROOT/$OUT/${TAG}_pipe_util_cfg3.json
# LDS pass of the same command (its own run: the SQ block has eight counter slots)
LDS="SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --pmc $LDS -d /tmp/p_l -o l -- $B > /dev/null 2>&1
python $ROOT/tools/rocpd_pmc_summary.py /tmp/p_l/l_results.db > $ROOT/$OUT/${TAG}_pmc_lds_bf16.md
C5="python $ROOT/bench.py --workload cfg5 --steps 4 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace -d /tmp/p_k5 -o k -- $C5 > /dev/null 2>&1
python $ROOT/tools/rocpd_summary.py /tmp/p_k5/k_results.db > $ROOT/$OUT/${TAG}_bench_cfg5_kernel_stats.md
(cd $ROOT/tools && python rocpd_timeline.py /tmp/p_k5/k_results.db --mark knn_bbox_partial_kernel --nth-last 2 > $ROOT/$OUT/${TAG}_cfg5_timeline.md)
rocprofv3 --pmc $SQ -d /tmp/p_s5 -o s -- $C5 > /dev/null 2>&1
python $ROOT/tools/make_pipe_json.py /tmp/p_s5/s_results.db "bench.py --workload cfg5 (fp32, 200 k hits), SQ counters, $TAG" > $ROOT/$OUT/${TAG}_pipe_util_cfg5.json
rocprofv3 --pmc $SQ -d /tmp/p_sd -o s -- python $ROOT/tools/bench_dbscan.py > /dev/null 2>&1
python $ROOT/tools/make_pipe_json.py /tmp/p_sd/s_results.db "tools/bench_dbscan.py (200 k hits, 8-d, max_eps 0.5), SQ counters, $TAG" > $ROOT/$OUT/${TAG}_pipe_util_dbscan.json
ls -la $ROOT/$OUT
