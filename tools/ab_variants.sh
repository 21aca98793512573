#!/bin/bash
# A/B of library variants (tools/build_variant.py) on the backward microbenchmark:
#   tools/ab_variants.sh out.log base prio1 prio2 ... base
OUT=$1; shift
: > $OUT
for v in "$@"; do
  if [ "$v" = base ]; then unset GNNTRK_LIB; else export GNNTRK_LIB=$(pwd)/tools/_bin/variants/$v/libgnntrk.so; fi
  echo "== $v" >> $OUT
  python tools/bench_bwd_io.py --only 0 --iters 12 $AB_ARGS 2>&1 | grep -v amdgpu.ids | grep -v "^$" >> $OUT
done
cat $OUT
