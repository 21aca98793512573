#!/bin/bash
# A/B of library variants on the whole step: ms/step and the forward kernels' averages
for v in "$@"; do
  if [ "$v" = base ]; then unset GNNTRK_LIB; else export GNNTRK_LIB=$(pwd)/tools/_bin/variants/$v/libgnntrk.so; fi
  python bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ks=d['kernels']
print('== $v', round(d['ms_per_step'],3), {k.replace('mlp16_','')[:40]: round(v['avg_ms'],3) for k,v in ks.items() if 'fwd' in k})"
done
