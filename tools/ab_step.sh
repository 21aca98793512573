#!/bin/bash
# A/B of library variants (tools/build_variant.py) on the whole cfg3 step: ms/step and the averages of the kernels whose
# names contain $AB_KERNELS (default: the forward kernels).   tools/ab_step.sh base g2 base
PAT=${AB_KERNELS:-fwd}
for v in "$@"; do
  if [ "$v" = base ]; then unset GNNTRK_LIB; else export GNNTRK_LIB=$(pwd)/tools/_bin/variants/$v/libgnntrk.so; fi
  python bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | PAT=$PAT V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
ks=d['kernels']
print('==', os.environ['V'], round(d['ms_per_step'],3), {k.replace('mlp16_','')[:40]: round(v['avg_ms'],3) for k,v in ks.items() if os.environ['PAT'] in k})"
done
