#!/bin/bash
# usage (on the GPU box): tools/ab_variant.sh <variant-name> <extras-regex> — bench.py's extras for the product library and for
# zignal_amd/variants/libzignal_hip_<variant>.so, side by side (ms per call).
v=$1; re=$2
for lib in "" "$PWD/zignal_amd/variants/libzignal_hip_$v.so"; do
  ZIGNAL_HIP_LIBRARY=$lib ZG_BENCH_EXTRAS="$re" python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('LIB', '${lib:-product}', 'headline ms', d['ms_per_step'], 'resize', d.get('resize',{}).get('ms_per_step'))
for k,v in d.get('extras',{}).items(): print('  %-64s %s' % (k, v.get('ms', v)))
"
done
