#!/bin/bash
# usage: tools/gpu_bench.sh  — GPU conv tests + bench on a gpurun box, prints a compact summary
timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$(cat /tmp/runbench.sh)" 2>&1 | grep -v "^\[gpurun\] send" | python -c "
import sys,json
for line in sys.stdin:
    line=line.rstrip()
    if line.startswith('{'):
        d=json.loads(line); r=d.get('roofline') or {}
        print('value',d['value'],'ms/step',d['ms_per_step'],'| kernel_ms',r.get('kernel_ms_mean'),'GB/s',r.get('achieved'),'frac',r.get('frac'))
        for k,v in (d.get('extras') or {}).items(): print('  ',k,v)
    else: print(line)
"
