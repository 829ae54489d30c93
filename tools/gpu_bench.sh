#!/bin/bash
# usage: tools/gpu_bench.sh  — runs /tmp/runbench.sh on a gpurun box, prints a compact summary of any JSON line
timeout 1500 /usr/local/graft/bin/gpurun --timeout 900 -- "$(cat /tmp/runbench.sh)" 2>&1 | grep -v "^\[gpurun\] send" | python tools/_fmt_bench.py
