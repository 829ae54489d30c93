#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counters>" <op> [<op> ...] — per-dispatch means of rocprofv3 --pmc <counters> (one pass, --kernel-trace only)
# for tools/run_op.py ops on a gpurun box; summaries land in gpurun_out/<tag>/pmc_<op>.txt
tag=$1; shift
ctr=$1; shift
ops="$*"
mkdir -p /tmp/gr
cat > /tmp/gr/pmc_cmd.sh <<EOS
cd /tmp && export TMPDIR=/tmp && cd \$GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for op in $ops; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/$tag/pmc_\$op -o r -- python tools/run_op.py \$op 5 > gpurun_out/$tag/pmc_\$op.log 2>&1
  db=\$(find gpurun_out/$tag/pmc_\$op -name '*.db' | head -1)
  [ -n "\$db" ] && python tools/pmc_summary.py \$db zg > gpurun_out/$tag/pmc_\$op.txt 2>&1
  rm -rf gpurun_out/$tag/pmc_\$op
  echo "== \$op"; cat gpurun_out/$tag/pmc_\$op.txt | cut -c1-130
done
EOS
timeout 2400 /usr/local/graft/bin/gpurun --timeout 1200 -- "$(cat /tmp/gr/pmc_cmd.sh)" 2>&1 | grep -v "^\[gpurun\] sending\|^cd /tmp\|^mkdir\|^for op\|^  timeout\|^  db=\|^  \[ -n\|^  rm -rf\|^  echo\|^done"
