#!/bin/bash
# usage: tools/gpu_kt.sh <tag> <op> [<op> ...] — per-kernel times (rocprofv3 --kernel-trace) of tools/run_op.py ops on a gpurun box;
# summaries land in gpurun_out/<tag>/kt_<op>.txt. Optional env PRE="shell commands to run first on the box".
tag=$1; shift
ops="$*"
cat > /tmp/gr/kt_cmd.sh <<EOS
cd /tmp && export TMPDIR=/tmp && cd \$GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
$PRE
for op in $ops; do
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/$tag/kt_\$op -o r -- python tools/run_op.py \$op 20 > gpurun_out/$tag/kt_\$op.log 2>&1
  db=\$(find gpurun_out/$tag/kt_\$op -name '*.db' | head -1)
  [ -n "\$db" ] && python tools/prof_summary.py \$db > gpurun_out/$tag/kt_\$op.txt 2>&1
  rm -rf gpurun_out/$tag/kt_\$op
  echo "== \$op"; grep -v "^#" gpurun_out/$tag/kt_\$op.txt | cut -c1-60,111-190 | head -8
  grep -i "error\|assert\|Traceback" gpurun_out/$tag/kt_\$op.log | head -5
done
EOS
timeout 2400 /usr/local/graft/bin/gpurun --timeout 1500 -- "$(cat /tmp/gr/kt_cmd.sh)" 2>&1 | grep -v "^\[gpurun\] sending\|^cd /tmp\|^mkdir\|^for op\|^  timeout\|^  db=\|^  \[ -n\|^  rm -rf\|^  echo\|^  grep\|^done"
