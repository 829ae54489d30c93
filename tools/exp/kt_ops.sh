# usage (on a GPU box): bash tools/exp/kt_ops.sh <op> [<op> ...] — per-kernel times of tools/run_op.py ops (the library's kernels only), printed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for op in "$@"; do
  rm -rf /tmp/p_$op
  rocprofv3 --kernel-trace -d /tmp/p_$op -o r -- python tools/run_op.py $op 10 > /tmp/p_$op.log 2>&1
  db=$(find /tmp/p_$op -name '*.db' | head -1)
  echo "== $op"; grep -v "^W2026\|^E2026" /tmp/p_$op.log | tail -1 | cut -c1-150
  python3 tools/prof_summary.py $db | cut -c1-56,111-160 | grep "^kernel\|_ZN2zg\|rocclr" | head -${KT_LINES:-20}
done
