# usage (on a GPU box): bash tools/exp/isef_run.sh — isef_bench for the default smoothing, a forced short warm-up (repairs everywhere), the
# sequential route and three more smoothing factors, then a kernel trace of the default
cd $GRAFT_REPO_ROOT
echo "== default"; ./tools/exp/isef_bench | grep -v "alone\|^   "
echo "== W=4 (repairs)"; ZIGNAL_HIP_ISEF_W=4 ./tools/exp/isef_bench | grep -v "alone\|^   "
echo "== serial"; ZIGNAL_HIP_ISEF_SERIAL=1 ./tools/exp/isef_bench | tail -1
for b in 0.95 0.7 0.5; do echo "== b=$b"; ISEF_B=$b ./tools/exp/isef_bench | grep -v "alone\|^   " | tail -2; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/p -o r -- $GRAFT_REPO_ROOT/tools/exp/isef_bench 4096 4096 2>/dev/null | tail -1; python3 $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/p/r_results.db | cut -c1-60,111-200
