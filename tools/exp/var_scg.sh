# usage (on a GPU box): bash tools/exp/var_scg.sh — k_sc_gradient inside Image.shenCastan (4096^2 noise) with pieces compiled out
cd $GRAFT_REPO_ROOT
for v in "" SCG_NO_LOADS SCG_NO_ARITH SCG_NO_HIST SCG_NO_STORE; do
  if [ -n "$v" ]; then export ZIGNAL_HIP_LIBRARY=$GRAFT_REPO_ROOT/zignal_amd/variants/libzignal_hip_$v.so; fi
  echo -n "variant '$v': "; KT_LINES=40 bash tools/exp/kt_ops.sh shen | grep "k_sc_gradient" | cut -c60-100
done
