# usage (on a GPU box): bash tools/exp/var_cc.sh — the hysteresis kernels inside Image.shenCastan / canny (4096^2 noise) with pieces of k_cc_tile compiled out
cd $GRAFT_REPO_ROOT
for v in "" CC_NO_UNITE CC_NO_FLAT; do
  if [ -n "$v" ]; then export ZIGNAL_HIP_LIBRARY=$GRAFT_REPO_ROOT/zignal_amd/variants/libzignal_hip_$v.so; fi
  echo "variant '$v': "; KT_LINES=40 bash tools/exp/kt_ops.sh shen | grep "k_cc_" | cut -c1-30,60-100
done
