# usage (on a GPU box): bash tools/exp/var_cc.sh <variant>... — the hysteresis kernels inside Image.shenCastan (4096^2 noise) for the product build and for
# variants of edges.hip (tools/build_variant.sh <name> edges.hip -D<name>)
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -n "$v" ]; then export ZIGNAL_HIP_LIBRARY=$GRAFT_REPO_ROOT/zignal_amd/variants/libzignal_hip_$v.so; fi
  echo "variant '$v': "; KT_LINES=40 bash tools/exp/kt_ops.sh shen | grep "k_cc_" | cut -c1-30,60-100
done
