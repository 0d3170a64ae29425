# placement orders of a cluster's workgroups, same box: bash tools/diag/xcd_orders.sh [lib]
lib=${1:-drake_ddp_amd/lib/libmi_ilqr.so}
for ord in 0 1 2; do
echo "== $lib MI_ILQR_CLUSTER_ORDER=$ord"
MI_ILQR_CLUSTER_ORDER=$ord MI_ILQR_LIB=$PWD/$lib python tools/cyc_large_models.py 2>/dev/null
MI_ILQR_CLUSTER=4 MI_ILQR_CLUSTER_ORDER=$ord MI_ILQR_LIB=$PWD/$lib python tools/cyc_large_models.py 2>/dev/null | head -1
MI_ILQR_CLUSTER_ORDER=$ord MI_ILQR_LIB=$PWD/$lib python tools/cyc_large_models.py 8 2>/dev/null | head -1
MI_ILQR_CLUSTER_ORDER=$ord MI_ILQR_LIB=$PWD/$lib MI_CYC_ARMS_ONLY=1 python tools/cyc_mid_models.py 2>/dev/null | grep -v "B     1 \|B  1024\|B   256"
done
