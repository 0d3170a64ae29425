# same-box A/B of two builds of the library: bash tools/diag/ab_libs.sh <libA> <libB>   (paths relative to the repo root)
for r in 1 2; do
for lib in "$@"; do
echo "== $lib run $r"
MI_ILQR_LIB=$PWD/$lib python tools/run_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  %-50s %12.0f it/s' % (d['config'][:50], d['iterations_per_s']))
"
done; done
for lib in "$@"; do echo "== $lib"; MI_ILQR_LIB=$PWD/$lib python tools/cyc_large.py 2>/dev/null | grep "B=64\|B=8 "; MI_ILQR_LIB=$PWD/$lib MI_CYC_ARMS_ONLY=1 python tools/cyc_mid_models.py 2>/dev/null | grep -v "B     1 "; done
