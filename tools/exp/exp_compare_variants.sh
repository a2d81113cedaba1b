# what bounds the comparison's kernels at configs[3]: experiment builds (tools/exp/build_variant.sh) with parts of the work left out
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-10000}
which=${2:-all}
echo "== product"; python3 $R/tools/c4_compare.py $N 10
echo "== product, has-a-list bits forced on / off"; SPSP_DEBUG_MULTI=1 python3 $R/tools/c4_compare.py $N 10; SPSP_DEBUG_MULTI=0 python3 $R/tools/c4_compare.py $N 10
if [ $which = all ] || [ $which = scatter ]; then
for v in 1 2 3 7; do
  echo "== scatter variant $v (1: no global reservation atomics, 2: no record stores, 4: no where stores), group + row sums skipped"
  SPSP_LIB=$R/tools/exp/ab/libspsp_exp$v.so SPSP_DEBUG_SKIP_STAGES=6 python3 $R/tools/c4_compare.py $N 10
done
echo "== product, group + row sums skipped"; SPSP_DEBUG_SKIP_STAGES=6 python3 $R/tools/c4_compare.py $N 10
fi
if [ $which = all ] || [ $which = acc ]; then
for v in 8 16 32 64; do
  echo "== row-sum variant $v (8: lists read but no LDS adds, 16: no list reads, 32: no reference reads, 64: LDS adds spread over 4096 words)"
  SPSP_LIB=$R/tools/exp/ab/libspsp_exp$v.so python3 $R/tools/c4_compare.py $N 10
done
fi
