# tile geometry sweep at configs[3]: sketches per block x entries per tile
R=${GRAFT_REPO_ROOT:-$(pwd)}
for sk in 16 32 64; do for tg in 3072 3712 4096; do
  echo "== TILE_SK=$sk target=$tg"; SPSP_DEBUG_TILE_SK=$sk SPSP_DEBUG_TILES=$tg python3 $R/tools/c4_compare.py ${1:-10000} 10 2>&1 | grep -v amdgpu
done; done
