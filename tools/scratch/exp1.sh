# A/B: resident row-halo / halo kernels (1 block of 8 waves per CU) vs the generic pre-split kernel (2 blocks of 4 waves per CU) on the short-K layers
export KB_F32=0 KB_SP=0 KB_SPW=0 KB_ITERS=20
for shp in 64-128-128-3 128-64-128-3 128-256-64-3 256-128-64-3 64-64-128-3; do
  echo "== $shp default"; KB_SHAPES=$shp python tools/convbench.py 2>&1 | grep -E "^spx "\|"^spxG"
  echo "== $shp generic 128x128 (2 blocks/CU)"; KB_TILE=128x128 KB_SHAPES=$shp python tools/convbench.py 2>&1 | grep -E "^spx "\|"^spxG"
  echo "== $shp generic 128x64"; KB_TILE=128x64 KB_SHAPES=$shp python tools/convbench.py 2>&1 | grep -E "^spx "\|"^spxG"
done
