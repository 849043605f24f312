run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo "== default"; run
echo "== SMALL_N64=0"; COLDDIFF_SPX_SMALL_N64=0 run
echo "== HALO_BM=128"; COLDDIFF_SPX_HALO_BM=128 run
echo "== default"; run
echo "== MAX_BM=128"; COLDDIFF_SPX_MAX_BM=128 run
echo "== DEPHASE=0"; COLDDIFF_SPX_DEPHASE=0 run
echo "== WGRAD_STACK=0"; COLDDIFF_WGRAD_STACK=0 run
echo "== ROWHALO_STREAM=0"; COLDDIFF_ROWHALO_STREAM=0 run
echo "== default"; run
echo "== HALO=111 (row-halo stream wherever it applies)"; COLDDIFF_SPX_HALO=111 run
echo "== batch 128 accum 1 (what a 4-micro-batch step would look like)"; python bench.py --steps 6 --warmup 2 --batch 128 --accum 1 --no-cpu-baseline --no-sample --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
