# batched slab reductions on / off, same call: the 32 x 32 configurations and the main step
for v in 1 0 1 0; do CDF_BATCH_REDUCE=$v python tools/cfgbench.py 1 2>&1 | tail -1 | cut -c1-200 | sed "s/^/batch=$v /"; done
for v in 1 0; do CDF_BATCH_REDUCE=$v python tools/cfgbench.py 2 2>&1 | tail -1 | cut -c1-200 | sed "s/^/batch=$v /"; done
for v in 1 0 1; do CDF_BATCH_REDUCE=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sample --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch=$v', d['value'], d['ms_per_step'], d.get('bf16_mode',{}).get('value'))"; done
