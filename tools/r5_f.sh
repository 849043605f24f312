cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-secondary"
for g in 0 1 0 1; do
  CDF_KV_PLANES=$g $B 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('x3 kv_planes $g', d['value'], 'img/s', d['ms_per_step'], 'ms')" >> $O/ab.txt
done
for g in 0 1 0 1; do
  COLDDIFF_PRECISION=bf16 CDF_KV_PLANES=$g $B 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bf16 kv_planes $g', d['value'], 'img/s', d['ms_per_step'], 'ms')" >> $O/ab.txt
done
cat $O/ab.txt
(timeout 600 python -m pytest tests/test_modules.py tests/test_gpu_parity2.py -m gpu -q -p no:cacheprovider -k "attention or bench_shape" 2>&1 | tail -3) > $O/tests.log 2>&1; cat $O/tests.log
