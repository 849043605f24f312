cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-secondary"
(timeout 300 python -m pytest tests/test_bf16_storage.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/tests.log 2>&1
for g in 0 1 0 1; do
  CDF_PRE_GRAD=$g $B 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('x3 pre_grad $g', d['value'], 'img/s', d['ms_per_step'], 'ms')" >> $O/ab.txt
done
for g in 0 1 0 1; do
  COLDDIFF_PRECISION=bf16 CDF_PRE_GRAD=$g $B 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bf16 pre_grad $g', d['value'], 'img/s', d['ms_per_step'], 'ms')" >> $O/ab.txt
done
cat $O/ab.txt; cat $O/tests.log
