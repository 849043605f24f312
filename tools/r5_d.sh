cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-secondary"
CDF_BENCH_SHAPES=1 CDF_BENCH_SHAPES_N=80 $B > $O/bench_x3.json 2> $O/bench_x3.err
grep "ms/step" $O/bench_x3.err > $O/shapes_x3.txt
for s in 0 1; do
  COLDDIFF_PRECISION=bf16 COLDDIFF_BF16_STORAGE=$s $B 2>> $O/bench_bf16.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('storage $s', d['value'], 'img/s', d['ms_per_step'], 'ms')" >> $O/bf16_ab.txt
done
cat $O/bf16_ab.txt
python -c "import json; d=json.load(open('$O/bench_x3.json')); print('x3', d['value'], d['ms_per_step'])"
