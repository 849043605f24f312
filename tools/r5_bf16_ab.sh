# Round 5: the bf16 arithmetic mode with fp32 tensors (rounds 2-4) against bf16 activation storage, same box, same call; plus the headline
# (bf16x3) line as a regression check and kernel traces of both bf16 forms.   gpurun -- 'bash tools/r5_bf16_ab.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-secondary"
(timeout 600 python -m pytest tests/test_bf16_storage.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -15) > $O/tests_bf16_storage.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_parity2.py -m gpu -q -s -p no:cacheprovider -k "bf16 or bench_shape or precision_modes" 2>&1 | tail -15) > $O/tests_parity2.log 2>&1
$B > $O/bench_x3.json 2> $O/bench_x3.err
for s in 0 1 0 1; do
  COLDDIFF_PRECISION=bf16 COLDDIFF_BF16_STORAGE=$s $B 2>> $O/bench_bf16.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('storage $s', d['value'], 'img/s', d['ms_per_step'], 'ms')" >> $O/bf16_ab.txt
done
cat $O/bf16_ab.txt
rm -rf /tmp/prof_bf0 /tmp/prof_bf1
B4="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample --no-secondary"
COLDDIFF_PRECISION=bf16 COLDDIFF_BF16_STORAGE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf1 -- $B4 > $O/trace_bf1.log 2>&1
python tools/prof_summary.py /tmp/prof_bf1 $O/kernel_trace_bf16_storage.md $O/kernel_trace_bf16_storage.json > /dev/null 2>&1; tail -1 $O/kernel_trace_bf16_storage.md
tail -4 $O/tests_bf16_storage.log; tail -6 $O/tests_parity2.log
python -c "import json; d=json.load(open('$O/bench_x3.json')); print('x3', d['value'], d['ms_per_step'])"
