cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
(timeout 600 python -m pytest tests/test_kernels.py tests/test_bf16_storage.py -m gpu -q -p no:cacheprovider -k "pack or bf16 or dwconv" 2>&1 | tail -3) > $O/tests.log 2>&1; cat $O/tests.log
rm -rf /tmp/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample --no-secondary > $O/trace.log 2>&1
python tools/prof_summary.py /tmp/prof_kt $O/kernel_trace_x3.md $O/kernel_trace.json > /dev/null 2>&1; grep "pack_many\|adam_kernel" $O/kernel_trace_x3.md; tail -1 $O/kernel_trace_x3.md
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-secondary"
$B 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('x3', d['value'], 'img/s', d['ms_per_step'], 'ms')"
COLDDIFF_PRECISION=bf16 $B 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bf16', d['value'], 'img/s', d['ms_per_step'], 'ms')"
python tools/bf16_grad_errors.py 2>&1 | tail -12 | head -4
python tools/cfgbench.py 1 2 2>&1 | grep img_per_s
