python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/g6_gpu_tests_full.log; grep -E "passed|failed|error" gpurun_out/g6_gpu_tests_full.log | tail -3 > gpurun_out/g6_gpu_tests.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample --no-secondary"
COLDDIFF_PRECISION=bf16 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf -- $B > $O/trace_bf16.log 2>&1
python tools/prof_summary.py /tmp/prof_bf $O/kernel_trace_bf16.md $O/kernel_trace_bf16.json > /dev/null 2>&1
