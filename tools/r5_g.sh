cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-secondary"
COLDDIFF_PRECISION=bf16 CDF_BENCH_SHAPES=1 CDF_BENCH_SHAPES_N=60 $B > $O/bench_bf16.json 2> $O/bench_bf16.err
grep "ms/step" $O/bench_bf16.err > $O/shapes_bf16.txt
rm -rf /tmp/prof_bf1
B4="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample --no-secondary"
COLDDIFF_PRECISION=bf16 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf1 -- $B4 > $O/trace_bf1.log 2>&1
python tools/prof_summary.py /tmp/prof_bf1 $O/kernel_trace_bf16.md $O/kernel_trace_bf16.json > /dev/null 2>&1; tail -1 $O/kernel_trace_bf16.md
for c in 1 2 4; do
  rm -rf /tmp/prof_c$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c$c -- python tools/cfgbench.py $c > $O/cfg$c.log 2>&1
  python tools/prof_summary.py /tmp/prof_c$c $O/cfg${c}_kernel_trace.md > /dev/null 2>&1; grep img_per_s $O/cfg$c.log; tail -1 $O/cfg${c}_kernel_trace.md
done
