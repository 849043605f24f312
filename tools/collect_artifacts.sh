# Round artifacts on the GPU box (run through gpurun from the repo root, CDF_GIT_HEAD=<git rev-parse HEAD> in the environment: the box has
# no .git): kernel traces of the train step in both arithmetic modes, the two PMC passes in both modes, the sampler trace.  Copy the
# summaries from gpurun_out/final into profiles/ afterwards (profiles/round<N>_*), then run bench.py (its hbm_bound_kernel_classes,
# roofline.traffic and measured_hbm_bytes_step figures read the newest profiles/round<N>_pmc_traffic*.json and _kernel_trace.json and
# quote them only when their csrc digest equals that of the running sources).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample --no-secondary"
rm -rf /tmp/prof_kt /tmp/prof_bf /tmp/pmcF /tmp/pmcW /tmp/pmcFb /tmp/pmcWb
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $B > $O/trace_x3.log 2>&1
python tools/prof_summary.py /tmp/prof_kt $O/kernel_trace_x3.md $O/kernel_trace.json > /dev/null 2>&1; head -8 $O/kernel_trace_x3.md; tail -1 $O/kernel_trace_x3.md
COLDDIFF_PRECISION=bf16 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf -- $B > $O/trace_bf16.log 2>&1
python tools/prof_summary.py /tmp/prof_bf $O/kernel_trace_bf16.md $O/kernel_trace_bf16.json > /dev/null 2>&1; tail -1 $O/kernel_trace_bf16.md
export CDF_BENCH_NOTIMER=1
B2="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sample --no-secondary"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcF -- $B2 > $O/pmcF.log 2>&1; tail -1 $O/pmcF.log | cut -c1-200
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmcW -- $B2 > $O/pmcW.log 2>&1; tail -1 $O/pmcW.log | cut -c1-200
python tools/pmc_traffic.py /tmp/pmcF /tmp/pmcW $O/pmc_traffic.json $O/pmc_traffic.md; head -6 $O/pmc_traffic.md; tail -3 $O/pmc_traffic.md
COLDDIFF_PRECISION=bf16 timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcFb -- $B2 > $O/pmcFb.log 2>&1
COLDDIFF_PRECISION=bf16 timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmcWb -- $B2 > $O/pmcWb.log 2>&1
python tools/pmc_traffic.py /tmp/pmcFb /tmp/pmcWb $O/pmc_traffic_bf16.json $O/pmc_traffic_bf16.md; tail -3 $O/pmc_traffic_bf16.md
unset CDF_BENCH_NOTIMER
# the sampler (200-step gen_sample at the sample batch): which kernels a reverse step spends its time in
rm -rf /tmp/prof_smp
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_smp -- python tools/sample_prof.py --steps 20 > $O/sample_trace.log 2>&1
python tools/prof_summary.py /tmp/prof_smp $O/sample_kernel_trace.md > /dev/null 2>&1; grep "ms per" $O/sample_trace.log; tail -1 $O/sample_kernel_trace.md
