cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 400 python bench.py > $O/bench_final.json 2> $O/bench_err.log; tail -c 600 $O/bench_final.json | head -c 300; echo
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sample"
rm -rf /tmp/prof_kt /tmp/pmcF /tmp/pmcW
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $B > $O/prof_v23.log 2>&1
python tools/prof_summary.py /tmp/prof_kt > $O/prof_v23_summary.md 2>&1; head -12 $O/prof_v23_summary.md
export CDF_BENCH_NOTIMER=1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcF -- $B > $O/pmcF.log 2>&1; tail -2 $O/pmcF.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmcW -- $B > $O/pmcW.log 2>&1; tail -2 $O/pmcW.log
python tools/pmc_traffic.py /tmp/pmcF /tmp/pmcW $O/pmc_traffic.json $O/pmc_traffic.md; head -14 $O/pmc_traffic.md; tail -4 $O/pmc_traffic.md
