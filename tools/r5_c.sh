cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 300 python tools/copy_sites.py > $O/copy_sites.txt 2>&1; tail -45 $O/copy_sites.txt
export CDF_BENCH_NOTIMER=1
B2="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sample --no-secondary"
rm -rf /tmp/pmcFb /tmp/pmcWb
COLDDIFF_PRECISION=bf16 timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcFb -- $B2 > $O/pmcFb.log 2>&1
COLDDIFF_PRECISION=bf16 timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmcWb -- $B2 > $O/pmcWb.log 2>&1
python tools/pmc_traffic.py /tmp/pmcFb /tmp/pmcWb $O/pmc_traffic_bf16.json $O/pmc_traffic_bf16.md; tail -3 $O/pmc_traffic_bf16.md
