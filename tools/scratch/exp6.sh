cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_smp; timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_smp -- python tools/sample_prof.py --steps 20 > gpurun_out/r4_sample_trace.log 2>&1
python tools/prof_summary.py /tmp/prof_smp gpurun_out/r4_sample_kernel_trace.md > /dev/null 2>&1; grep "ms per" gpurun_out/r4_sample_trace.log; head -45 gpurun_out/r4_sample_kernel_trace.md | cut -c1-120; tail -1 gpurun_out/r4_sample_kernel_trace.md
