run() { for lib in "" tools/_ablate/ab/lib_SLACK.so tools/_ablate/ab/lib_STAG3.so tools/_ablate/ab/lib_STAG6.so tools/_ablate/ab/lib_STAG9.so tools/_ablate/ab/lib_BOTH.so; do COLDDIFF_LIB=$lib GA_B=64 GA_SHAPE=$1 GA_EPI=$2 GA_VARIANTS="halo=47" python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-120 | sed "s|^|$(basename ${lib:-base}) |"; done; }
for e in plain gelu res mulg; do run 64-128-128 $e; done
for e in gelu res; do run 128-128-128 $e; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $B > $O/trace_x3.log 2>&1
python tools/prof_summary.py /tmp/prof_kt $O/kernel_trace_x3.md $O/kernel_trace.json > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_smp -- python tools/sample_prof.py --steps 20 > $O/sample_trace.log 2>&1
python tools/prof_summary.py /tmp/prof_smp $O/sample_kernel_trace.md > /dev/null 2>&1; grep "ms per" $O/sample_trace.log
