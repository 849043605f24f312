# Round 5 evidence in one call (gpurun -- 'CDF_GIT_HEAD=<sha> bash tools/r5_final.sh'): the whole hardware test-suite with its printed error
# values, the artifacts of tools/collect_artifacts.sh (kernel traces + PMC traffic in both arithmetic modes, sampler trace), the bf16
# storage A/B, and the bench line.  Copy gpurun_out/final/* to profiles/round5_* afterwards.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -120) > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
bash tools/collect_artifacts.sh > $O/collect.log 2>&1; tail -12 $O/collect.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-secondary"
for s in 0 1 0 1; do
  COLDDIFF_PRECISION=bf16 COLDDIFF_BF16_STORAGE=$s $B 2>> $O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bf16 mode, COLDDIFF_BF16_STORAGE=$s:', d['value'], 'img/s', d['ms_per_step'], 'ms / step')" >> $O/bf16_storage_ab.txt
done
cat $O/bf16_storage_ab.txt
