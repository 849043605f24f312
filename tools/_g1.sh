set -x
python -m pytest tests/test_gpu_fullsize.py -x -q -k "cfg2" -s 2>&1 | tail -5 > gpurun_out/g1_cfg2.log
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sample > gpurun_out/g1_bench.json 2> gpurun_out/g1_bench.err
for e in gelu res plain; do GA_B=64 GA_SHAPE=64-128-128 GA_EPI=$e GA_VARIANTS="halo=47" python tools/gemm_ab.py; done > gpurun_out/g1_ab.txt 2>&1
GA_B=64 GA_SHAPE=128-64-128 GA_EPI=res GA_VARIANTS="halo=47" python tools/gemm_ab.py >> gpurun_out/g1_ab.txt 2>&1
GA_B=64 GA_SHAPE=128-128-64 GA_EPI=gelu GA_VARIANTS="halo=47" python tools/gemm_ab.py >> gpurun_out/g1_ab.txt 2>&1
GA_B=64 GA_SHAPE=256-128-64 GA_EPI=res GA_VARIANTS="halo=47" python tools/gemm_ab.py >> gpurun_out/g1_ab.txt 2>&1
