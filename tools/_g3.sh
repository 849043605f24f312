python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sample > gpurun_out/g3_bench_new.json 2> gpurun_out/g3_bench_new.err
COLDDIFF_LIB=tools/_ablate/ab/lib_prev.so python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sample > gpurun_out/g3_bench_prev.json 2> gpurun_out/g3_bench_prev.err
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sample > gpurun_out/g3_bench_new2.json 2>> gpurun_out/g3_bench_new.err
python -m pytest tests/test_kernels.py tests/test_gpu_parity2.py tests/test_bf16_storage.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/g3_tests.log
