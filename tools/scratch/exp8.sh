export COLDDIFF_DIST_BACKEND=gloo COLDDIFF_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/r4_bench_2rank_gloo.json 2> gpurun_out/r4_bench_2rank_gloo.err
echo rc=$?; tail -5 gpurun_out/r4_bench_2rank_gloo.err; python -c "
import json; d=json.load(open('gpurun_out/r4_bench_2rank_gloo.json')); print(d['value'], d['n_gpus'], d['ms_per_step'], d['config']['parallelism']); print(d.get('gradient_exchange'))"
