cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5bench; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
COLDDIFF_DIST_BACKEND=gloo COLDDIFF_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_2rank_gloo_one_gpu.json 2> $O/bench_2rank.err; cut -c1-200 $O/bench_2rank_gloo_one_gpu.json
