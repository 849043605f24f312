cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
python tools/bf16_grad_errors.py > $O/bf16_grad_errors.txt 2>&1; tail -13 $O/bf16_grad_errors.txt
COLDDIFF_BF16_STORAGE=0 python tools/bf16_grad_errors.py > $O/bf16_grad_errors_storage0.txt 2>&1; tail -6 $O/bf16_grad_errors_storage0.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
