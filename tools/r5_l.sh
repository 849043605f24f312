cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
export GA_B=64 GA_ROUNDS=5 GA_VARIANTS="rowhalo_stream=1;rowhalo_stream=2"
for sh in 64-128-128 128-128-128 128-64-128; do
  for epi in plain gelu res; do
    GA_SHAPE=$sh GA_EPI=$epi timeout 120 python tools/gemm_ab.py 2>&1 | grep "epi=" >> $O/roles_ab.txt
  done
done
GA_SINGLE=1 true
cat $O/roles_ab.txt
