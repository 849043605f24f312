cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
export KB_B=64 KB_F32=0 KB_SP=0 KB_SPW=0 KB_WGRAD=0 KB_ITERS=10 KB_HALO=0 KB_TILE=256x128
for dp in 1 2 1 2; do
  echo "== generic 256x128 tile, dephase=$dp (2 = four waves of 128x64)" >> $O/wide.txt
  KB_DEPHASE=$dp python tools/convbench.py 2>/dev/null | grep "^spx \|^spxG" >> $O/wide.txt
done
cat $O/wide.txt
