run() { GA_B=$3 GA_SHAPE=$1 GA_EPI=$2 GA_VARIANTS="halo=47;halo=111" python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-150; }
for e in plain gelu res mulg; do run 128-64-128 $e 64; done
for e in gelu res mulg; do run 128-128-64 $e 64; run 128-256-64 $e 64; run 64-64-128 $e 64; done
for e in gelu res; do run 128-64-128 $e 16; run 128-128-64 $e 16; run 128-256-64 $e 16; run 64-128-128 $e 16; done
