# same-call A/B: specialised epilogue (this tree) vs round-5 library (tools/_ablate/ab/lib_prev.so)
run() { for lib in "" tools/_ablate/ab/lib_prev.so; do COLDDIFF_LIB=$lib GA_B=64 GA_SHAPE=$1 GA_EPI=$2 GA_VARIANTS="halo=47" python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids | sed "s|^|${lib:-new} |"; done; }
for e in plain gelu gelun res mulg acc; do run 64-128-128 $e; done
for e in plain gelu res mulg; do run 128-64-128 $e; done
for e in gelu res; do run 128-128-128 $e; done
for e in gelu res mulg; do run 128-256-64 $e; run 256-128-64 $e; done
for e in gelu res; do run 256-512-32 $e; run 512-256-32 $e; run 512-1024-16 $e; done
