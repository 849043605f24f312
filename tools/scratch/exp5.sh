echo "== new (4x4 tiles)"; python tools/blurbench.py 2>&1 | grep -v amdgpu.ids
echo "== previous"; COLDDIFF_LIB=tools/_ablate/ab/lib_prev.so python tools/blurbench.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_kernels.py tests/test_gpu_fullsize.py tests/test_modules.py -m gpu -x -q -k "blur or deblur" 2>&1 | tail -3
