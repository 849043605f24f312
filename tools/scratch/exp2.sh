echo "== new (4x4 thread tiles, 128 threads)"; python tools/dwtime.py
echo "== previous (4x2, 256 threads)"; COLDDIFF_LIB=tools/_ablate/ab/lib_prev.so python tools/dwtime.py
python -m pytest tests/test_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "dwconv7 or cfg5" 2>&1 | tail -3
