python -m pytest tests/test_dispatch_gpu.py -q -x 2>&1 | tail -30
echo ======
python -m pytest "tests/test_lazy_ln_gpu.py" -q -x -k "train_nodrop and b37" 2>&1 | tail -30
echo ======
BMNAS_FUSE_HEAD=0 python -m pytest tests/test_lazy_ln_gpu.py -q -x 2>&1 | tail -30
