python -m pytest tests/test_gpu_backward.py -x -q 2>&1 | tail -3
python scripts/diag/wgrad_time.py 4096 2>&1 | tail -2
python scripts/diag/wgrad_time.py 8192 2>&1 | tail -2
python scripts/diag/wg_prof.py 2>&1 | tail -8
