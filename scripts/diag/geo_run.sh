python -m pytest tests/test_gpu_losses.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
python scripts/bench_losses.py 2>&1 | tail -12
python scripts/diag/graph_step_time.py 2>&1 | tail -6
