python -m pytest tests/test_gpu_losses.py tests/test_gpu_configs.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -8
python scripts/diag/graph_step_time.py 2>&1 | tail -6
