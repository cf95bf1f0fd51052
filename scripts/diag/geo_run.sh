python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python scripts/diag/graph_step_time.py 2>&1 | tail -6
