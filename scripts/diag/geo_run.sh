R=$PWD
for i in 1 2 3; do
  (cd $R/exp/old_tree && echo "old (row-major):" && python scripts/diag/lp_save_time.py 2>&1 | grep -E "plain|save")
  (cd $R && echo "new (tile-major):" && python scripts/diag/lp_save_time.py 2>&1 | grep -E "plain|save")
done
(cd $R/exp/old_tree && echo "old:" && python scripts/diag/wgrad_time.py 4096 2>&1 | grep "S=" && python scripts/diag/graph_step_time.py 1 1 0 2>&1 | tail -1)
(cd $R && echo "new:" && python scripts/diag/wgrad_time.py 4096 2>&1 | grep "S=" && python scripts/diag/graph_step_time.py 1 1 0 2>&1 | tail -1)
(cd $R/exp/old_tree && echo "old:" && python scripts/diag/graph_step_time.py 1 1 0 2>&1 | tail -1)
(cd $R && echo "new:" && python scripts/diag/graph_step_time.py 1 1 0 2>&1 | tail -1)
