cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 1 2; do
  rm -rf /tmp/tr$B
  rocprofv3 --kernel-trace -d /tmp/tr$B -o t --output-format csv -- python $R/scripts/diag/graph_step_time.py $B 1 0 > /tmp/tr$B.log 2>&1
  f=$(find /tmp/tr$B -name "*kernel_trace.csv" | head -1)
  python $R/scripts/diag/step_timeline.py $f 20 > $R/gpurun_out/timeline_B$B.txt
  tail -2 /tmp/tr$B.log
done
python $R/scripts/diag/graph_step_time.py
