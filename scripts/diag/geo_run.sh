python -m pytest tests/test_gpu_losses.py -x -q 2>&1 | tail -3
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/tl; cd /tmp
for m in "1 1 0"; do
  tag=$(echo $m | tr -d ' ')
  rocprofv3 --kernel-trace -d $R/gpurun_out/tl/$tag -o t --output-format csv -- python $R/scripts/diag/graph_step_time.py $m > $R/gpurun_out/tl/log_$tag.txt 2>&1
  f=$(find $R/gpurun_out/tl/$tag -name "*kernel_trace.csv" | head -1)
  python $R/scripts/diag/step_timeline.py $f 20 > $R/gpurun_out/tl/timeline_$tag.txt
  rm -f $f
done
grep -E "wall per step|app_|pair_cols_kernel<false|pair_rows_kernel<false" $R/gpurun_out/tl/timeline_110.txt
