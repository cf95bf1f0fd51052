#!/bin/bash
# Round-end evidence, part 3 (runs ON THE GPU BOX via gpurun; the round's final build): GPU tests, smoke, every bench line,
# the training step's timers / timeline / kernel stats, the loss and head-gradient kernel timers.  The PMC traffic passes of
# part 1 stay valid while profiles/r03/traffic.json's source hashes match (bench.py checks them).
set -x
O=gpurun_out/r03ev3; mkdir -p $O
python -m pytest tests -m gpu -q > $O/b_gpu_tests.log 2>&1; tail -3 $O/b_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/b_smoke.log 2>&1; tail -1 $O/b_smoke.log
python bench.py > $O/i_bench_default_final.json 2> $O/i_bench_default_final.err
python bench.py --config c3 --steps 30 --warmup 5 > $O/e_bench_c3.json 2>/dev/null
python bench.py --config c4 --steps 30 --warmup 5 > $O/e_bench_c4.json 2>/dev/null
python bench.py --config c5 --steps 3 --warmup 1 > $O/e_bench_c5.json 2>/dev/null
python scripts/bench_full_train.py 4096 fp16x3 > $O/e_bench_full_train.json 2>/dev/null
python scripts/diag/graph_step_time.py > $O/d_graph_step_time.txt 2>&1
(python scripts/diag/geo_fuse_time.py; python scripts/diag/wgrad_time.py 4096; python scripts/diag/wgrad_time.py 8192) > $O/k_loss_and_head_gradient_kernels.txt 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/c3trace -o t --output-format csv -- python $R/scripts/diag/graph_step_time.py 1 1 0 > $R/$O/c3trace.log 2>&1
cd $R
f=$(find $O/c3trace -name "*kernel_trace.csv" | head -1)
python scripts/diag/step_timeline.py $f 20 > $O/h_c3_step_timeline.txt
cp $(find $O/c3trace -name "*kernel_stats.csv" | head -1) $O/h_c3_step_kernel_stats.csv
rm -rf $O/c3trace
tail -c 700 $O/i_bench_default_final.json
