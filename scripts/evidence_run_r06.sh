#!/bin/bash
# Round-6 evidence (runs ON THE GPU BOX via gpurun, on the round's final build).  Stages, each one gpurun call:
#   evidence_run_r06.sh kernels   PMC traffic passes (-> gpurun_out/traffic_r06; scripts/make_traffic_json.py turns them into
#                                 profiles/r06/traffic.json in the build container), phase tables of mlp_lp16_kernel and
#                                 mlp_x316_kernel, the two split-fp16 forward kernels side by side, the 16-bit kernels' times
#   evidence_run_r06.sh bench     GPU tests, smoke, every bench line (default with variants + CPU baseline, c1, c3, c4, c5, c5 with a
#                                 split-fp16 coarse pass, full training), kernel stats of the default bench command, step timelines
set -x
STAGE=${1:-bench}
O=gpurun_out/r06ev; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
if [ $STAGE = kernels ]; then
    bash scripts/profile_traffic.sh r06 > $O/traffic.log 2>&1
    python scripts/phase_profile_lp.py 2 fp16 > $O/g_phase_lp16_semcoord_fp16.txt 2>&1
    python scripts/phase_profile_x316.py 0 > $O/g_phase_x316_nosem.txt 2>&1
    python scripts/phase_profile_x316.py 2 > $O/g_phase_x316_semcoord.txt 2>&1
    python scripts/diag/x3_ab.py > $O/g_x3_kernels_ab.txt 2>&1
    (python scripts/diag/lp_time.py 4096 5; python scripts/diag/lp_time.py 65536 2 3; python scripts/diag/lp_save_time.py) > $O/g_lp_times.txt 2>&1
    tail -4 $O/g_x3_kernels_ab.txt
else
    python -m pytest tests -m gpu -q > $O/h_gpu_tests.log 2>&1; tail -3 $O/h_gpu_tests.log
    python -c "import __graft_entry__ as g; g.smoke()" > $O/h_smoke.log 2>&1; tail -1 $O/h_smoke.log
    python bench.py > $O/i_bench_default.out 2> $O/i_bench_default.err
    python bench.py --config c1 > $O/i_bench_c1.out 2>/dev/null
    python bench.py --config c3 --steps 30 --warmup 5 > $O/i_bench_c3.out 2>/dev/null
    python bench.py --config c4 --steps 30 --warmup 5 > $O/i_bench_c4.out 2>/dev/null
    python bench.py --config c5 --steps 3 --warmup 1 > $O/i_bench_c5.out 2>/dev/null
    python bench.py --config c5 --coarse-precision fp16x3 --steps 3 --warmup 1 > $O/i_bench_c5_coarse_fp16x3.out 2>/dev/null
    python scripts/bench_full_train.py 4096 fp16x3 > $O/i_bench_full_train.json 2>/dev/null
    python scripts/diag/graph_step_time.py > $O/i_graph_step_time.txt 2>&1
    cd /tmp
    rocprofv3 --kernel-trace --stats -d $R/$O/deftrace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-variants > $R/$O/deftrace.log 2>&1
    rocprofv3 --kernel-trace --stats -d $R/$O/c3trace -o t --output-format csv -- python $R/scripts/diag/graph_step_time.py 1 1 0 > $R/$O/c3trace.log 2>&1
    rocprofv3 --kernel-trace --stats -d $R/$O/c5trace -o t --output-format csv -- python $R/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/c5trace.log 2>&1
    rocprofv3 --kernel-trace --stats -d $R/$O/c5mtrace -o t --output-format csv -- python $R/bench.py --config c5 --coarse-precision fp16x3 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/c5mtrace.log 2>&1
    cd $R
    cp $(find $O/deftrace -name "*kernel_stats.csv" | head -1) $O/j_kernel_stats_default_bench_no_variants.csv
    python scripts/diag/step_timeline.py $(find $O/c3trace -name "*kernel_trace.csv" | head -1) 20 > $O/j_c3_step_timeline.txt
    cp $(find $O/c3trace -name "*kernel_stats.csv" | head -1) $O/j_c3_step_kernel_stats.csv
    cp $(find $O/c5trace -name "*kernel_stats.csv" | head -1) $O/j_c5_image_kernel_stats.csv
    cp $(find $O/c5mtrace -name "*kernel_stats.csv" | head -1) $O/j_c5_coarse_fp16x3_image_kernel_stats.csv
    grep '^{"metric"' $O/deftrace.log | tail -1 > $O/j_default_bench_line_under_rocprof.json
    rm -rf $O/deftrace $O/c3trace $O/c5trace $O/c5mtrace
    tail -c 600 $O/i_bench_default.out
fi
