#!/bin/bash
# Round-5 evidence (runs ON THE GPU BOX via gpurun, on the round's final build).  Stages, each one gpurun call:
#   evidence_run_r05.sh kernels   PMC traffic passes (-> gpurun_out/traffic_r05final; scripts/make_traffic_json.py turns them into
#                                 profiles/r05/traffic.json HERE), SQ / LDS PMC passes of the kernel driver, phase table of mlp_lp16_kernel,
#                                 the 16-bit kernels timed side by side, loss / head-gradient kernel timers, generic kernels' phase profiles
#   evidence_run_r05.sh bench     GPU tests, smoke, every bench line (default with variants + CPU baseline, c1, c3, c4, c5, full training),
#                                 generic timings, kernel stats of the default bench command, step timelines, pose-step trace
set -x
STAGE=${1:-bench}
O=gpurun_out/r05ev; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
if [ $STAGE = kernels ]; then
    bash scripts/profile_traffic.sh r05final > $O/traffic.log 2>&1
    PROFILE_CMD="python $R/scripts/diag/traffic_driver.py" bash scripts/profile_gpu.sh r05kernels > $O/profile_kernels.log 2>&1
    cp gpurun_out/prof_r05kernels/summary.txt $O/f_pmc_kernel_driver_summary.txt
    python scripts/phase_profile_lp.py 2 fp16 > $O/c_phase_lp16_semcoord_fp16.txt 2>&1
    (python scripts/diag/lp_time.py 4096 5; python scripts/diag/lp_time.py 65536 2 3; python scripts/diag/lp_save_time.py) > $O/d_lp_times.txt 2>&1
    (python scripts/diag/geo_fuse_time.py; python scripts/diag/wgrad_time.py 4096; python scripts/diag/wgrad_time.py 8192) > $O/k_loss_and_head_gradient_kernels.txt 2>&1
    python scripts/diag/composite_importance_time.py > $O/k2_composite_importance_time.txt 2>&1
    python scripts/diag/gen_prof.py 8x256m6 16x256 4x128 deepsem > $O/h_generic_forward_phase_profile.txt 2>&1
    python scripts/diag/gen_prof_bwd.py 8x256m6 6x96 > $O/i_generic_backward_phase_profile.txt 2>&1
    PROFILE_CMD="python $R/scripts/bench_generic_train.py 1024" bash scripts/profile_gpu.sh r05generic > $O/profile_generic.log 2>&1
    cp gpurun_out/prof_r05generic/summary.txt $O/o_pmc_generic_train_summary.txt
    cp $(find gpurun_out/prof_r05generic/trace -name "*kernel_stats.csv" | head -1) $O/n_generic_train_kernel_stats.csv
    rm -rf gpurun_out/prof_r05generic gpurun_out/prof_r05kernels/pmc_* gpurun_out/prof_r05kernels/trace
    tail -4 $O/d_lp_times.txt
else
    python -m pytest tests -m gpu -q > $O/b_gpu_tests.log 2>&1; tail -3 $O/b_gpu_tests.log
    python -c "import __graft_entry__ as g; g.smoke()" > $O/b_smoke.log 2>&1; tail -1 $O/b_smoke.log
    python bench.py > $O/i_bench_default.out 2> $O/i_bench_default.err
    python bench.py --config c1 > $O/e_bench_c1.out 2>/dev/null
    python bench.py --config c3 --steps 30 --warmup 5 > $O/e_bench_c3.out 2>/dev/null
    python bench.py --config c4 --steps 30 --warmup 5 > $O/e_bench_c4.out 2>/dev/null
    python bench.py --config c5 --steps 3 --warmup 1 > $O/e_bench_c5.out 2>/dev/null
    python scripts/bench_full_train.py 4096 fp16x3 > $O/e_bench_full_train.json 2>/dev/null
    python scripts/diag/generic_time.py > $O/l_generic_kernel_times.txt 2>&1
    python scripts/bench_generic_train.py > $O/m_generic_train.txt 2>&1
    python scripts/diag/raygrad_time.py > $O/p_raygrad_time.txt 2>&1
    python scripts/diag/generic_head_only_time.py > $O/r_generic_head_only.txt 2>&1
    python scripts/diag/graph_step_time.py > $O/d_graph_step_time.txt 2>&1
    python scripts/soak_determinism.py 100 2048 > $O/j_soak_determinism.txt 2>&1
    cd /tmp
    rocprofv3 --kernel-trace --stats -d $R/$O/deftrace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-variants > $R/$O/deftrace.log 2>&1
    rocprofv3 --kernel-trace --stats -d $R/$O/c3trace -o t --output-format csv -- python $R/scripts/diag/graph_step_time.py 1 1 0 > $R/$O/c3trace.log 2>&1
    rocprofv3 --kernel-trace --stats -d $R/$O/c4trace -o t --output-format csv -- python $R/scripts/diag/graph_step_time.py 2 1 0 > $R/$O/c4trace.log 2>&1
    rocprofv3 --kernel-trace --stats -d $R/$O/c5trace -o t --output-format csv -- python $R/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/c5trace.log 2>&1
    rocprofv3 --kernel-trace --stats -d $R/$O/posetrace -o t --output-format csv -- python $R/scripts/diag/raygrad_time.py > $R/$O/posetrace.log 2>&1
    cd $R
    cp $(find $O/deftrace -name "*kernel_stats.csv" | head -1) $O/f_kernel_stats_default_bench_no_variants.csv
    python scripts/diag/step_timeline.py $(find $O/c3trace -name "*kernel_trace.csv" | head -1) 20 > $O/h_c3_step_timeline.txt
    python scripts/diag/step_timeline.py $(find $O/c4trace -name "*kernel_trace.csv" | head -1) 20 > $O/h_c4_step_timeline.txt
    cp $(find $O/c3trace -name "*kernel_stats.csv" | head -1) $O/h_c3_step_kernel_stats.csv
    cp $(find $O/c5trace -name "*kernel_stats.csv" | head -1) $O/h_c5_image_kernel_stats.csv
    cp $(find $O/posetrace -name "*kernel_stats.csv" | head -1) $O/p_pose_step_kernel_stats.csv
    grep '^{"metric"' $O/deftrace.log | tail -1 > $O/f_default_bench_line_under_rocprof.json
    rm -rf $O/deftrace $O/c3trace $O/c4trace $O/c5trace $O/posetrace
    tail -c 600 $O/i_bench_default.out
fi
