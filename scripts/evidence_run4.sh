#!/bin/bash
# Round-4 evidence (runs ON THE GPU BOX via gpurun, on the round's final build).  Two stages so that each fits one gpurun call:
#   evidence_run4.sh kernels   PMC traffic passes (-> gpurun_out/traffic_r04final: scripts/make_traffic_json.py turns them into
#                              profiles/r04/traffic.json HERE, before stage `bench`), SQ / LDS PMC passes of the kernel driver, phase
#                              tables of mlp_lp16_kernel, the three 16-bit kernels timed side by side, the MFMA-mix microbenchmark,
#                              loss / head-gradient kernel timers
#   evidence_run4.sh bench     GPU tests, smoke, every bench line (default with variants + CPU baseline, c1, c3, c4, c5, full
#                              training), kernel stats of the default bench command, the C3 step's timeline
set -x
STAGE=${1:-kernels}
O=gpurun_out/r04ev; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
if [ $STAGE = kernels ]; then
    bash scripts/profile_traffic.sh r04final > $O/traffic.log 2>&1
    PROFILE_CMD="python $R/scripts/diag/traffic_driver.py" bash scripts/profile_gpu.sh r04kernels > $O/profile_kernels.log 2>&1
    cp gpurun_out/prof_r04kernels/summary.txt $O/f_pmc_kernel_driver_summary.txt
    python scripts/phase_profile_lp.py 2 fp16 > $O/c_phase_lp16_semcoord_fp16.txt 2>&1
    python scripts/phase_profile_lp.py 2 bf16 > $O/c_phase_lp16_semcoord_bf16.txt 2>&1
    python scripts/phase_profile_lp.py 0 fp16 > $O/c_phase_lp16_nosem_fp16.txt 2>&1
    python scripts/phase_profile_lp.py 2 bf16 3 --save > $O/c_phase_lp16_semcoord_bf16_save.txt 2>&1
    (python scripts/diag/lp_time.py 4096 5; python scripts/diag/lp_time.py 65536 2 3; python scripts/diag/lp_save_time.py) > $O/d_lp_times.txt 2>&1
    (hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix) 2>&1 | grep -v "warning\|\^\|^ *[0-9]* |" > $O/a_mfma_mix.txt
    (python scripts/diag/geo_fuse_time.py; python scripts/diag/wgrad_time.py 4096; python scripts/diag/wgrad_time.py 8192) > $O/k_loss_and_head_gradient_kernels.txt 2>&1
    tail -4 $O/d_lp_times.txt
else
    python -m pytest tests -m gpu -q > $O/b_gpu_tests.log 2>&1; tail -3 $O/b_gpu_tests.log
    python -c "import __graft_entry__ as g; g.smoke()" > $O/b_smoke.log 2>&1; tail -1 $O/b_smoke.log
    python bench.py > $O/i_bench_default.json 2> $O/i_bench_default.err
    python bench.py --config c1 > $O/e_bench_c1.json 2>/dev/null
    python bench.py --config c3 --steps 30 --warmup 5 > $O/e_bench_c3.json 2>/dev/null
    python bench.py --config c4 --steps 30 --warmup 5 > $O/e_bench_c4.json 2>/dev/null
    python bench.py --config c5 --steps 3 --warmup 1 > $O/e_bench_c5.json 2>/dev/null
    python scripts/bench_full_train.py 4096 fp16x3 > $O/e_bench_full_train.json 2>/dev/null
    python scripts/bench_full_train.py 4096 fp16x3 --compact > $O/e_bench_full_train_compact.json 2>/dev/null
    python scripts/diag/generic_time.py > $O/l_generic_kernel_times.txt 2>&1
    python scripts/diag/graph_step_time.py > $O/d_graph_step_time.txt 2>&1
    cd /tmp
    rocprofv3 --kernel-trace --stats -d $R/$O/deftrace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline > $R/$O/deftrace.log 2>&1
    rocprofv3 --kernel-trace --stats -d $R/$O/c3trace -o t --output-format csv -- python $R/scripts/diag/graph_step_time.py 1 1 0 > $R/$O/c3trace.log 2>&1
    rocprofv3 --kernel-trace --stats -d $R/$O/c5trace -o t --output-format csv -- python $R/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/c5trace.log 2>&1
    cd $R
    cp $(find $O/deftrace -name "*kernel_stats.csv" | head -1) $O/f_kernel_stats_default_bench.csv
    python scripts/diag/step_timeline.py $(find $O/c3trace -name "*kernel_trace.csv" | head -1) 20 > $O/h_c3_step_timeline.txt
    cp $(find $O/c3trace -name "*kernel_stats.csv" | head -1) $O/h_c3_step_kernel_stats.csv
    cp $(find $O/c5trace -name "*kernel_stats.csv" | head -1) $O/h_c5_image_kernel_stats.csv
    tail -2 $O/deftrace.log > $O/f_default_bench_line_under_rocprof.json
    rm -rf $O/deftrace $O/c3trace $O/c5trace
    tail -c 600 $O/i_bench_default.json
fi
