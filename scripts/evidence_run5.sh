#!/bin/bash
# Round-4 evidence, second half (generic-architecture training, ray gradients), ON THE GPU BOX via gpurun, on the final build:
#   GPU tests + smoke, the default bench line, the generic training bench, rocprofv3 kernel stats and PMC passes of that bench.
set -x
O=gpurun_out/r04ev2; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
python -m pytest tests -m gpu -q > $O/b_gpu_tests.log 2>&1; tail -3 $O/b_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/b_smoke.log 2>&1; tail -1 $O/b_smoke.log
python bench.py > $O/i_bench_default.json 2> $O/i_bench_default.err
python scripts/bench_generic_train.py > $O/m_generic_train.txt 2>&1
python scripts/diag/raygrad_time.py > $O/p_raygrad_time.txt 2>&1
PROFILE_CMD="python $R/scripts/bench_generic_train.py 1024" bash scripts/profile_gpu.sh r04generic > $O/profile_generic.log 2>&1
cp gpurun_out/prof_r04generic/summary.txt $O/o_pmc_generic_train_summary.txt
cp $(find gpurun_out/prof_r04generic/trace -name "*kernel_stats.csv" | head -1) $O/n_generic_train_kernel_stats.csv
rm -rf gpurun_out/prof_r04generic
tail -c 400 $O/i_bench_default.json; cat $O/m_generic_train.txt $O/p_raygrad_time.txt
