#!/bin/bash
# Round-end evidence, part 2 (runs ON THE GPU BOX via gpurun, after profiles/r03/traffic.json was regenerated from part 1's PMC
# passes): the default bench line, the other configurations as main lines, rocprofv3 kernel stats of the c3 / c5 steps, and the
# kernel-stats + SQ / LDS PMC passes of the default bench command.
set -x
O=gpurun_out/r03ev; mkdir -p $O
python bench.py > $O/e_bench_default.json 2> $O/e_bench_default.err
python bench.py --config c3 --steps 30 --warmup 5 > $O/e_bench_c3.json 2>/dev/null
python bench.py --config c4 --steps 30 --warmup 5 > $O/e_bench_c4.json 2>/dev/null
python bench.py --config c5 --steps 3 --warmup 1 > $O/e_bench_c5.json 2>/dev/null
python scripts/bench_full_train.py 4096 fp16x3 > $O/e_bench_full_train.json 2>/dev/null
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/c3trace -o t --output-format csv -- python $R/bench.py --config c3 --steps 30 --warmup 5 > $R/$O/c3trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/c5trace -o t --output-format csv -- python $R/bench.py --config c5 --steps 2 --warmup 1 > $R/$O/c5trace.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +4M -delete
bash scripts/profile_gpu.sh r03 > $O/profile_gpu.log 2>&1
cp gpurun_out/prof_r03/summary.txt $O/f_pmc_default_bench_summary.txt
cp $(find gpurun_out/prof_r03/trace -name "*kernel_stats.csv" | head -1) $O/f_kernel_stats_default_bench.csv
tail -c 600 $O/e_bench_default.json
