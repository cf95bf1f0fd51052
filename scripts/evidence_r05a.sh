set -x
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.out 2> $O/bench_default.err; tail -c 1500 $O/bench_default.out
timeout 300 python scripts/diag/generic_time.py > $O/l_generic_kernel_times.txt 2>&1; cat $O/l_generic_kernel_times.txt
