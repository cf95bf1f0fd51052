set -x
O=gpurun_out/r05final; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/b_gpu_tests.log 2>&1; tail -3 $O/b_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/b_smoke.log 2>&1; tail -1 $O/b_smoke.log
python bench.py > $O/i_bench_default.out 2> $O/i_bench_default.err; tail -c 400 $O/i_bench_default.out
