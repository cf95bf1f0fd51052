set -x
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_generic_arch.py tests/test_gpu_raygrad.py tests/test_gpu_pointquery.py tests/test_gpu_generic_round5.py tests/test_gpu_sharded.py -m gpu -q -x > $O/gpu_tests_generic.log 2>&1; tail -3 $O/gpu_tests_generic.log
timeout 600 python scripts/bench_generic_train.py > $O/m_generic_train.txt 2>&1; cat $O/m_generic_train.txt
