set -x
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_generic_arch.py tests/test_gpu_raygrad.py tests/test_gpu_pointquery.py tests/test_gpu_generic_round5.py tests/test_gpu_sharded.py -m gpu -q -x > $O/gpu_tests_generic.log 2>&1; tail -3 $O/gpu_tests_generic.log
timeout 300 python scripts/diag/generic_time.py > $O/l_generic_kernel_times.txt 2>&1; cat $O/l_generic_kernel_times.txt
timeout 200 python scripts/diag/raygrad_time.py > $O/p_raygrad_time.txt 2>&1; cat $O/p_raygrad_time.txt
timeout 200 python scripts/diag/generic_head_only_time.py > $O/r_generic_head_only.txt 2>&1; cat $O/r_generic_head_only.txt
