set -x
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python scripts/bench_generic_train.py > $O/m_generic_train.txt 2>&1; cat $O/m_generic_train.txt
timeout 200 python scripts/diag/raygrad_time.py > $O/p_raygrad_time.txt 2>&1; cat $O/p_raygrad_time.txt
timeout 200 python scripts/diag/generic_head_only_time.py > $O/r_generic_head_only.txt 2>&1; cat $O/r_generic_head_only.txt
