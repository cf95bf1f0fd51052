#!/bin/bash
# Round-end evidence, part 1 (runs ON THE GPU BOX via gpurun): GPU tests, smoke, PMC traffic passes, phase tables, kernel timers.
set -x
O=gpurun_out/r03ev; mkdir -p $O
python -m pytest tests -m gpu -q > $O/b_gpu_tests.log 2>&1; tail -3 $O/b_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/b_smoke.log 2>&1; tail -1 $O/b_smoke.log
bash scripts/profile_traffic.sh r03final > $O/traffic.log 2>&1
python scripts/phase_profile_lp.py 2 bf16 > $O/c_phase_lp8_semcoord_bf16.txt 2>&1
python scripts/phase_profile_lp.py 0 fp16 > $O/c_phase_lp8_nosem_fp16.txt 2>&1
python scripts/phase_profile_lp.py 2 bf16 2 --save > $O/c_phase_lp8_semcoord_bf16_save.txt 2>&1
(python scripts/diag/lp_save_time.py; python scripts/diag/lp_time.py 4096 3; python scripts/diag/lp_time.py 65536 2) > $O/d_lp_times.txt 2>&1
python scripts/diag/graph_step_time.py > $O/d_graph_step_time.txt 2>&1
tail -3 $O/d_graph_step_time.txt
