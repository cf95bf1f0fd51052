#!/bin/bash
# GPU box: A/B of mlp_lp16_kernel builds (scripts/diag/build_variant.sh -> ab/lib_<name>.so) against the tree's library:
# fine-pass time (sem+coord fp16 / bf16, 4096 rays) and the phase table of each.   usage: lp16_ab.sh <out dir> [names...]
OUT=$1; shift
mkdir -p $OUT
for name in tree "$@"; do
    if [ $name = tree ]; then unset NERF_SOS_HIP_LIB; else export NERF_SOS_HIP_LIB=$PWD/ab/lib_$name.so; fi
    echo "== $name" | tee -a $OUT/ab_times.txt
    timeout 300 python scripts/diag/lp_time.py 4096 3 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_times.txt
    timeout 200 python scripts/phase_profile_lp.py 2 fp16 3 > $OUT/phase_${name}_2_fp16.txt 2>&1
    tail -3 $OUT/phase_${name}_2_fp16.txt | tee -a $OUT/ab_times.txt
done
