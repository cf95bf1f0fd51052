for v in base nomfma noread nocvt l2 nomfma_noread l2_nomfma_noread; do
  echo "== $v"; NERF_SOS_HIP_LIB=$PWD/exp/lib_wg_$v.so python scripts/diag/wgrad_time.py 4096 2>&1 | grep "S="
done
