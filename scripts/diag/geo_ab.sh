for v in base nosched sched8 sched2 r8 noreduce; do
  echo "== $v"; NERF_SOS_HIP_LIB=$PWD/exp/lib_geo_$v.so NSOS_SKIP_HASH_CHECK=1 python scripts/diag/geo_fuse_time.py 2>&1 | grep fused
done
