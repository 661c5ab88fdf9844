L=$GRAFT_REPO_ROOT/point-cloud-compression_amd/pccx/lib
for v in base map32 base map8 map64; do
  if [ $v = base ]; then unset PCCX_LIB; else export PCCX_LIB=$L/libpccx_$v.so; fi
  echo "== $v"; python3 tools/experiments/r5/pppf_union_max_ab.py 2>&1 | grep "union_max True" | tail -1 | cut -c1-260
done
