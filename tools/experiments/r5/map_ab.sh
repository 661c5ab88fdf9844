# A/B of the dropped m-block map of planes_gemm_kernel (README.md: "m-blocks of a row block spread over a CU's workgroup slots").  The kernel-side
# change is NOT in the tree: in planes_gemm_kernel, `blk = (chunk * G + c % G) * 8 + (blockIdx.x & 7); mb = c / G` with
# chunk = s / (G * MBS), c = s % (G * MBS), and the grid padded to a multiple of G row blocks per XCD (-DPG_MAP_CU=G, built as libpccx_map<G>.so).
L=$GRAFT_REPO_ROOT/point-cloud-compression_amd/pccx/lib
for v in base map32 base map8 map64; do
  if [ $v = base ]; then unset PCCX_LIB; else export PCCX_LIB=$L/libpccx_$v.so; fi
  echo "== $v"; python3 tools/experiments/r5/pppf_union_max_ab.py 2>&1 | grep "union_max True" | tail -1 | cut -c1-260
done
