# on the GPU box: bash tools/experiments/r4/run_graph_probe.sh   (needs pccx/lib/libpccx_memset.so, built here with PCCX_BUILD_TAG=memset)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
L=$GRAFT_REPO_ROOT/point-cloud-compression_amd/pccx/lib
cp $L/libpccx.so /tmp/base.so
P=tools/experiments/r4/graph_memset_probe.py
timeout -k 10 200 python3 $P product 3 2>&1 | grep "^\[\|memset" | tee $O/product.log
cp $L/libpccx_memset.so $L/libpccx.so
timeout -k 10 200 python3 $P memset 6 2>&1 | grep "^\[\|memset" | tee $O/memset.log
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout -k 10 200 python3 $P memset_nocapture 6 2>&1 | grep "^\[\|memset" | tee $O/memset_nocapture.log
AMD_SERIALIZE_KERNEL=3 timeout -k 10 200 python3 $P memset_serialize 4 2>&1 | grep "^\[\|memset" | tee $O/memset_serialize.log
cp /tmp/base.so $L/libpccx.so
