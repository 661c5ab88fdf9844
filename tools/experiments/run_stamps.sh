L=point-cloud-compression_amd/pccx/lib; cp $L/libpccx.so /tmp/base.so
for w in 0 256; do cp $L/libpccx_stamps$w.so $L/libpccx.so; echo "== observed thread $w"; timeout -k 10 200 python3 tools/experiments/fused_stamps.py || break; done
cp /tmp/base.so $L/libpccx.so
