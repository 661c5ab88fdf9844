# usage: bash ab_multi.sh v1 v2 ...  : base and each variant, two rounds, stage times
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O
L=$GRAFT_REPO_ROOT/point-cloud-compression_amd/pccx/lib
cp $L/libpccx.so /tmp/base.so
for r in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so $L/libpccx.so; else cp $L/libpccx_$v.so $L/libpccx.so; fi
  timeout -k 10 200 python3 bench.py --one-mode --cpu-clouds 0 --steps 5 --warmup 2 > $O/$v.json 2> $O/$v.err || { cp /tmp/base.so $L/libpccx.so; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("$O/$v.json") if l.startswith("{")][-1])
s=d["stage_ms_per_step"]
print("$v", round(d["value"]/1e6,2), s["sa_pn_forward"], s["ae_decode"], s["knn_patches"], s["prob"], flush=True)
PY
done
done
cp /tmp/base.so $L/libpccx.so
