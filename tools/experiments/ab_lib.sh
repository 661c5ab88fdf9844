# A/B of two builds of libpccx.so on the default bench in ONE box: bash tools/experiments/ab_lib.sh <variant>  (libpccx_<variant>.so beside
# libpccx.so); alternates base / variant twice and restores the base library.
set -o pipefail
V=${1:-old}
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O
L=$GRAFT_REPO_ROOT/point-cloud-compression_amd/pccx/lib
cp $L/libpccx.so /tmp/base.so
for v in base $V base $V; do
  if [ $v = base ]; then cp /tmp/base.so $L/libpccx.so; else cp $L/libpccx_$v.so $L/libpccx.so; fi
  timeout -k 10 200 python3 bench.py --one-mode --cpu-clouds 0 --no-secondary --no-files --steps 5 --warmup 2 > $O/$v.json 2> $O/$v.err || { cp /tmp/base.so $L/libpccx.so; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("$O/$v.json") if l.startswith("{")][-1])
print("$v", round(d["value"]/1e6,2), d["stage_ms_per_step"]["sa_pn_forward"], d["stage_ms_per_step"]["ae_decode"], flush=True)
PY
done
cp /tmp/base.so $L/libpccx.so
