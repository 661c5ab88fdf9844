O=$GRAFT_REPO_ROOT/gpurun_out/r4z; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --one-mode --cpu-clouds 0 --steps 10 --warmup 3 > $O/sdma_$n.json 2> $O/sdma_$n.err
  python3 - <<PY
import json
d=json.loads([l for l in open("$O/sdma_$n.json") if l.startswith("{")][-1])
print("$n", "host", round(d["ms_per_step"],3), "resident", round(d["ms_per_step_resident"],3), flush=True)
PY
}
run default A=1
run sdma1 HSA_ENABLE_SDMA=1
run sdma0 HSA_ENABLE_SDMA=0
run blit0 GPU_FORCE_BLIT_COPY_SIZE=0
run default2 A=1
