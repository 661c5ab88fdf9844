# Round-3 profile collection on the GPU box: bash tools/profile_round.sh   (outputs under gpurun_out/r3p, copied to profiles/ by hand)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/${PROFILE_TAG:-r3p}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/trace.err; echo "trace rc=$?"
B="python3 bench.py --steps 1 --warmup 1 --batch 256 --cpu-clouds 0 --one-mode"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq --output-format csv -- $B > $O/sq.log 2>&1; echo "sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $B > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- $B > $O/write.log 2>&1; echo "write rc=$?"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc --output-format csv -- $B > $O/tcc.log 2>&1; echo "tcc rc=$?"
# the secondary workloads: bench line (with its CPU baseline) and kernel statistics of the same command without the CPU leg
WL=("s3dis" "pppf --batch 256" "pppe-train --graph")
[ -n "$PROFILE_ONLY_S3DIS" ] && WL=("s3dis")
for w in "${WL[@]}"; do
  n=$(echo $w | tr -d " -")
  timeout -k 10 300 python3 bench.py --workload $w > $O/$n.json 2> $O/$n.err; echo "$w bench rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$n -- python3 bench.py --workload $w --cpu-clouds 0 > $O/${n}_under_rocprof.json 2> $O/trace_$n.err; echo "$w trace rc=$?"
done
[ -n "$PROFILE_ONLY_S3DIS" ] || timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pppetraineager -- python3 bench.py --workload pppe-train --steps 10 --warmup 2 --cpu-clouds 0 > $O/pppetraineager_under_rocprof.json 2> $O/trace_pppetraineager.err; echo "train eager trace rc=$?"
ls $O | head -40
