# Profile collection on the GPU box: PROFILE_TAG=r5p bash tools/profile_round.sh   (outputs under gpurun_out/$PROFILE_TAG, copied to profiles/
# by tools/collect_profiles.py).  Kernel statistics and counters come from SEPARATE runs (gpurun refuses --pmc together with trace domains).
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/${PROFILE_TAG:-r5p}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ -z "$PROFILE_SKIP_MAIN" ]; then
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/trace.err; echo "trace rc=$?"
# the headline legs alone: the kernel statistics whose sa_pn_forward_h2_kernel average is the HEADLINE's launches only (the default run's
# statistics mix in the room-scale workload's launches of the same kernel, which cover fewer patches)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_headline -- python3 bench.py --steps 20 --warmup 5 --no-secondary --cpu-clouds 0 > $O/headline_under_rocprof.json 2> $O/trace_headline.err; echo "headline trace rc=$?"
fi
if [ -z "$PROFILE_SKIP_PMC" ]; then
B="python3 bench.py --steps 1 --warmup 1 --batch 256 --cpu-clouds 0 --no-secondary --no-files"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq --output-format csv -- $B > $O/sq.log 2>&1; echo "sq rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $B > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- $B > $O/write.log 2>&1; echo "write rc=$?"
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc --output-format csv -- $B > $O/tcc.log 2>&1; echo "tcc rc=$?"
fi
# the secondary workloads: bench line (with its CPU baseline) and kernel statistics of the same command without the CPU leg
if [ -z "$PROFILE_SKIP_SECONDARY" ]; then
WL=("s3dis" "pppf --batch 256" "pppe-train --graph" "ipdae-train")
for w in "${WL[@]}"; do
  n=$(echo $w | tr -d " -")
  timeout -k 10 400 python3 bench.py --workload $w > $O/$n.json 2> $O/$n.err; echo "$w bench rc=$?"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$n -- python3 bench.py --workload $w --cpu-clouds 0 --one-mode > $O/${n}_under_rocprof.json 2> $O/trace_$n.err; echo "$w trace rc=$?"
done
fi
ls $O | head -40
