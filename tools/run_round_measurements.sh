# Round measurements (run on the GPU box through gpurun): bench lines, rocprof summaries, PMC counters.
# usage: bash tools/run_round_measurements.sh <tag> <part>
#   part 1: bench lines: the default run (cfg3, with the CPU baseline and the other configs' values), cfg1, cfg2, cfg4,
#           cfg5, and the 2-rank rehearsal of the launcher on the one-GPU box
#   part 2: rocprofv3 --kernel-trace --stats of bench cfg3 / cfg1 / cfg2 / cfg5, cfg3 step timeline, FETCH/WRITE traffic
#           of the cfg3 and cfg2 hot paths
#   part 3: MFMA / clock / traffic counters of the cfg3, cfg5 (one real 2048-sample chunk) and cfg1 hot-path replay
#           (tools/pmc_mfma.sh)
# Summaries are written into profiles/ on the box and copied under gpurun_out/<tag>/ (which is what comes back).
set -e
TAG=${1:-r02}
PART=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
if [ "$PART" = "1" ]; then
timeout -k 10 500 python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout -k 10 200 python bench.py --config cfg1 --no-cpu-baseline --no-other-configs --steps 20 > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout -k 10 300 python bench.py --config cfg2 --no-cpu-baseline --no-other-configs > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout -k 10 300 python bench.py --config cfg4 --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout -k 10 400 python bench.py --config cfg5 --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
DPK_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --no-other-configs --steps 5 > $O/bench_cfg3_2ranks_one_gpu.json 2> $O/bench_cfg3_2ranks_one_gpu.err
elif [ "$PART" = "2" ]; then
for C in cfg3 cfg1 cfg2 cfg5; do
  S=10; if [ $C = cfg5 ]; then S=2; fi; if [ $C = cfg1 ]; then S=20; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$C -o $C -- python3 bench.py --config $C --no-cpu-baseline --no-other-configs --steps $S > $O/prof_$C.log 2>&1
  python tools/summarize_prof.py $O/prof_$C ${TAG}_bench_$C rocprofv3 --kernel-trace --stats -- python3 bench.py --config $C --no-cpu-baseline --no-other-configs --steps $S > $O/prof_${C}_summary.md
  if [ $C = cfg3 ]; then python tools/step_timeline.py $O/prof_$C > $O/${TAG}_cfg3_step_timeline.txt; fi
  rm -rf $O/prof_$C
done
cp profiles/${TAG}_bench_cfg*_kernel_stats.* $O/
for C in cfg3 cfg2; do
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 tools/hotpath_replay.py $C 4 > $O/pmc_fetch_$C.log 2>&1
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 tools/hotpath_replay.py $C 4 > $O/pmc_write_$C.log 2>&1
  python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write ${TAG}_$C > $O/pmc_summary_$C.txt
  cp profiles/${TAG}_${C}_pmc_traffic.json $O/
  rm -rf $O/pmc_fetch $O/pmc_write
done
else
bash tools/pmc_mfma.sh $TAG cfg3 > $O/mfma_cfg3.txt
bash tools/pmc_mfma.sh $TAG cfg5 2048 2 > $O/mfma_cfg5.txt
bash tools/pmc_mfma.sh $TAG cfg1 > $O/mfma_cfg1.txt
rm -rf $O/mfma_cfg*/sq $O/mfma_cfg*/grbm $O/mfma_cfg*/fetch $O/mfma_cfg*/write $O/mfma_cfg*/kt
fi
