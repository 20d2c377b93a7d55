# Round measurements (run on the GPU box through gpurun): bench lines, rocprof summaries, PMC counters.
# usage: bash tools/run_round_measurements.sh <tag> <part>
#   part 1: bench lines of every config (+ the 2-rank rehearsal of the launcher on the one-GPU box)
#   part 2: rocprofv3 --kernel-trace --stats of bench cfg2 / cfg3, step timeline, FETCH/WRITE traffic of the cfg2 hot path
#   part 3: MFMA / clock / traffic counters of the cfg3 and cfg5 hot-path replay (tools/pmc_mfma.sh)
# Summaries are written into profiles/ on the box and copied under gpurun_out/<tag>/ (which is what comes back).
set -e
TAG=${1:-r02}
PART=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
if [ "$PART" = "1" ]; then
timeout -k 10 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout -k 10 200 python bench.py --config cfg1 --no-cpu-baseline --steps 20 > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout -k 10 300 python bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout -k 10 300 python bench.py --config cfg4 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout -k 10 400 python bench.py --config cfg5 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
DPK_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --no-other-configs --steps 5 > $O/bench_cfg2_2ranks_one_gpu.json 2> $O/bench_cfg2_2ranks_one_gpu.err
elif [ "$PART" = "2" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o cfg2 -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 > $O/prof.log 2>&1
python tools/summarize_prof.py $O/prof ${TAG}_bench_cfg2 rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 > $O/prof_summary.md
python tools/step_timeline.py $O/prof > $O/${TAG}_cfg2_step_timeline.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3 -o cfg3 -- python3 bench.py --config cfg3 --no-cpu-baseline --steps 10 > $O/prof3.log 2>&1
python tools/summarize_prof.py $O/prof3 ${TAG}_bench_cfg3 rocprofv3 --kernel-trace --stats -- python3 bench.py --config cfg3 --no-cpu-baseline --steps 10 > $O/prof3_summary.md
cp profiles/${TAG}_bench_cfg*_kernel_stats.* $O/
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 tools/hotpath_replay.py cfg2 4 > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 tools/hotpath_replay.py cfg2 4 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write ${TAG}_cfg2 > $O/pmc_summary.txt
cp profiles/${TAG}_cfg2_pmc_traffic.json $O/
rm -rf $O/prof $O/prof3 $O/pmc_fetch $O/pmc_write
else
bash tools/pmc_mfma.sh $TAG cfg3 > $O/mfma_cfg3.txt
bash tools/pmc_mfma.sh $TAG cfg5 32 > $O/mfma_cfg5.txt
bash tools/pmc_mfma.sh $TAG cfg2 > $O/mfma_cfg2.txt
rm -rf $O/mfma_cfg*/sq $O/mfma_cfg*/grbm $O/mfma_cfg*/fetch $O/mfma_cfg*/write $O/mfma_cfg*/kt
fi
