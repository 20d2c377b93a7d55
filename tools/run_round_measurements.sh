set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.log 2>&1
timeout -k 10 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout -k 10 200 python bench.py --config cfg1 --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout -k 10 300 python bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o cfg2 -- python3 bench.py --no-cpu-baseline --steps 20 > $O/prof.log 2>&1
python tools/summarize_prof.py $O/prof r01_bench_cfg2_fused rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 20 > $O/prof_summary.md
python tools/step_timeline.py $O/prof > $O/timeline.txt
cp profiles/r01_bench_cfg2_fused_kernel_stats.* $O/
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 tools/hotpath_replay.py cfg2 4 > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 tools/hotpath_replay.py cfg2 4 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write r01b_cfg2 > $O/pmc_summary.txt
cp profiles/r01b_cfg2_pmc_traffic.json $O/
