# MFMA / occupancy counters of the hot-path kernels of one config (GPU box; each rocprofv3 pass is its own run, the
# program directly after `--`).  usage: bash tools/pmc_mfma.sh <tag> <cfg> [batch] [iters]
set -e
TAG=$1; CFG=$2; BATCH=${3:-}; ITERS=${4:-4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG/mfma_$CFG
mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o p -- python3 tools/hotpath_replay.py $CFG $ITERS $BATCH > $O/sq.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/grbm -o p -- python3 tools/hotpath_replay.py $CFG $ITERS $BATCH > $O/grbm.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 tools/hotpath_replay.py $CFG $ITERS $BATCH > $O/fetch.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 tools/hotpath_replay.py $CFG $ITERS $BATCH > $O/write.txt 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python3 tools/hotpath_replay.py $CFG $ITERS $BATCH > $O/kt.txt 2>&1
python3 tools/pmc_mfma_summary.py $O ${TAG}_${CFG} "hotpath_replay.py $CFG $ITERS $BATCH"
cp profiles/${TAG}_${CFG}_pmc_mfma.md profiles/${TAG}_${CFG}_pmc_mfma.json $O/
