# usage: bash tools/lab/pmc_k1.sh <outdir> <T> <in> <out> lib1.so lib2.so ...   (GPU box; one rocprofv3 --pmc pass per lib and counter set)
set -e
OUT=$1; T=$2; IN=$3; OUTF=$4; shift 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
for lib in "$@"; do
  n=$(basename $lib .so)
  timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc_${n}_sq -o p -- python3 tools/lab/k1_one.py $lib $T $IN $OUTF > $OUT/pmc_${n}_sq.txt 2>&1
  timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_${n}_grbm -o p -- python3 tools/lab/k1_one.py $lib $T $IN $OUTF > $OUT/pmc_${n}_grbm.txt 2>&1
  timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_${n} -o p -- python3 tools/lab/k1_one.py $lib $T $IN $OUTF > $OUT/kt_${n}.txt 2>&1
done
python3 tools/lab/pmc_k1_summary.py $OUT "$@"
