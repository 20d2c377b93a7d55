# usage: bash tools/lab/pmc_traffic.sh <outdir> <T> <in> <out> lib1.so ...   (FETCH_SIZE / WRITE_SIZE per launch)
OUT=$1; T=$2; IN=$3; OUTF=$4; shift 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
for lib in "$@"; do
  n=$(basename $lib .so)
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_${n} -o p -- python3 tools/lab/k1_one.py $lib $T $IN $OUTF > $OUT/f_${n}.txt 2>&1
  timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w_${n} -o p -- python3 tools/lab/k1_one.py $lib $T $IN $OUTF > $OUT/w_${n}.txt 2>&1
  python3 - $OUT $n <<'PY'
import csv, glob, os, sys
out, n = sys.argv[1:3]
def mean(sub):
    v = []
    for f in glob.glob(os.path.join(out, f"{sub}_{n}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "linear_gs" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    return sum(v[2:]) / max(len(v[2:]), 1)
print(f"{n}: FETCH_SIZE x2 = {mean('f')*2*1024/1e6:.1f} MB   WRITE_SIZE = {mean('w')*1024/1e6:.1f} MB per launch")
PY
done
