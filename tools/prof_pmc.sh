# usage: bash tools/prof_pmc.sh <tag> "<counters>" <python script> [args...]  -> per-kernel mean of each counter (rocprofv3 --pmc, counters only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
CTRS=$1; shift
cd /tmp
rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/"$@" > $R/gpurun_out/$TAG.log 2>&1
TAG=$TAG python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/" + os.environ["TAG"] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "march" in k or "composite" in k or (os.environ.get("ALLK") and (os.environ["ALLK"] == "1" or os.environ["ALLK"] in k)):
        print(k, {c: (round(sum(v) / len(v)), len(v)) for c, v in d.items()})
PY
