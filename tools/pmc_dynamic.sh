export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_dyn
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/a -- python3 $R/tools/bench_dynamic.py > $O/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 $R/tools/bench_dynamic.py > $O/b.log 2>&1
python3 - $O <<'PY'
import csv,sys,glob,collections
for sub in ('a','b'):
    f=glob.glob(sys.argv[1]+'/'+sub+'/**/*counter_collection.csv',recursive=True)
    if not f: print('no csv',sub); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'][:60]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    for k,v in acc.items():
        if any(s in k for s in ('k_hash_dynamic3','k_hash3d_lagrange','k_planes_fwd_runs','levels8','k_density_dynamic')):
            print(sub,k,{a:round(b) for a,b in v.items()})
PY
