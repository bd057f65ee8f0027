set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_solve; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU --output-format csv -d $OUT/g1 -- python3 $ROOT/scripts/tmp/t_solve.py > $OUT/g1.log 2>&1 || tail -5 $OUT/g1.log
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(out+'/g*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        tot[k][r['Counter_Name']]+=float(r['Counter_Value'])
dur=collections.defaultdict(list)
for f in glob.glob(out+'/g*/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'].split('(')[0][-40:]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,d in tot.items():
    if 'fit_' not in k: continue
    n=len(dur[k]); print(k, 'calls',n,'avg us', sum(dur[k])/n)
    for c,v in sorted(d.items()): print('   ',c, f'{v/n:.4g}')
PY
