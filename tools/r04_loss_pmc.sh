set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/loss_pmc; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/a -- python3 $R/tools/loss_probe.py > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -- python3 $R/tools/loss_probe.py > $O/b.log 2>&1 || rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/b -- python3 $R/tools/loss_probe.py > $O/b.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 $R/tools/loss_probe.py > $O/s.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'/[ab]/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'loss_' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c:'%.3g'%(sum(x)/len(x)) for c,x in sorted(v.items())})
for f in glob.glob(O+'/s/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'loss_' in r['Name']: print(r['Name'][:60], r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
