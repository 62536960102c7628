O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  SPLAT_PROBE_STATS=0 timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/g28_pmc$i -o splat -- python $R/tools/splat_cells_probe.py 30000000 > $O/g28_pmc$i.log 2>&1
  tail -2 $O/g28_pmc$i.log | cut -c1-200
done
python - <<'PY'
import csv, glob, collections, os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
for d in sorted(glob.glob(O+'/g28_pmc?')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][-60:]
            acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        if 'cells_pass' in k or 'merge' in k or 'seed' in k:
            print(d[-5:], k[-50:], {a:int(b) for a,b in v.items()})
PY
