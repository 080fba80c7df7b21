set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  n=$(echo $pass | cut -d' ' -f1)
  MB_GRAPH=0 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_lds_$n -- python3 tools/k1_lds_bench.py > gpurun_out/pmc_lds_$n.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_lds_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'k_agg_lds' not in k: continue
            agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            print(k)
            for c,vals in v.items():
                vals=vals[len(vals)//2:]
                print('   %-24s %14.0f  (n=%d)'%(c, sum(vals)/len(vals), len(vals)))
PY
