# SQ counters of the K1 phase kernels at h = 500 (tools/phase_bench.py): where a wave's cycles go
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_WAVE32_LDS SQ_WAVES"; do
  n=$(echo $pass | cut -d' ' -f1)
  PHASE_BENCH_DEPTHS=0 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_phase_$n -- python3 tools/phase_bench.py 500 > gpurun_out/pmc_phase_$n.log 2>&1 || echo "pass $n failed"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_phase_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'k_agg_stream' not in k and 'k_agg_phase' not in k: continue
            agg[k[:64]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()):
            print(k)
            for c,vals in v.items():
                vals=vals[len(vals)//2:]
                print('   %-24s %14.0f  (n=%d)'%(c, sum(vals)/len(vals), len(vals)))
PY
