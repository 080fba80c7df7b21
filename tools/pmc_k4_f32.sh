#!/bin/bash
# MFMA utilisation of the fp32 K4 kernels (gv_made_chain_f32, gv_made_gradw_f32) from SQ counters, one counter pass:
#   tools/pmc_k4_f32.sh  -> gpurun_out/k4_f32_pmc.txt
# SQ_VALU_MFMA_BUSY_CYCLES sums the cycles a SIMD's matrix pipe is busy over all SIMDs; SQ_INSTS_VALU_MFMA_MOPS_F32 counts
# 512-flop units.  utilisation = busy cycles / (launch duration x 2.4 GHz x 1024 SIMDs).
set -e -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/k4_f32_pmc.txt
rm -rf /tmp/pmc_k4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_WAVES --output-format csv -d /tmp/pmc_k4 -o p -- python3 tools/probes/chain32_probe.py 14741 > /tmp/pmc_k4.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_WAVES --output-format csv -d /tmp/pmc_k4g -o p -- python3 tools/probes/gradw32_probe.py 73705 > /tmp/pmc_k4g.log 2>&1
python3 - "$(find /tmp/pmc_k4 -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_k4g -name '*counter_collection.csv' | head -1)" > "$out" <<'PY'
import collections, csv, sys
print('# tools/pmc_k4_f32.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_WAVES over')
print('# tools/probes/chain32_probe.py 14741 (forward / forward without hidden stores / backward-x chains, masked and dense plans)')
print('# and tools/probes/gradw32_probe.py 73705; per-launch averages over all launches of the kernel in the probe.')
for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r['Kernel_Name'].split('(')[0][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in sorted(agg.items()):
        if 'chain_f32' in k or 'gradw32' in k:
            m = {c: sum(v) / len(v) for c, v in d.items()}
            n = len(next(iter(d.values())))
            print(f"{k:40s} launches {n:5d}  MFMA busy cycles {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):14.0f}  MFMA 512-flop units {m.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0):12.0f}"
                  f"  = {m.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) * 512 / 1e9:6.2f} GFLOP executed  CU busy cycles {m.get('SQ_BUSY_CU_CYCLES', 0):14.0f}  waves {m.get('SQ_WAVES', 0):6.0f}")
PY
grep -v amdgpu.ids /tmp/pmc_k4.log >> "$out"
grep -v amdgpu.ids /tmp/pmc_k4g.log >> "$out"
cat "$out"
