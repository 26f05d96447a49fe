#!/bin/bash
# PMC passes over one conv shape (run on the GPU box): bash devtools/pmc_conv.sh B Ci Co H W ks cfg
# (PMC_SCRIPT / PMC_KERNEL: another driver script and kernel-name pattern, e.g. devtools/attn_one.py + attn_h_kernel)
export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  mkdir -p gpurun_out/pmcc; rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcc/p$i -o p -- python ${PMC_SCRIPT:-devtools/conv_time.py} "$@" > gpurun_out/pmcc/log$i.txt 2>&1
done
PMC_KERNEL=${PMC_KERNEL:-conv_f16x2} python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmcc/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if os.environ['PMC_KERNEL'] in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    v = v[5:] if len(v) > 5 else v
    print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
