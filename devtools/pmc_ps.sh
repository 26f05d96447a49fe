#!/bin/bash
# PMC passes over one conv shape on the pre-split kernel:  bash devtools/pmc_ps.sh TAG [conv_bench args]
export TMPDIR=/tmp
tag=$1; shift
O=$PWD/gpurun_out/pmc_$tag
mkdir -p $O
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  (cd /tmp; rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python $GRAFT_REPO_ROOT/devtools/conv_bench.py "$@" > $O/log$i.txt 2>&1)
done
python - "$O" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + '/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'conv_f16x2' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    v = v[5:] if len(v) > 5 else v
    print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
find $O -name "*.csv" -size +200k -delete
