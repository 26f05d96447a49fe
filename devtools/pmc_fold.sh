#!/bin/bash
# PMC passes over one shape of the folded down-sampling stage (run on the GPU box): bash devtools/pmc_fold.sh B Ci Co H W
# Separate passes per counter set (MI355X_MICROARCH.md); FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE x1, units KB.
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmcf; rm -rf $O; mkdir -p $O
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp; rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python $GRAFT_REPO_ROOT/devtools/fold_down_one.py "$@" > $O/log$i.txt 2>&1)
done
python - "$O" "$@" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
print("# shape B Ci Co H W =", " ".join(sys.argv[2:]), "(30 launches, the first 5 dropped; per launch)")
for pat in ("conv_s2_shared_w", "fir_down2_prefilter"):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob(O + '/p*/p_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print("##", pat)
    for k, v in acc.items():
        v = v[5:] if len(v) > 5 else v
        print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
find $O -name "*.csv" -size +200k -delete
