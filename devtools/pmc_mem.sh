#!/bin/bash
# Memory-path PMC passes over one conv shape -- UNTESTED counter sets: the first attempt aborted inside rocprofv3 and hung;
# every pass now runs under its own 90 s timeout. (run on the GPU box): bash devtools/pmc_mem.sh B:Ci:Co:H:W [flags] --cfg N
export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_READ_REQ_LATENCY_sum" "TCC_WRITE_REQ_LATENCY_sum TCC_READ_REQ_sum TCC_WRITE_REQ_sum TCC_BUSY_sum"; do
  i=$((i+1))
  mkdir -p gpurun_out/pmcm; timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcm/p$i -o p -- python devtools/conv_time.py "$@" > gpurun_out/pmcm/log$i.txt 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmcm/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'conv_f16x2' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    v = v[5:] if len(v) > 5 else v
    print(f"{k:40s} n={len(v):3d} mean={sum(v)/len(v):18.1f}")
PY
rm -rf gpurun_out/pmcm
