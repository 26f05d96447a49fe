# round 5: merged 8-wave attention + integrated pre-split 1x1 + tall kernel: attention / cond tests, new tests (RCCL single GPU,
# statistics-entry stress), the layout-conditioned rows (LC_ATTN_WAVES=4 vs default A/B)
export TMPDIR=/tmp
O=gpurun_out/r05n
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_rccl_single_gpu.py tests/test_presplit.py -m gpu -q -x -p no:cacheprovider -k "attention or cond or rccl or bench_takes or cli_under or entries_stress or presplit" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -8 | tee $O/pytest.txt
LC_ATTN_WAVES=4 timeout 600 python devtools/bench_rows.py --only cond_layout_v6_32x1024,cond_autoreg_v2_64x2048 > $O/rows_w4.json 2> $O/rows_w4.err
timeout 600 python devtools/bench_rows.py --only cond_layout_v6_32x1024,cond_autoreg_v2_64x2048,uncond_32x1024 > $O/rows.json 2> $O/rows.err
python - <<'PY'
import json
for f in ('rows_w4', 'rows'):
    d = json.load(open(f'gpurun_out/r05n/{f}.json'))
    for k, v in d.items():
        if isinstance(v, list):
            for r in v:
                print(f, k, {kk: r[kk] for kk in r if kk in ('batch', 'ms_per_step', 'steps_per_s', 'frac_of_f16_peak', 'roofline_frac', 'B')})
PY
