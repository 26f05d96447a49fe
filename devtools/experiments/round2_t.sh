S="8:64:64:32:1024:3 8:128:64:32:1024:3 8:64:128:32:1024:3 8:128:128:16:512:3"
for v in pf0 pf2 cur pf5 pfx; do
  L="--lib devtools/variants/liblc_$v.so"; [ $v = cur ] && L=""
  echo "== $v plain";       python devtools/conv_bench.py $L $S
  echo "== $v gn emit res"; python devtools/conv_bench.py $L --gn --emit --res $S
done
for v in pf0 cur; do
  L=""; [ $v = pf0 ] && L="LC_HIP_LIB=$PWD/devtools/variants/liblc_pf0.so"
  echo "== bench $v"; env $L python bench.py --no-cpu-baseline --no-verify 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'])"
done
