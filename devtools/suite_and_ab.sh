# full GPU suite + smoke, then same-box A/Bs of the headline: up-path fold off / on, pre-split in_proj off / on.  bash devtools/suite_and_ab.sh TAG
export TMPDIR=/tmp
T=${1:-r06w}
O=$PWD/gpurun_out/$T
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -12 > $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
B="timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-rows"
LC_FOLD_UP=0 LC_PS1X1_MIN_CO=100000 $B 2>&1 | tail -1 > $O/bench_old.json
$B 2>&1 | tail -1 > $O/bench_new.json
LC_FOLD_UP=0 $B 2>&1 | tail -1 > $O/bench_fold0.json
LC_PS1X1_MIN_CO=100000 $B 2>&1 | tail -1 > $O/bench_inproj0.json
LC_FOLD_UP=0 LC_PS1X1_MIN_CO=100000 $B 2>&1 | tail -1 > $O/bench_old_b.json
$B 2>&1 | tail -1 > $O/bench_new_b.json
cat $O/pytest.txt $O/smoke.txt; for f in old new fold0 inproj0 old_b new_b; do echo -n "$f: "; python -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print(d['value'], d['ms_per_step'], d.get('verify'))"; done
