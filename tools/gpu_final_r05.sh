# Round 5: the end-of-round check on one box: build() + smoke(), the whole GPU suite, the default bench line
set -o pipefail
O=gpurun_out/${1:-r05final}
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; tail -2 $O/smoke.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 560 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -2 $O/bench.err | cut -c1-300
python3 - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "check", d.get("check"), "frac", d["roofline"]["frac"])
oc = d.get("other_configs", {})
print("other_configs all_checks_ok:", oc.get("all_checks_ok"), " wall s:", d.get("wall_s") or d.get("driver_run_s"))
c3 = oc.get("config3", {})
for mode in ("fast", "faithful"):
    for k, v in (c3.get(mode) or {}).items():
        if isinstance(v, dict) and "us" in v:
            print(f"  config3 {mode:8s} {k:24s} {v['us']:7.1f} us  valu {v.get('frac_of_valu_bound')}  issue {v.get('frac_of_issue_bound')}  {v.get('check')}")
PY
exit $rc
