#!/bin/bash
# round 5, call A: issue rate by wavefronts per SIMD (tools/issue_rate.hip) + the per-env-terrain legs as they stand
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5a
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 120 tools/issue_rate > $O/issue_rate.txt 2>&1
cat $O/issue_rate.txt
for cfg in 4 5; do
  timeout 300 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-traffic > $O/config$cfg.json 2>> $O/err.log
done
python3 - <<PY
import json
for c in (4, 5):
    try:
        d = json.load(open("$O/config%d.json" % c))
        print(c, "%.1f G frac %.3f launch_ms %.4f parity %s" % (d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], (d.get("parity_check") or {}).get("mismatches")))
    except Exception as e:
        print(c, "ERR", e)
PY
tail -3 $O/err.log 2>/dev/null
