#!/bin/bash
# round 6: the rollout rate against the batch size on one GPU (the mover / interact kernel runs in rounds of one workgroup per CU up to
# 8 rounds, the one-wavefront instances beyond): python bench.py --envs N for the headline layout and the 5-layout mix
cd ${GRAFT_REPO_ROOT:-.}
for cfg in 2 4; do
for n in 65536 131072 262144 524288 1048576; do
  l=$(( n > 262144 ? 20 : 60 ))
  timeout 300 python3 bench.py --config $cfg --envs $n --steps 1 --warmup 1 --launches-per-step $l --no-extras --no-cpu-baseline --no-traffic 2>/dev/null > /tmp/sweep.json
  python3 - <<PY
import json
d=json.load(open("/tmp/sweep.json"))
print("config %d  %8d envs: %6.1f G env-steps/s  frac %.3f  launch %.3f ms  parity %s  (%s)" % ($cfg, $n, d["value"]/1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], (d.get("parity_check") or {}).get("mismatches"), d["config"].get("flags_layout")))
PY
done
done
