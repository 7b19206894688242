#!/bin/bash
# round 5: MODE 4 (joint-table kernel split into mover + interact wavefronts): parity + the headline A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5d}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_launch_shapes.py -x -q -m gpu -k "mover or tiled or bench_launch" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
run() {  # tag, args...
  tag=$1; shift
  timeout 300 python3 bench.py --steps ${STEPS:-5} --warmup 2 --no-extras --no-cpu-baseline --no-traffic "$@" > $O/$tag.json 2>> $O/err.log
}
for rep in 1 2 3; do
  run head_duo_$rep
  run head_one_$rep --one-wavefront
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
        so = (d["roofline"].get("store_only") or {})
        print("%-30s %7.1f G  frac %.3f  launch_ms %.4f  parity %s  store_only %.1f G (%s)" % (os.path.basename(f), d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], (d.get("parity_check") or {}).get("mismatches"), so.get("env_steps_per_s", 0) / 1e9, so.get("flags_layout")))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
grep -v amdgpu.ids $O/err.log 2>/dev/null | tail -5
