#!/bin/bash
# round 5: same-box A/B of two library builds on the mover / interact legs: the 5-layout mix (--config 4) and single two-pot layouts;
# the parity tests of the tree's build first
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5ab2}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -x -q -m gpu -k "${TESTK:-launch_shape or tiled or mover or regen or random_start or layout}" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
run() { tag=$1; shift; timeout 300 python3 bench.py --steps ${STEPS:-3} --warmup 1 --no-extras --no-cpu-baseline --no-traffic "$@" > $O/$tag.json 2>> $O/err.log; }
for rep in 1 2; do
for lib in ${LIBS}; do
  t=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  run ${t}_mix_$rep --config 4
  for lay in ${LAYOUTS:-asymmetric_advantages coordination_ring counter_circuit}; do run ${t}_${lay}_$rep --layout $lay; done
done
done
unset OC_AMD_LIB
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
        print("%-50s %7.1f G  frac %.3f  launch_ms %.4f  parity %s" % (os.path.basename(f), d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], (d.get("parity_check") or {}).get("mismatches")))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
grep -v amdgpu.ids $O/err.log 2>/dev/null | tail -5
