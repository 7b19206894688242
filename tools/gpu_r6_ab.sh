#!/bin/bash
# round 6: same-box A/B of library builds on the mover / interact legs — the 5-layout mix (--config 4), single layouts and the
# headline; optional parity tests of the tree's build first (TESTK), REPS alternations
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r6ab}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ -n "${TESTK:-}" ]; then
  timeout 900 python3 -m pytest tests -x -q -m gpu -k "$TESTK" > $O/pytest.log 2>&1
  tail -3 $O/pytest.log
fi
run() { tag=$1; shift; timeout 300 python3 bench.py --steps ${STEPS:-3} --warmup 1 --no-extras --no-cpu-baseline --no-traffic ${PARITY:-} "$@" > $O/$tag.json 2>> $O/err.log; }
for rep in $(seq 1 ${REPS:-1}); do
for lib in ${LIBS}; do
  t=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  [ "${MIX:-1}" = 1 ] && run ${t}_mix_$rep --config 4
  for lay in ${LAYOUTS:-asymmetric_advantages counter_circuit}; do run ${t}_${lay}_$rep --layout $lay; done
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
