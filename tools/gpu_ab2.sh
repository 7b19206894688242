#!/bin/bash
# A/B of library variants on one box (round 4, second session): every lib in $LIBS runs the headline twice; the libs in
# $LIBS_FULL additionally run BASELINE configs[3] / [4] (the per-env-terrain legs); $PYTEST_LIB (if set) runs the gpu suite.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-ab2}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ -n "${PYTEST_LIB:-}" ]; then
  OC_AMD_LIB=$R/$PYTEST_LIB timeout 900 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
fi
for rep in 1 2; do
for lib in ${LIBS:-overcooked_ai_amd/liboc_amd.so}; do
  tag=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  timeout 300 python3 bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-traffic > $O/${tag}_c2_$rep.json 2>> $O/err.log
done
for lib in ${LIBS_FULL:-}; do
  tag=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  timeout 300 python3 bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --parity-steps 1200 > $O/${tag}_c4_$rep.json 2>> $O/err.log
  timeout 300 python3 bench.py --config 5 --envs 131072 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --parity-steps 1200 > $O/${tag}_c5_$rep.json 2>> $O/err.log
  for lay in ${EXTRA_LAYOUTS:-}; do
    timeout 300 python3 bench.py --layout $lay --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --parity-steps 1200 > $O/${tag}_${lay}_$rep.json 2>> $O/err.log
  done
done
done
unset OC_AMD_LIB
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_[12].json")):
    try:
        d = json.load(open(f))
        print("%-44s %7.1f G  frac %.3f  launch_ms %.4f  parity %s" % (os.path.basename(f), d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], (d.get("parity_check") or {}).get("mismatches")))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -5 $O/err.log 2>/dev/null
