#!/bin/bash
# BASELINE configs[2] (asymmetric_advantages + u8 observation of every step, k_rollout_encode): tests, then $LIBS alternating
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-ab8}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "${PYTEST:-1}" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x -k "${PYTEST_K:-observations or baseline_configs or encode}" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
fi
for rep in 1 2; do
for lib in ${LIBS}; do
  tag=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  timeout 300 python3 bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-traffic > $O/${tag}_c3_$rep.json 2>> $O/err.log
  for lay in ${EXTRA_LAYOUTS:-}; do
    timeout 300 python3 bench.py --config 3 --layout $lay --steps 4 --warmup 1 --no-cpu-baseline --no-traffic > $O/${tag}_c3_${lay}_$rep.json 2>> $O/err.log
  done
done
done
unset OC_AMD_LIB
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_[12].json")):
    try:
        d = json.load(open(f))
        print("%-48s %7.3f G  frac %.3f  launch_ms %.4f  parity %s" % (os.path.basename(f), d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], (d.get("parity_check") or {}).get("mismatches")))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -3 $O/err.log 2>/dev/null
