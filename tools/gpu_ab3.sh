#!/bin/bash
# A/B of $LIBS on BASELINE configs[4] at 131 072 and 65 536 envs (the two launch shapes of the one-pot HBM-table instances)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-ab3}
mkdir -p $O
cd $R
export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in ${LIBS}; do
  tag=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  timeout 300 python3 bench.py --config 5 --envs 131072 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --parity-steps 1200 > $O/${tag}_c5_$rep.json 2>> $O/err.log
  timeout 300 python3 bench.py --config 5 --envs 65536 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --parity-steps 1200 > $O/${tag}_c5s_$rep.json 2>> $O/err.log
done
done
unset OC_AMD_LIB
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_[123].json")):
    try:
        d = json.load(open(f))
        print("%-44s %7.1f G  frac %.3f  launch_ms %.4f  parity %s" % (os.path.basename(f), d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], (d.get("parity_check") or {}).get("mismatches")))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -3 $O/err.log 2>/dev/null
