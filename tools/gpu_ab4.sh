#!/bin/bash
# sustained (driver-length) and short headline runs + the 1 M-env shape, alternating between the libs in $LIBS on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-ab4}
mkdir -p $O
cd $R
export TMPDIR=/tmp
for rep in 1 2; do
for lib in ${LIBS}; do
  tag=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/${tag}_long_$rep.json 2>> $O/err.log
  timeout 300 python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/${tag}_short_$rep.json 2>> $O/err.log
  timeout 300 python3 bench.py --envs 1048576 --steps 1 --warmup 1 --launches-per-step 20 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/${tag}_1M_$rep.json 2>> $O/err.log
done
done
unset OC_AMD_LIB
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_[12].json")):
    try:
        d = json.load(open(f))
        print("%-40s %7.1f G  frac %.3f  launch_ms %.4f  region %.2f s" % (os.path.basename(f), d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d.get("timed_region_s", 0)))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -3 $O/err.log 2>/dev/null
