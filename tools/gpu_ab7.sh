#!/bin/bash
# the big-batch instance of the headline kernel (16-bit cell words, no one-step-ahead reads): 1 M and 131 072 cramped_room envs, alternating $LIBS
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-ab7}
mkdir -p $O
cd $R
export TMPDIR=/tmp
for rep in 1 2; do
for lib in ${LIBS}; do
  tag=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  timeout 300 python3 bench.py --envs 1048576 --steps 1 --warmup 1 --launches-per-step 20 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/${tag}_1M_$rep.json 2>> $O/err.log
  timeout 300 python3 bench.py --envs 131072 --steps 1 --warmup 1 --launches-per-step 100 --no-extras --no-cpu-baseline --no-traffic --parity-steps 800 > $O/${tag}_128k_$rep.json 2>> $O/err.log
done
done
unset OC_AMD_LIB
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_[12].json")):
    try:
        d = json.load(open(f))
        print("%-36s %7.1f G  frac %.3f  launch_ms %.4f parity %s" % (os.path.basename(f), d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], (d.get("parity_check") or {}).get("mismatches")))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -3 $O/err.log 2>/dev/null
