#!/bin/bash
# the driver's own command (sustained headline region + the BASELINE configs as side legs), alternating between $LIBS on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-ab6}
mkdir -p $O
cd $R
export TMPDIR=/tmp
for rep in 1 2; do
for lib in ${LIBS}; do
  tag=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/${tag}_drv_$rep.json 2>> $O/err.log
done
done
unset OC_AMD_LIB
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_drv_[12].json")):
    try:
        d = json.load(open(f))
        c = d.get("configs") or {}
        print("%-28s %7.1f G frac %.3f parity %s | c3 %.3f G (%.3f) c4 %.1f G (%.3f) c5 %.1f G (%.3f) | parities %s" % (
            os.path.basename(f), d["value"] / 1e9, d["roofline"]["frac"], (d.get("parity_check") or {}).get("mismatches"),
            c["3"]["value"] / 1e9, c["3"]["roofline"]["frac"], c["4"]["value"] / 1e9, c["4"]["roofline"]["frac"],
            c["5"]["value"] / 1e9, c["5"]["roofline"]["frac"], [(c[k].get("parity_check") or {}).get("mismatches") for k in "345"]))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -3 $O/err.log 2>/dev/null
