#!/bin/bash
# Does a run's steady rate follow the shader clock / socket power?  Each lib in $LIBS runs $REPS separate processes of the
# sustained headline region while rocm-smi samples clocks and power twice a second.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-clk}
mkdir -p $O
cd $R
export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-3}); do
for lib in ${LIBS}; do
  tag=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  ( while true; do rocm-smi -c -P --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.4; done ) > $O/${tag}_smi_$rep.log 2>&1 &
  SMI=$!
  timeout 300 python3 bench.py --steps ${STEPS:-10} --warmup 5 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/${tag}_run_$rep.json 2>> $O/err.log
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
done
done
unset OC_AMD_LIB
python3 - <<PY
import json, glob, os, re
for f in sorted(glob.glob("$O/*_run_*.json")):
    try:
        d = json.load(open(f))
        smi = open(f.replace("_run_", "_smi_").replace(".json", ".log")).read().splitlines()
        rows = [l for l in smi if l.strip()]
        # keep the busiest samples (largest power figure on the line)
        def nums(l): return [float(x) for x in re.findall(r"(?<![\w.])(\d+(?:\.\d+)?)", l)]
        print("%-34s %7.1f G  launch_ms %.4f | smi samples %d, last busy lines:" % (os.path.basename(f), d["value"] / 1e9, d["roofline"]["launch_ms"], len(rows)))
        for l in rows[-9:-3]: print("      ", l[:200])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
head -3 $O/*_smi_1.log | head -8
tail -3 $O/err.log 2>/dev/null
