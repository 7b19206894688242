#!/bin/bash
# round 5: the rollout by batch size with the shipped dispatch (mover / interact workgroups in rounds of one per CU above 65 536 envs)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05sweep
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -x -q -m gpu -k "tiled or launch_shape or xcd" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
run() { tag=$1; shift; timeout 300 python3 bench.py --steps 2 --warmup 1 --launches-per-step 30 --no-extras --no-cpu-baseline --no-traffic --no-parity-check "$@" > $O/$tag.json 2>> $O/err.log; }
for n in 32768 65536 98304 131072 196608 262144 524288 1048576; do run cr_$n --envs $n; done
for n in 65536 131072 262144; do run mix_$n --config 4 --envs $n; run gen_$n --config 5 --envs $n; run aa_$n --layout asymmetric_advantages --envs $n; done
python3 - <<PY
import json, glob, os
print("rollout by batch size, shipped dispatch, one box: G env-steps/s, frac of the 8 TB/s roofline, launch, flags layout served")
for pre in ("cr","mix","gen","aa"):
    for f in sorted(glob.glob("$O/%s_*.json" % pre), key=lambda p: int(p.split("_")[-1][:-5])):
        try:
            d=json.load(open(f)); r=d["roofline"]
            print("%-14s %7.1f G  frac %.3f  launch %.3f ms  flags %s" % (os.path.basename(f)[:-5], d["value"]/1e9, r["frac"], r["launch_ms"], d["config"].get("flags_layout","")[:22]))
        except Exception as e: print(os.path.basename(f), "ERR", e)
PY
grep -v amdgpu.ids $O/err.log | tail -3
