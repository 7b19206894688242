#!/bin/bash
# round 5: quick A/B of the mover / interact kernel on one box: parity tests of the split + the legs it serves
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5c}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_launch_shapes.py -x -q -m gpu -k "mover or five_layout" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
run() {  # tag, args...
  tag=$1; shift
  timeout 300 python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-traffic "$@" > $O/$tag.json 2>> $O/err.log
}
for lib in ${LIBS:-overcooked_ai_amd/liboc_amd.so}; do
  t=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  run ${t}_mix --config 4
  run ${t}_gen65536 --config 5 --envs 65536
  for lay in asymmetric_advantages coordination_ring counter_circuit; do
    run ${t}_${lay} --layout $lay
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
