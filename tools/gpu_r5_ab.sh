#!/bin/bash
# round 5: same-box A/B of library builds ($LIBS, default: the tree's build against gpurun_scratch/lib_prev.so) on the legs the
# mover / interact kernels serve; parity tests of the tree's build first
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5ab}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "${TESTS:-1}" = "1" ]; then
timeout 900 python3 -m pytest tests/test_gpu_launch_shapes.py -x -q -m gpu -k "${TESTK:-mover or tiled}" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
fi
run() {  # tag, args...
  tag=$1; shift
  timeout 300 python3 bench.py --steps ${STEPS:-4} --warmup 1 --no-extras --no-cpu-baseline --no-traffic "$@" > $O/$tag.json 2>> $O/err.log
}
for rep in $(seq 1 ${REPS:-2}); do
for lib in ${LIBS:-overcooked_ai_amd/liboc_amd.so gpurun_scratch/lib_prev.so}; do
  t=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  run ${t}_head_$rep
  run ${t}_mix_$rep --config 4
  for lay in ${LAYOUTS:-asymmetric_advantages counter_circuit}; do
    run ${t}_${lay}_$rep --layout $lay
  done
  if [ "${GEN:-0}" = "1" ]; then run ${t}_gen65536_$rep --config 5 --envs 65536; fi
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
