#!/bin/bash
# round 5, call B: parity of the mover / interact kernel (MODE 3) + its rate against the one-wavefront kernels, same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_launch_shapes.py -x -q -m gpu -k "mover or tiled or config3" > $O/pytest.log 2>&1
tail -15 $O/pytest.log
run() {  # tag, args...
  tag=$1; shift
  timeout 300 python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-traffic "$@" > $O/$tag.json 2>> $O/err.log
}
for rep in 1 2; do
run mix_duo_$rep --config 4
run mix_one_$rep --config 4 --one-wavefront
done
run gen65536_duo --config 5 --envs 65536
run gen65536_one --config 5 --envs 65536 --one-wavefront
for lay in asymmetric_advantages coordination_ring forced_coordination counter_circuit; do
run ${lay}_duo --layout $lay
run ${lay}_one --layout $lay --one-wavefront
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
        print("%-40s %7.1f G  frac %.3f  launch_ms %.4f  %s parity %s" % (os.path.basename(f), d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d["config"].get("flags_layout"), (d.get("parity_check") or {}).get("mismatches")))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
tail -5 $O/err.log 2>/dev/null
