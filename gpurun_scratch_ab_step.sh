set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r5ntstep; mkdir -p $O
run() { tag=$1; shift; timeout 300 python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-traffic "$@" > $O/$tag.json 2>> $O/err.log; }
for rep in 1 2; do for lib in liboc_amd prev; do
  export OC_AMD_LIB=$R/overcooked_ai_amd/$lib.so
  run ${lib}_crstep_$rep --flags-layout step
  run ${lib}_mixstep_$rep --config 4 --flags-layout step
  run ${lib}_cr1M_$rep --envs 1048576 --launches-per-step 12 --steps 2 --no-parity-check
  run ${lib}_cr_$rep
done; done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    d=json.load(open(f)); r=d["roofline"]
    print("%-26s %7.1f G  frac %.3f  launch %.3f ms  parity %s  flags %s" % (os.path.basename(f)[:-5], d["value"]/1e9, r["frac"], r["launch_ms"], (d.get("parity_check") or {}).get("mismatches"), d["config"].get("flags_layout","")[:16]))
PY
