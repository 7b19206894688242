#!/bin/bash
# round 5: same-box A/B of library builds ($LIBS) on every bandwidth leg: headline, configs[2..4], the encode / featurize / training
# legs of the default line (bench.py extras), each with its parity check
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5xcd}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ -n "${TESTK:-}" ]; then
timeout 900 python3 -m pytest tests -x -q -m gpu -k "$TESTK" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
fi
for rep in $(seq 1 ${REPS:-2}); do
for lib in ${LIBS}; do
  t=$(basename $lib .so)
  export OC_AMD_LIB=$R/$lib
  timeout 400 python3 bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-traffic > $O/${t}_$rep.json 2>> $O/err.log
  if [ "${GEN:-0}" = "1" ]; then timeout 300 python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --config 5 --envs 65536 > $O/${t}_gen65536_$rep.json 2>> $O/err.log; fi
done
done
unset OC_AMD_LIB
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
        row = ["%-28s" % os.path.basename(f), "head %6.1f G (%.3f) par %s" % (d["value"] / 1e9, d["roofline"]["frac"], (d.get("parity_check") or {}).get("mismatches"))]
        so = d["roofline"].get("store_only") or {}
        if so: row.append("store_only %.1f G" % (so.get("env_steps_per_s", 0) / 1e9))
        for k, v in (d.get("configs") or {}).items():
            row.append("c%s %.4g (%.3f) par %s" % (k, v.get("value", 0) / 1e9, (v.get("roofline") or {}).get("frac", 0), (v.get("parity_check") or {}).get("mismatches")))
        e = d.get("encode") or {}
        if e:
            row.append("enc u8 %.3f f32 %.3f feat %.3f" % (e["u8"]["frac"], e["f32"]["frac"], e["featurize_state"]["frac"]))
        t = d.get("training_env") or {}
        if t: row.append("train u8 %.1f us f32 %.1f us" % (t["obs_u8"]["us_per_batched_step"], t["obs_f32"]["us_per_batched_step"]))
        sm = (d.get("step_api") or {}).get("step_many")
        if sm: row.append("step_many %.3f" % sm["frac"])
        print("  ".join(row))
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex)
PY
grep -v amdgpu.ids $O/err.log 2>/dev/null | tail -5
