"""Per-wave summary of SQ PMC counters for kernels matching a substring:  python tools/pmc_summary.py <dir>... -- <kernel substr>..."""
import glob
import sqlite3
import sys

args = sys.argv[1:]
json_out = steps = None
if "--json" in args:  # --json <path> <steps per launch>: per-env-step counters of the first kernel -> profiles/sq_counters.json
    i = args.index("--json")
    json_out, steps = args[i + 1], float(args[i + 2])
    del args[i:i + 3]
dirs, kerns = args[:args.index("--")], args[args.index("--") + 1:]
for kern in kerns:
    vals = {}
    for d in dirs:
        for f in glob.glob(d + "/*.db"):
            db = sqlite3.connect(f)
            for name, cnt, mean in db.execute("select counter_name,count(*),avg(counter_value) from pmc_events where name like ? group by counter_name", ("%" + kern + "%",)):
                vals[name] = mean
    print(kern, {k: int(v) for k, v in sorted(vals.items())})
    if json_out and "SQ_WAVES" in vals:
        import json

        w, wc = vals["SQ_WAVES"], vals["SQ_WAVE_CYCLES"]
        names = set()
        for d in dirs:
            for f in glob.glob(d + "/*.db"):
                names |= {r[0] for r in sqlite3.connect(f).execute("select distinct name from pmc_events where name like ?", ("%" + kern + "%",))}
        import os

        # per env-step of a 64-env group: one wavefront per group in the one-wavefront kernels, a mover + an interact wavefront in
        # k_rollout5 (their counters add up)
        # (the database holds one row per counter instance — XCD x shader engine —, so only RATIOS of its means are meaningful:
        #  per wavefront = counter / SQ_WAVES; wavefronts per group from the kernel's MODE template argument)
        kname = sorted(n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] for n in names)[0] if names else kern
        targs = kname.split("<", 1)[1].split(",") if "<" in kname else []
        wpg = 2 if kname.startswith("k_rollout5") else 1  # (k_rollout5: a mover + an interact wavefront per 64 envs)
        per = lambda k: round(vals.get(k, 0.0) / w / steps * wpg, 2)

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from overcooked_ai_amd import build

        json.dump({
            "kernel_source_sha": build.source_hash(),
            "_note": "rocprofv3 --pmc SQ_* passes of tools/prof_rollout.py (tools/pmc_rollout.sh), 65 536 cramped_room envs, %d fused "
                     "steps per launch; per 64-env group and env-step — with k_rollout5 the mover's and the interact wavefront's instructions "
                     "together (wavefronts_per_64_envs = 2) — the launch prologue (e.g. the joint move table build) included; "
                     "wave_clk_per_env_step = clocks of ONE wavefront per step (mean over movers and interact wavefronts)" % steps,
            "kernel": sorted(n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] for n in names)[0] if names else kern,
            "valu_per_env_step": per("SQ_INSTS_VALU"), "salu_per_env_step": per("SQ_INSTS_SALU"),
            "lds_per_env_step": per("SQ_INSTS_LDS"), "vmem_wr_per_env_step": per("SQ_INSTS_VMEM_WR"),
            "branches_per_env_step": per("SQ_INSTS_BRANCH"), "wavefronts_per_64_envs": wpg,
            "wave_clk_per_env_step": round(4 * wc / w / steps, 1),
            "valu_busy_frac": round(vals.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 3),
            "wait_any_frac": round(vals.get("SQ_WAIT_ANY", 0.0) / wc, 3),
            "lds_bank_conflict_per_active_cycle": round(vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, vals.get("SQ_LDS_IDX_ACTIVE", 0.0)), 2),
        }, open(json_out, "w"), indent=1)
        json_out = None
    if "SQ_WAVES" in vals:
        w, wc = vals["SQ_WAVES"], vals["SQ_WAVE_CYCLES"]
        g = lambda k: vals.get(k, 0.0)
        print("  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_WR %.1f VMEM_RD %.1f | wave quad-cycles %.0f (= %.0f clk) | wait_any %.0f%% wait_inst %.0f%% active %.0f%% | lds conflict/active %.2f"
              % (g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_LDS") / w, g("SQ_INSTS_VMEM_WR") / w, g("SQ_INSTS_VMEM_RD") / w,
                 wc / w, 4 * wc / w, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc,
                 g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))))
