"""Per-wave summary of SQ PMC counters for kernels matching a substring:  python tools/pmc_summary.py <dir>... -- <kernel substr>..."""
import glob
import sqlite3
import sys

args = sys.argv[1:]
dirs, kerns = args[:args.index("--")], args[args.index("--") + 1:]
for kern in kerns:
    vals = {}
    for d in dirs:
        for f in glob.glob(d + "/*.db"):
            db = sqlite3.connect(f)
            for name, cnt, mean in db.execute("select counter_name,count(*),avg(counter_value) from pmc_events where name like ? group by counter_name", ("%" + kern + "%",)):
                vals[name] = mean
    print(kern, {k: int(v) for k, v in sorted(vals.items())})
    if "SQ_WAVES" in vals:
        w, wc = vals["SQ_WAVES"], vals["SQ_WAVE_CYCLES"]
        g = lambda k: vals.get(k, 0.0)
        print("  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_WR %.1f VMEM_RD %.1f | wave quad-cycles %.0f (= %.0f clk) | wait_any %.0f%% wait_inst %.0f%% active %.0f%% | lds conflict/active %.2f"
              % (g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_LDS") / w, g("SQ_INSTS_VMEM_WR") / w, g("SQ_INSTS_VMEM_RD") / w,
                 wc / w, 4 * wc / w, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc,
                 g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))))
