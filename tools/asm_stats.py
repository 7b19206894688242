"""Instruction histogram of one kernel from hipcc -S output:  python tools/asm_stats.py file.s <substring-of-mangled-name>"""
import collections
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].rstrip().endswith(":"):
        start = i
        break
assert start is not None, "kernel not found"
ins = []
for l in lines[start + 1:]:
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
        break
    t = l.strip()
    if not l.startswith("\t") or not t or t[0] in ".;":
        continue
    ins.append(t.split()[0])
c = collections.Counter(ins)
print("total instructions:", len(ins))
cats = collections.Counter()
for k, v in c.items():
    cat = "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else \
        "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
    cats[cat] += v
print(dict(cats))
for k, v in c.most_common(40):
    print("  %-28s %d" % (k, v))
