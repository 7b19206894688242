"""Build overcooked_ai_amd/data/layouts.json from the reference's layout data files.

The `.layout` files of the reference (src/overcooked_ai_py/data/layouts/*.layout) are data — a grid
string plus recipe/order parameters — and are the input format of the hot path.
/root/reference does not exist on the GPU box, so the grids travel with this package as one JSON
registry.  Run in the build container only:   python tools/gen_layout_registry.py
"""
import ast
import glob
import json
import os

SRC = "/root/reference/src/overcooked_ai_py/data/layouts"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "overcooked_ai_amd", "data", "layouts.json")

out = {}
for path in sorted(glob.glob(os.path.join(SRC, "*.layout"))):
    name = os.path.basename(path)[: -len(".layout")]
    text = open(path).read().replace("float('inf')", "1e999")  # tutorial_3.layout: order_bonus = inf
    d = ast.literal_eval(text)
    d["grid"] = [row.strip() for row in d["grid"].split("\n")]
    out[name] = d
with open(DST, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("wrote %d layouts to %s" % (len(out), os.path.normpath(DST)))
