"""Variant libraries for same-box A/B runs (tools/gpu_ab5.sh, tools/gpu_ab8.sh, tools/gpu_r5_xcd.sh, tools/gpu_r5_ab2.sh): python tools/build_variants.py name=-DFLAG1,-DFLAG2 ...
builds overcooked_ai_amd/<name>.so with the given extra defines next to the default library (OC_AMD_LIB selects one)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from overcooked_ai_amd import build

for arg in sys.argv[1:]:
    name, _, flags = arg.partition("=")
    defines = tuple(f for f in flags.split(",") if f)
    print(build.build_extension(force=True, defines=defines, out=os.path.join(build.PKG, name + ".so")))
