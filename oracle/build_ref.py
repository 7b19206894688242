"""Build the reference itself for the CPU-baseline leg:  python -m oracle.build_ref

TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by the product package).

north_star asks for "the reference Python OvercookedEnv.step timed on the same box's host cores".  /root/reference
does not exist on the GPU box and its SOURCES are never copied into this repo.  What a compiled reference gets — its
own few source files compiled where they lie, outputs only into oracle/_ref/ (git-ignored, travels with the snapshot
like liboc_amd.so) — a Python reference gets too: CPython's compiler (`py_compile`) turns the modules of the hot path
into sourceless byte-code files `oracle/_ref/src/overcooked_ai_py/**/<module>.pyc`.  No .py, no data file, nothing of
the reference's build system; layouts are handed to `OvercookedGridworld.from_grid` from this repo's own layout data
(tools/time_reference_python.py).  The byte code only loads under the CPython minor version that wrote it (the build
container and the GPU box run the same image); `bench.py` falls back to the stored profile otherwise and says so.

Modules compiled (the import closure of `OvercookedEnv.step`, src/overcooked_ai_py/mdp/overcooked_env.py:244; the
package __init__, gymnasium / cv2 / pygame and the sprite-loading visualizer are stubbed by oracle/ref_harness.py):
"""
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "src")
SRC = "/root/reference/src"
MODULES = [
    "overcooked_ai_py/__init__.py",
    "overcooked_ai_py/static.py",
    "overcooked_ai_py/utils.py",
    "overcooked_ai_py/mdp/__init__.py",
    "overcooked_ai_py/mdp/actions.py",
    "overcooked_ai_py/mdp/overcooked_mdp.py",
    "overcooked_ai_py/mdp/overcooked_env.py",
    "overcooked_ai_py/mdp/overcooked_trajectory.py",
    "overcooked_ai_py/planning/__init__.py",
    "overcooked_ai_py/planning/planners.py",
    "overcooked_ai_py/planning/search.py",
    "overcooked_ai_py/data/__init__.py",
    "overcooked_ai_py/data/planners/__init__.py",
    "overcooked_ai_py/visualization/__init__.py",
]


def available():
    return os.path.isdir(os.path.join(SRC, "overcooked_ai_py"))


def built():
    return os.path.exists(os.path.join(OUT, "overcooked_ai_py", "mdp", "overcooked_env.pyc"))


def build(force=False):
    """Compile MODULES into oracle/_ref/src (sourceless .pyc).  Returns the output directory, or None when
    /root/reference is absent (the GPU box: only what was built in the container is used)."""
    if not available():
        return OUT if built() else None
    stamp = os.path.join(OUT, "PYTHON_VERSION")
    ver = "%d.%d.%d" % sys.version_info[:3]
    if built() and not force and os.path.exists(stamp) and open(stamp).read().strip() == ver:
        newest = max(os.path.getmtime(os.path.join(SRC, m)) for m in MODULES if os.path.exists(os.path.join(SRC, m)))
        if newest <= os.path.getmtime(stamp):
            return OUT
    shutil.rmtree(OUT, ignore_errors=True)
    for m in MODULES:
        src = os.path.join(SRC, m)
        dst = os.path.join(OUT, m[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if os.path.exists(src):
            py_compile.compile(src, cfile=dst, doraise=True, optimize=0)
        else:  # a namespace directory upstream: an empty package here
            py_compile.compile(os.devnull, cfile=dst, doraise=True, optimize=0)
    with open(stamp, "w") as f:
        f.write(ver + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
