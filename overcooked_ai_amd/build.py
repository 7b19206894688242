"""Build liboc_amd.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.

    python -m overcooked_ai_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is written next to this package so that it travels with
a source snapshot (and is git-ignored)."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG, "csrc", "oc_amd.hip")
HDR = os.path.join(os.path.dirname(PKG), "include", "oc_amd.h")
LIB = os.path.join(PKG, "liboc_amd.so")
ARCH = "gfx950"
# The step kernels run one wavefront per SIMD at the headline batch size, so latency has to be hidden inside the
# wavefront: LLVM's max-ILP scheduling strategy (instead of the default, which schedules for occupancy) is worth
# +14 % on k_rollout3 and is neutral for the bandwidth-bound kernels.
# Scheduling the pre-RA list top-down (default: bidirectional) is worth another +1.3 % on k_rollout4 at 65 536 envs and
# +4 % on its per-env-terrain instances (BASELINE configs[3] / [4]); post-RA scheduling off, bottom-up or no memop
# clustering measured equal or worse.
SCHED = ("-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-misched-prera-direction=topdown")


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and the C-ABI header: profiles that bench.py replays
    (profiles/traffic.json, profiles/sq_counters.json) carry it, and are dropped when it no longer matches."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.dirname(SRC)
    for p in sorted(os.listdir(csrc)) + [HDR]:
        path = p if os.path.isabs(p) else os.path.join(csrc, p)
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    csrc = os.path.dirname(SRC)
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [HDR]
    return any(os.path.getmtime(p) > t for p in deps if os.path.exists(p))


def build_extension(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    tmp = LIB + ".%d.tmp" % os.getpid()
    cmd = [hipcc_path(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", *SCHED, "-shared", "-fPIC", "-o", tmp, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
