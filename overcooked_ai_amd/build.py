"""Build liboc_amd.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.

    python -m overcooked_ai_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is written next to this package so that it travels with
a source snapshot (and is git-ignored)."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG, "csrc", "oc_amd.hip")  # (+ rollout4.hip: UNITS below)
HDR = os.path.join(os.path.dirname(PKG), "include", "oc_amd.h")
LIB = os.path.join(PKG, "liboc_amd.so")
ARCH = "gfx950"
# The step kernels run one wavefront per SIMD at the headline batch size, so latency has to be hidden inside the
# wavefront: LLVM's max-ILP scheduling strategy (instead of the default, which schedules for occupancy) is worth
# +14 % on the round-1 rollout kernel and is neutral for the bandwidth-bound kernels.
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


UNITS = (("oc_amd.hip", ()), ("rollout4.hip", ("-DOC_R4_PART=0",)), ("rollout4.hip", ("-DOC_R4_PART=1",)),
         ("rollout4.hip", ("-DOC_R4_PART=2",)))


def build_extension(force=False, verbose=False, defines=(), out=None):
    """Compile the translation units side by side (one hipcc process each) and link them into liboc_amd.so.
    defines: extra -D flags (e.g. ("-DOC_AMD_TUNING",) for the measurement scripts' environment knobs); out: another
    output path (variant libraries for A/B runs: OC_AMD_LIB selects one at load time)."""
    lib = out or LIB
    if not force and out is None and not is_stale():
        return LIB
    objdir = os.path.join(PKG, "csrc", "_obj", "%d" % os.getpid())
    os.makedirs(objdir, exist_ok=True)
    csrc = os.path.dirname(SRC)
    base = [hipcc_path(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", *SCHED, "-fPIC", *defines]
    procs, objs = [], []
    for i, (src, flags) in enumerate(UNITS):
        obj = os.path.join(objdir, "u%d.o" % i)
        cmd = base + list(flags) + ["-c", "-o", obj, os.path.join(csrc, src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    tmp = lib + ".%d.tmp" % os.getpid()
    subprocess.check_call([hipcc_path(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs)
    os.replace(tmp, lib)
    for o in objs:
        os.remove(o)
    os.rmdir(objdir)
    return lib


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True,
                          defines=tuple(a for a in sys.argv[1:] if a.startswith("-D"))))
