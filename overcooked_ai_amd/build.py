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


_SRC_SUFFIXES = (".hip", ".hpp", ".h")


def _source_files():
    """The regular source files the library is built from (kernel sources + the C-ABI header): build debris in csrc/
    (object directories of a running or interrupted build) is not a dependency."""
    csrc = os.path.dirname(SRC)
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(_SRC_SUFFIXES)]
    return [p for p in files if os.path.isfile(p)] + [HDR]


def strip_comments(text):
    """C/C++ source without comments and with whitespace runs collapsed: what the compiler sees, near enough.  String
    and character literals are kept verbatim (a '//' inside one is not a comment)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" ")
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def source_hash():
    """sha256 (first 16 hex digits) over the comment-stripped, whitespace-normalised kernel sources and C-ABI header:
    profiles that bench.py replays (profiles/traffic.json, profiles/sq_counters.json) carry it and are dropped when it
    no longer matches.  Comment-only edits (docs in oc_amd.h) leave it unchanged, so that they do not void the stored
    profiles of a binary they did not change."""
    import hashlib

    h = hashlib.sha256()
    for path in _source_files():
        h.update(os.path.basename(path).encode())
        h.update(strip_comments(open(path, "r", encoding="utf-8", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _source_files() if os.path.exists(p))


UNITS = (("oc_amd.hip", ()), ("rollout4.hip", ("-DOC_R4_PART=0",)), ("rollout4.hip", ("-DOC_R4_PART=1",)),
         ("rollout4.hip", ("-DOC_R4_PART=2",)))


def build_extension(force=False, verbose=False, defines=(), out=None):
    """Compile the translation units side by side (one hipcc process each) and link them into liboc_amd.so.
    defines: extra -D flags (e.g. ("-DOC_AMD_TUNING",) for the measurement scripts' environment knobs); out: another
    output path (variant libraries for A/B runs: OC_AMD_LIB selects one at load time)."""
    lib = out or LIB
    if not force and out is None and not is_stale():
        return LIB
    import shutil
    import tempfile

    # objects go to a private temporary directory OUTSIDE csrc/ (round 4 kept them in csrc/_obj, whose mtime then
    # marked the library stale after every build)
    objdir = tempfile.mkdtemp(prefix="oc_amd_obj_")
    csrc = os.path.dirname(SRC)
    base = [hipcc_path(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", *SCHED, "-fPIC", *defines]
    procs, objs = [], []
    tmp = lib + ".%d.tmp" % os.getpid()
    try:
        for i, (src, flags) in enumerate(UNITS):
            obj = os.path.join(objdir, "u%d.o" % i)
            cmd = base + list(flags) + ["-c", "-o", obj, os.path.join(csrc, src)]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(obj)
        failed = None
        for cmd, p in procs:
            if p.wait() != 0 and failed is None:
                failed = (p.returncode, cmd)
        if failed:
            raise subprocess.CalledProcessError(*failed)
        subprocess.check_call([hipcc_path(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs)
        os.replace(tmp, lib)
    finally:
        for _, p in procs:  # a failed sibling: do not leave compilers running behind the exception
            if p.poll() is None:
                p.kill()
                p.wait()
        shutil.rmtree(objdir, ignore_errors=True)
        if os.path.exists(tmp):
            os.remove(tmp)
    return lib


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True,
                          defines=tuple(a for a in sys.argv[1:] if a.startswith("-D"))))
