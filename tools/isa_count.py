#!/usr/bin/env python
"""Static instruction counts of a kernel's hot loop in a built library (no GPU needed).

    python tools/isa_count.py [path/to/liboc_amd.so] 'k_rollout4<true, 1, true, 1, true, false, 6, false, true, false, 4, true>'

Extracts the gfx950 code object (llvm-objcopy + clang-offload-bundler), disassembles it, finds the kernel whose demangled
name contains the given text, and prints — for every loop (backward branch) of at least 300 instructions — the number of
VALU / SALU / LDS / VMEM / waitcnt instructions in its body.  The unrolled 8-step block of a rollout kernel is the loop whose
body holds eight `global_store_dwordx4` (the reward quads): counts / 8 = per env-step."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(lib):
    d = tempfile.mkdtemp(prefix="isa_")
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib])
    raw = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [i for i in range(len(raw)) if raw.startswith(magic, i)]  # one bundle per translation unit
    out = []
    for k, a in enumerate(starts):
        part, co = os.path.join(d, "fat%d.bin" % k), os.path.join(d, "co%d.elf" % k)
        open(part, "wb").write(raw[a:starts[k + 1] if k + 1 < len(starts) else len(raw)])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + part, "--output=" + co])
        out += subprocess.check_output([LLVM + "/llvm-objdump", "-d", co], text=True, stderr=subprocess.DEVNULL).splitlines()
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "overcooked_ai_amd", "liboc_amd.so")
    want = sys.argv[-1].replace(" ", "")
    lines = disassemble(lib)
    heads = [(i, l) for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l)]
    names = subprocess.run(["c++filt"], input="\n".join(l.split("<", 1)[1].rsplit(">", 1)[0] for _, l in heads), text=True,
                           capture_output=True).stdout.splitlines()
    for (i, _), nm in zip(heads, names):
        if want in nm.replace(" ", "").replace("(anonymousnamespace)::", ""):
            start = i
            end = next((j for j, _ in heads if j > i), len(lines))
            break
    else:
        raise SystemExit("kernel not found: " + want)
    body = lines[start:end]
    addr = {}
    for k, l in enumerate(body):
        m = re.search(r"//\s*([0-9A-F]{12}):", l)
        if m:
            addr[int(m.group(1), 16)] = k
    print(nm[:160])
    print("kernel: %d instructions" % len(addr))
    for k, l in enumerate(body):
        m = re.match(r"\s*(s_cbranch\w+|s_branch)\s+(\d+)", l)
        if not m:
            continue
        pc = int(re.search(r"//\s*([0-9A-F]{12}):", l).group(1), 16)
        off = int(m.group(2))
        off -= 65536 if off >= 32768 else 0
        tgt = addr.get(pc + 4 + 4 * off)
        if tgt is None or tgt >= k or k - tgt < 300:
            continue
        ins = [x.split()[0] for x in body[tgt:k + 1] if x.strip() and not x.strip().startswith("//")]
        c = lambda p: sum(1 for x in ins if re.match(p, x))
        stores = c(r"global_store_dwordx4")
        print("loop %6d..%6d: %5d instr | VALU %4d SALU %4d LDS %3d VMEM %3d waitcnt %3d | quad stores %d%s"
              % (tgt, k, len(ins), c(r"v_"), c(r"s_(?!waitcnt|nop)"), c(r"ds_"), c(r"global_|buffer_|scratch_"), c(r"s_waitcnt"),
                 stores, ("  -> per step: %.1f VALU %.1f SALU %.1f LDS" % (c(r"v_") / 8, c(r"s_(?!waitcnt|nop)") / 8, c(r"ds_") / 8)) if stores == 8 else ""))


if __name__ == "__main__":
    main()
