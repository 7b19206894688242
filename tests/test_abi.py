"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/oc_amd.h declares.
No compute calls here (those need a GPU: tests/test_gpu_parity.py)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "oc_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(oc_[a-z_0-9]+)\s*\(", hdr)))


def test_library_builds_and_exports_every_declared_symbol():
    from overcooked_ai_amd import _lib, build

    path = build.build_extension()
    assert os.path.exists(path)
    L = _lib.load()
    syms = _declared_symbols()
    assert set(syms) == set(_lib.EXPORTS), (syms, _lib.EXPORTS)
    for s in syms:
        assert getattr(L, s) is not None
    assert L.oc_abi_version() == 6
    assert L.oc_layout_size() == 256
    assert L.oc_state_planes(5, 4) == 3 and L.oc_state_planes(9, 5) == 4 and L.oc_state_planes(14, 9) == 9


def test_batch_struct_layout_matches_header():
    from overcooked_ai_amd import _lib

    # OcBatch: two pointers, int64, four int32, two uint32 = 48 bytes on LP64
    assert ctypes.sizeof(_lib.OcBatch) == 48 and _lib.OcBatch.batch_flags.offset == 40
    assert _lib.OcBatch.n_envs.offset == 16 and _lib.OcBatch.n_layouts.offset == 24
    assert _lib.OcBatch.width.offset == 28 and _lib.OcBatch.height.offset == 32 and _lib.OcBatch.max_pots.offset == 36


def test_batch_hints_from_a_host_table():
    """oc_batch_hints derives the kernel-variant hints from a HOST copy of the layout table (no GPU involved)."""
    import numpy as np

    from overcooked_ai_amd import _lib
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    L = _lib.load()
    from overcooked_ai_amd.layouts import LayoutSpec

    other_shaping = dict(spec_from_name("cramped_room").to_layout_dict(),
                         rew_shaping_params={"PLACEMENT_IN_POT_REW": 1, "DISH_PICKUP_REWARD": 3, "SOUP_PICKUP_REWARD": 5})
    # flags: two players everywhere (1) | new dynamics (2) | one set of shaping rewards for the whole table (4) | no non-floor
    # cell touches two floor cells (8: true for cramped_room, not for asymmetric_advantages)
    for names, pots, free, flags in ((["cramped_room"], 1, 6, 15), (["asymmetric_advantages", "cramped_room"], 2, None, 7),
                                     (["cramped_room", LayoutSpec(other_shaping)], 1, 6, 11)):
        table = LayoutTable([spec_from_name(nm) if isinstance(nm, str) else nm for nm in names],
                            pad_to=(9, 5) if len(names) > 1 else None)
        rec = np.ascontiguousarray(table.records)
        b = _lib.OcBatch()
        assert L.oc_batch_hints(rec.ctypes.data, len(table), ctypes.byref(b)) == 0
        want_free = max(sum(row.count(" ") for row in s.terrain_mtx) for s in table.specs)
        assert b.max_pots == pots == table.max_pots and b.batch_flags == flags and b.max_free_cells == want_free
        if free is not None:
            assert b.max_free_cells == free
    assert ctypes.sizeof(_lib.OcStartSpec) == 40 and _lib.OcStartSpec.rnd_obj_prob_thresh.offset == 24
    assert _lib.OcStartSpec.regen_first.offset == 32 and _lib.OcStartSpec.regen_count.offset == 36
    assert L.oc_batch_hints(None, 1, None) == -1


def test_argument_validation_without_gpu():
    """Argument errors are detected on the host before any launch, so they are testable without a GPU."""
    from overcooked_ai_amd import _lib

    L = _lib.load()
    assert L.oc_step(None, None, None, None, None, None, None, None, 400, 0, None, None, None) == -1
    assert b"batch is NULL" in L.oc_last_error()
    b = _lib.OcBatch(d_layouts=None, d_layout_id=None, n_envs=4, n_layouts=1, width=5, height=4)
    assert L.oc_reset(ctypes.byref(b), None, None, None, None) == -1
    assert b"d_layouts" in L.oc_last_error()
    b = _lib.OcBatch(d_layouts=4096, d_layout_id=None, n_envs=4, n_layouts=1, width=40, height=40)
    assert L.oc_reset(ctypes.byref(b), 4096, None, None, None) == -1
    assert b"grid shape" in L.oc_last_error()
    # oc_rollout_encode / oc_step_encode: pointer, alignment, option and range checks come before any launch
    b = _lib.OcBatch(d_layouts=4096, d_layout_id=None, n_envs=4, n_layouts=1, width=5, height=4)
    br = ctypes.byref(b)
    assert L.oc_rollout_encode(br, 4096, None, None, None, None, None, 0, 0, 400, 1, 0, 0, 0, 3, None, None) == -1
    assert b"NULL state / observation" in L.oc_last_error()
    assert L.oc_rollout_encode(br, 4096, None, None, None, None, 4096, 0, 24, 400, 1, 0, 0, 0, 3, None, None) == -1
    assert b"multiples of 16" in L.oc_last_error()
    assert L.oc_rollout_encode(br, 4096, None, None, None, None, 4096, 7, 0, 400, 1, 0, 0, 0, 3, None, None) == -1
    assert b"obs_dtype" in L.oc_last_error()
    assert L.oc_rollout_encode(br, 4096, None, None, None, None, 4096, 0, 0, 400, 0x8, 0, 0, 0, 3, None, None) == -1
    assert b"options" in L.oc_last_error()
    assert L.oc_rollout_encode(br, 4096, None, None, None, None, 4096, 0, 0, 70000, 1, 0, 0, 0, 3, None, None) == -1
    assert b"horizon" in L.oc_last_error()
    assert L.oc_rollout_encode(br, 4096, 4096, None, None, None, 4096, 0, 0, 400, 1, 0, 0, 0, 3, None, None) == -1
    assert b"caller actions need" in L.oc_last_error()
    assert L.oc_rollout_encode(br, 4096, None, None, None, None, 4096, 0, 0, 400, 1, 0, 0, 0, 0, None, None) == 0  # no steps: nothing to do
    assert L.oc_step_encode(br, 4096, 4096, 4096, 4096, None, None, 0, 400, 1, None, None) == -1
    assert b"NULL pointer" in L.oc_last_error()
    # OC_OPT_FLAGS_TILED8 (0x40): the launch-shape conditions are checked before anything is launched
    assert L.oc_rollout_random(br, 4096, 4096, 4096, None, 400, 0x41, 0, 0, 0, 12, None, None, None) == -1
    assert b"multiples of 8" in L.oc_last_error()
    assert L.oc_rollout_random(br, 4096, 4096, 4096, None, 400, 0x41, 0, 0, 4, 16, None, None, None) == -1
    assert b"multiples of 8" in L.oc_last_error()
    assert L.oc_rollout_random(br, 4096, 4096, 4100, None, 400, 0x41, 0, 0, 0, 16, None, None, None) == -1
    assert b"8-byte aligned" in L.oc_last_error()
    assert L.oc_rollout_random(br, 4096, None, 4096, None, 400, 0x41, 0, 0, 0, 16, None, None, None) == -1
    assert b"needs d_rewards" in L.oc_last_error()
    assert L.oc_rollout_random(br, 4096, 4096, 4096, None, 400, 0x45, 0, 0, 0, 16, None, None, None) == -1
    assert b"default kernel" in L.oc_last_error()
    # the measurement aid: argument checks before the launch, nothing to do for an empty job
    assert L.oc_output_stores_only(64, 8, None, None, 0, None) == -1 and b"no rewards array" in L.oc_last_error()
    assert L.oc_output_stores_only(64, 8, 4100, None, 0, None) == -1 and b"16-byte aligned" in L.oc_last_error()
    assert L.oc_output_stores_only(0, 8, 4096, None, 0, None) == 0 and L.oc_output_stores_only(64, 0, 4096, 4096, 0, None) == 0
    assert L.oc_output_stores_only(64, 8, 4096, 4096, 0x1, None) == -1 and b"only option" in L.oc_last_error()
    assert L.oc_output_stores_only(64, 12, 4096, 4096, 0x40, None) == -1 and b"multiple of 8" in L.oc_last_error()
    assert L.oc_output_stores_only(64, 8, 4096, None, 0x40, None) == -1 and L.oc_output_stores_only(0, 8, 4096, 4096, 0x40, None) == 0


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under overcooked_ai_amd/ may reference it."""
    pkg = os.path.join(ROOT, "overcooked_ai_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text, f


def test_vec_env_fails_loudly_without_gpu():
    import pytest
    import torch

    from overcooked_ai_amd import _lib
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.OcAmdError):
        VecOvercookedEnv("cramped_room", 4, device="cpu")


def _code_objects(tmp_path):
    """The gfx950 code objects inside the built library, one per translation unit (llvm-objcopy dumps .hip_fatbin — the
    units' offload bundles back to back —, clang-offload-bundler unbundles each)."""
    import subprocess

    from overcooked_ai_amd import _lib

    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools) or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("ROCm llvm tools or the built library not available")
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call([tools[0], "--dump-section", ".hip_fatbin=" + fat, _lib.LIB_PATH])
    raw = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [i for i in range(len(raw)) if raw.startswith(magic, i)]
    assert starts, "no offload bundle in .hip_fatbin"
    cos = []
    for k, a in enumerate(starts):
        part, co = str(tmp_path / ("fat%d.bin" % k)), str(tmp_path / ("co%d.elf" % k))
        open(part, "wb").write(raw[a:starts[k + 1] if k + 1 < len(starts) else len(raw)])
        subprocess.check_call([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + part, "--output=" + co])
        assert os.path.getsize(co) > 0
        cos.append(co)
    return cos, tools[2]


def test_no_kernel_keeps_a_stack_object_or_reads_the_dispatch_packet(tmp_path):
    """Two silent performance traps, checked on the built code object (docs/NOTEBOOK.md 4.2c / 4.4): a private segment (scratch) in a
    step / rollout kernel — a stack object that only the restart path touches still makes the step loop wait, through vmcnt,
    for its output stores: 272 instead of 300+ G env-steps/s on the headline — and the dispatch-packet pointer among a kernel's
    user SGPRs (a private array the compiler parks in LDS makes it read the workgroup size from host memory: ~10 us per
    launch).  Only the general many-pot k_potential may keep scratch."""
    import struct
    import subprocess

    cos, readelf = _code_objects(tmp_path)
    seen = checked = 0
    for co in cos:
        notes = subprocess.check_output([readelf, "--notes", co], text=True)
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:") and "_Z" in line:
                name = line.split(":", 1)[1].strip()
            elif line.startswith(".private_segment_fixed_size:") and name:
                size = int(line.split(":", 1)[1])
                seen += 1
                assert size == 0 or "11k_potentialE" in name, "%s keeps %d bytes of scratch" % (name, size)
        # kernel descriptors (<kernel>.kd, 64 bytes): kernel_code_properties at byte 56, bit 1 = ENABLE_SGPR_DISPATCH_PTR, bit 2 = QUEUE_PTR
        # (checked against a probe kernel known to read the packet: tools/launch_floor.hip)
        raw = open(co, "rb").read()
        shoff, shentsize, shnum = struct.unpack_from("<Q", raw, 0x28)[0], *struct.unpack_from("<HH", raw, 0x3A)
        sections = [struct.unpack_from("<IIQQQQIIQQ", raw, shoff + i * shentsize) for i in range(shnum)]
        symtab = next(s for s in sections if s[1] == 2)  # SHT_SYMTAB
        strtab = sections[symtab[6]]
        for i in range(symtab[5] // 24):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", raw, symtab[4] + 24 * i)
            end = raw.index(b"\0", strtab[4] + st_name)
            sym = raw[strtab[4] + st_name:end].decode()
            if not sym.endswith(".kd") or st_shndx == 0 or st_shndx >= shnum:
                continue
            sec = sections[st_shndx]
            kd = raw[sec[4] + (st_value - sec[3]):][:64]
            props = struct.unpack_from("<H", kd, 56)[0]
            assert not (props & 0x6), "%s reads the dispatch packet / queue (kernel_code_properties %#x)" % (sym, props)
            checked += 1
    assert len(cos) == 4 and seen > 80 and checked > 80  # four translation units (oc_amd.hip + rollout4.hip x 3)


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md (the binding a maintainer of the reference would write) mentions every exported function."""
    from overcooked_ai_amd import _lib

    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        doc = f.read()
    names = set(re.findall(r"\boc_[a-z_0-9]+\b", doc))
    for prefix, tails in re.findall(r"`(oc_[a-z_]+?)_[a-z]+ ((?:/ _[a-z_]+ ?)+)", doc):  # "`oc_mailbox_open / _buffer / _step`"
        names |= {prefix + t.strip() for t in tails.split("/") if t.strip()}
    missing = [e for e in _lib.EXPORTS if e not in names]
    assert not missing, missing
