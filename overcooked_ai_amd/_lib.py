"""ctypes binding of liboc_amd.so (include/oc_amd.h).  There is no CPU fallback: if the HIP library is
missing or fails to load, every entry point raises."""
import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OC_AMD_LIB") or os.path.join(PKG, "liboc_amd.so")  # OC_AMD_LIB: developer override

ABI_VERSION = 6
F_DONE, F_BAD_ACTION, F_RESET = 0x01, 0x02, 0x04
OPT_AUTO_RESET = 0x1
OPT_LANE_PAIR = 0x4
OPT_PREDICATE_INTERACT = 0x8
OPT_ONE_KERNEL = 0x20
OPT_FLAGS_TILED8 = 0x40
OPT_ONE_WAVEFRONT = 0x80
BATCH_TWO_PLAYERS = 0x1
BATCH_NEW_DYNAMICS = 0x2
BATCH_UNIFORM_SHAPING = 0x4
BATCH_NO_SHARED_FACES = 0x8
OBS_U8, OBS_F32 = 0, 1

EXPORTS = ("oc_abi_version", "oc_layout_size", "oc_last_error", "oc_state_planes", "oc_batch_hints", "oc_step", "oc_step_many",
           "oc_rollout_random", "oc_encode_lossless", "oc_step_encode", "oc_rollout_encode", "oc_featurize", "oc_potential",
           "oc_phi_table_size", "oc_reset", "oc_reset_random", "oc_regen_layouts", "oc_shape_rewards", "oc_multi_agent_step",
           "oc_mailbox_open", "oc_mailbox_buffer", "oc_mailbox_step", "oc_mailbox_close", "oc_output_stores_only", "oc_rollout_plan",
           "oc_step_server_open", "oc_step_server_requests", "oc_step_server_responses", "oc_step_server_resume",
           "oc_step_server_play", "oc_step_server_sync", "oc_step_server_steps", "oc_step_server_close")
MB_STATE_IN, MB_ACTIONS, MB_STATE_OUT, MB_REWARDS, MB_FLAGS, MB_EVENTS, MB_BYTES = 256, 336, 512, 592, 608, 616, 4096


class OcBatch(ctypes.Structure):
    _fields_ = [
        ("d_layouts", ctypes.c_void_p),
        ("d_layout_id", ctypes.c_void_p),
        ("n_envs", ctypes.c_int64),
        ("n_layouts", ctypes.c_int32),
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("max_pots", ctypes.c_int32),
        ("batch_flags", ctypes.c_uint32),
        ("max_free_cells", ctypes.c_uint32),
    ]


class OcStartSpec(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("env_offset", ctypes.c_int64),
        ("epoch", ctypes.c_uint32),
        ("random_start_pos", ctypes.c_int32),
        ("rnd_obj_prob_thresh", ctypes.c_double),
        ("regen_first", ctypes.c_uint32),
        ("regen_count", ctypes.c_uint32),
    ]


class OcEventSink(ctypes.Structure):
    _fields_ = [("d_events", ctypes.c_void_p), ("d_counts", ctypes.c_void_p), ("d_counts_done", ctypes.c_void_p)]


class OcAmdError(RuntimeError):
    pass


_lib = None


def load():
    """Load liboc_amd.so and declare prototypes. Raises OcAmdError when the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OcAmdError(
            "liboc_amd.so not found at %s — build it with `python -m overcooked_ai_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # torch first: the library's libamdhip64 dependency must resolve to the HIP runtime torch has loaded (its wheel bundles one),
    # or the process ends up with two runtimes and the second one finds "no ROCm-capable device" (seen when
    # __graft_entry__.build() loaded the library before smoke() imported torch)
    import torch  # noqa: F401

    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise OcAmdError("failed to load %s: %s" % (LIB_PATH, e))
    vp, i32, i64, u32, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64
    bp = ctypes.POINTER(OcBatch)
    L.oc_abi_version.restype = i32
    L.oc_abi_version.argtypes = []
    L.oc_layout_size.restype = ctypes.c_size_t
    L.oc_layout_size.argtypes = []
    L.oc_last_error.restype = ctypes.c_char_p
    L.oc_last_error.argtypes = []
    L.oc_state_planes.restype = i32
    L.oc_state_planes.argtypes = [i32, i32]
    L.oc_batch_hints.restype = i32
    L.oc_batch_hints.argtypes = [vp, i32, bp]
    L.oc_step.restype = i32
    sp = ctypes.POINTER(OcStartSpec)
    ep = ctypes.POINTER(OcEventSink)
    L.oc_step.argtypes = [bp, vp, vp, vp, vp, vp, vp, vp, i32, u32, sp, ep, vp]
    L.oc_step_many.restype = i32
    L.oc_step_many.argtypes = [bp, vp, vp, vp, vp, vp, i32, i32, u32, sp, ep, vp]
    L.oc_rollout_random.restype = i32
    L.oc_rollout_random.argtypes = [bp, vp, vp, vp, vp, i32, u32, u64, i64, i64, i32, sp, ep, vp]
    L.oc_encode_lossless.restype = i32
    L.oc_encode_lossless.argtypes = [bp, vp, vp, i32, i32, vp]
    L.oc_step_encode.restype = i32
    L.oc_step_encode.argtypes = [bp, vp, vp, vp, vp, vp, vp, i32, i32, u32, sp, vp]
    L.oc_rollout_encode.restype = i32
    L.oc_rollout_encode.argtypes = [bp, vp, vp, vp, vp, vp, vp, i32, i64, i32, u32, u64, i64, i64, i32, sp, vp]
    L.oc_featurize.restype = i32
    L.oc_featurize.argtypes = [bp, vp, vp, vp, vp, i32, vp]
    L.oc_potential.restype = i32
    L.oc_potential.argtypes = [bp, vp, vp, vp, vp, vp, vp]
    L.oc_phi_table_size.restype = i32
    L.oc_multi_agent_step.restype = i32
    L.oc_multi_agent_step.argtypes = [bp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_double, vp, vp, vp, i32, i32, sp, ep, vp]
    L.oc_shape_rewards.restype = i32
    L.oc_shape_rewards.argtypes = [bp, vp, vp, vp, vp, vp, ctypes.c_double, vp, vp, vp]
    L.oc_reset_random.restype = i32
    L.oc_reset_random.argtypes = [bp, vp, vp, vp, u64, i64, u32, i32, ctypes.c_double, vp]
    L.oc_reset.restype = i32
    L.oc_reset.argtypes = [bp, vp, vp, vp, vp]
    L.oc_regen_layouts.restype = i32
    L.oc_regen_layouts.argtypes = [bp, vp, vp, ctypes.c_uint8, sp, vp]
    L.oc_mailbox_open.restype = i32
    L.oc_mailbox_open.argtypes = [bp, i32, ctypes.POINTER(vp)]
    L.oc_mailbox_buffer.restype = vp
    L.oc_mailbox_buffer.argtypes = [vp]
    L.oc_mailbox_step.restype = i32
    L.oc_mailbox_step.argtypes = [vp]
    L.oc_mailbox_close.restype = i32
    L.oc_mailbox_close.argtypes = [vp]
    L.oc_output_stores_only.restype = i32
    L.oc_output_stores_only.argtypes = [i64, i32, vp, vp, u32, vp]
    L.oc_step_server_open.restype = i32
    L.oc_step_server_open.argtypes = [bp, vp, vp, i32, u32, sp, ctypes.c_double, ctypes.c_double, ctypes.POINTER(vp)]
    for name in ("oc_step_server_requests", "oc_step_server_responses"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    for name in ("oc_step_server_resume", "oc_step_server_sync", "oc_step_server_close"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp]
    L.oc_step_server_steps.restype = i64
    L.oc_step_server_steps.argtypes = [vp]
    L.oc_step_server_play.restype = i32
    L.oc_step_server_play.argtypes = [vp, vp, vp, vp, i32, vp, ctypes.POINTER(ctypes.c_float)]
    L.oc_rollout_plan.restype = i32
    L.oc_rollout_plan.argtypes = [bp, i32, u32, i64, i32, i32, i32, sp, ctypes.c_char_p, ctypes.c_size_t]
    if L.oc_abi_version() != ABI_VERSION:
        raise OcAmdError("liboc_amd.so ABI version %d != expected %d; rebuild" % (L.oc_abi_version(), ABI_VERSION))
    if L.oc_layout_size() != 256:
        raise OcAmdError("OcLayout size mismatch")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = load().oc_last_error().decode(errors="replace")
        raise OcAmdError("%s failed (%d): %s" % (what, rc, msg))
