"""Where a single-env OvercookedEnv.step spends its time: python tools/time_single_env.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from overcooked_ai_amd.actions import Action
from overcooked_ai_amd.env import OvercookedEnv
from overcooked_ai_amd.mdp import OvercookedGridworld

mdp = OvercookedGridworld.from_layout_name("cramped_room", device="cuda:0")
env = OvercookedEnv.from_mdp(mdp, horizon=400, info_level=0)
rng = np.random.RandomState(0)
acts = rng.randint(0, 6, (400, 2))


def episode():
    env.reset(regen_mdp=False)
    for k in range(400):
        env.step((Action.INDEX_TO_ACTION[acts[k, 0]], Action.INDEX_TO_ACTION[acts[k, 1]]))


episode()
t0 = time.perf_counter()
for _ in range(3):
    episode()
print("OvercookedEnv.step: %.1f us per step" % ((time.perf_counter() - t0) / 1200 * 1e6))
port = mdp._port()  # (the episodes above have brought the mailbox up: it opens with the third single-state call)
state = mdp.get_standard_start_state()
N = 2000
t0 = time.perf_counter()
for _ in range(N):
    port.codec.pack(state, port.mv_in)
t1 = time.perf_counter()
for _ in range(N):
    port.codec.unpack(port.mv_in)
t2 = time.perf_counter()
t6 = time.perf_counter()
if port.mailbox is not None:  # the resident kernel: post + spin, no launch
    for _ in range(N):
        port._step(port.mailbox)
    t3 = time.perf_counter()
    if "tuning" in os.environ.get("OC_AMD_LIB", ""):  # a -DOC_AMD_TUNING build leaves the kernel's own phase times (10 ns ticks) in the spare payload dwords 27..29 of the response
        from overcooked_ai_amd import _lib
        ph = np.zeros(3)
        for _ in range(200):
            port._step(port.mailbox)
            ph += port.np[2048 + 64 + 48:2048 + 64 + 60].view(np.uint32)  # payload dwords 27..29 = words 12..14 of response line 1
        import ctypes
        port.lib.oc_mailbox_bench.restype = ctypes.c_double
        port.lib.oc_mailbox_bench.argtypes = [ctypes.c_void_p, ctypes.c_int]
        print("oc_mailbox_step called from C, back to back: %.2f us per call" % port.lib.oc_mailbox_bench(port.mailbox, 5000))
        print("inside k_mailbox: request seen -> payload loaded %.2f us, -> transition computed %.2f us; previous answer -> this request seen %.2f us"
              % tuple(ph / 200 * 0.01))
        t3 = time.perf_counter()
    for _ in range(N):
        port.transition(state, 4, 4)
    t6 = time.perf_counter()
    print("pack %.1f us, unpack %.1f us, oc_mailbox_step alone (post + spin) %.1f us, port.transition %.1f us"
          % tuple(x / N * 1e6 for x in (t1 - t0, t2 - t1, t3 - t2, t6 - t3)))
else:
    p = port.ptrs
    for _ in range(N):
        port.lib.oc_step(port.bref, p[0], p[1], p[2], p[3], p[4], None, p[5], 65535, 0, None, None, port.stream_ptr)
    port.stream.synchronize()
    t3 = time.perf_counter()
    for _ in range(N):
        port.lib.oc_step(port.bref, p[0], p[1], p[2], p[3], p[4], None, p[5], 65535, 0, None, None, port.stream_ptr)
        port.stream.synchronize()
    t4 = time.perf_counter()
    for _ in range(N):
        port.transition(state, 4, 4)
    t6 = time.perf_counter()
    print("pack %.1f us, unpack %.1f us, launch only (async, back to back) %.1f us, launch + synchronize %.1f us, port.transition %.1f us"
          % tuple(x / N * 1e6 for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t6 - t4)))
# the Python layers above the port
import cProfile
import pstats

pr = cProfile.Profile()
pr.enable()
episode()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
