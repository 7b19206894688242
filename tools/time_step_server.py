"""Round trip of the resident batched step (oc_step_server_*): python tools/time_step_server.py [layout] [n_envs] [K]
K dependent steps through k_step_server + k_step_client (one client launch) against K oc_step launches and one oc_step_many."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
dev = torch.device("cuda:0")
env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=0)
acts = torch.randint(0, 6, (K, n, 2), dtype=torch.uint8, device=dev)
rew = torch.zeros((K, n, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((K, n), dtype=torch.uint8, device=dev)
with env.step_server(idle_ms=10.0, life_s=30.0) as sv:
    sv.play(acts[:100], rew[:100], fl[:100])
    best = 1e9
    for rep in range(5):
        sv.play(acts, rew, fl)
        best = min(best, sv.last_play_ms / K * 1e3)
    print("%s n=%d: resident step %.2f us per batched step (best of 5 plays of %d dependent steps; last %.2f)" % (layout, n, best, K, sv.last_play_ms / K * 1e3))
s = torch.cuda.current_stream()
for i in range(50):
    env.step(acts[i])
s.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(500):
    env.step(acts[i])
e1.record()
s.synchronize()
print("%s n=%d: oc_step %.2f us per call (back to back)" % (layout, n, e0.elapsed_time(e1) / 500 * 1e3))
e0.record()
env.step_many(acts, rew, fl)
e1.record()
s.synchronize()
print("%s n=%d: oc_step_many %.2f us per batched step (actions known %d steps ahead)" % (layout, n, e0.elapsed_time(e1) / K * 1e3, K))
