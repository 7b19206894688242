"""Where the host time of one VecOvercookedEnv.step call goes (oc_step, one launch per batched step):
    python tools/time_step_host.py [layout] [n_envs]
Times the pieces of the Python wrapper separately (wall clock over many repetitions, the GPU idle or trailing)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=0)
acts = torch.randint(0, 6, (16, n, 2), dtype=torch.uint8, device=dev)
rows = [acts[i] for i in range(16)]
N = 20000


def wall(f, reps=N):
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return dt / reps * 1e6


a0 = rows[0]
print("acts[i]                      %.2f us" % wall(lambda: acts[3]))
print("dtype/shape/contig checks    %.2f us" % wall(lambda: (a0.dtype != torch.uint8 or a0.shape != (n, 2) or not a0.is_contiguous() or a0.device != env.state.device)))
print("data_ptr()                   %.2f us" % wall(lambda: a0.data_ptr()))
print("torch.cuda.current_device()  %.2f us" % wall(torch.cuda.current_device))
print("current_stream().cuda_stream %.2f us" % wall(lambda: torch.cuda.current_stream().cuda_stream))
raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if raw is not None:
    print("_cuda_getCurrentRawStream    %.2f us" % wall(lambda: raw(0)))
print("env.options                  %.2f us" % wall(lambda: env.options))
print("env._start_spec()            %.2f us" % wall(lambda: env._start_spec()))
args = (env._bref, env._state_ptr, env._state_ptr, a0.data_ptr(), env._rewards_ptr, env._flags_ptr, env._ep_ptr, None,
        env.horizon, env.options, None, None, 0)
fn = env.lib.oc_step
print("raw ctypes oc_step call      %.2f us  (includes the launch; GPU-bound if above the kernel time)" % wall(lambda: fn(*args)))
print("env.step(row)                %.2f us" % wall(lambda: env.step(a0)))
i = [0]


def loop():
    i[0] += 1
    env.step(acts[i[0] % 16])


print("env.step(acts[i %% 16])       %.2f us" % wall(loop))
