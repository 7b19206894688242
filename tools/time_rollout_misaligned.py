import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from overcooked_ai_amd.vec_env import VecOvercookedEnv
dev = torch.device("cuda:0")
n, T = 65536, 4000
for lay in ("asymmetric_advantages", "cramped_room"):
    for pre in (0, 3):
        for ow in (False, True):
            env = VecOvercookedEnv(lay, n, horizon=400, device=dev, auto_reset=True, seed=0)
            env.one_wavefront = ow
            rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev); fl = torch.zeros((T, n), dtype=torch.uint8, device=dev)
            if pre: env.rollout_random(pre)
            env.rollout_random(T, rew, fl); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): env.rollout_random(T, rew, fl)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("%s first step %% 8 = %d, %s: %.3f ms per 4000-step call -> %.1f G env-steps/s" % (lay, pre % 8, "one-wavefront instances" if ow else "default dispatch", ms, n * T / ms / 1e6))
            del env, rew, fl
