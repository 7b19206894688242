import sys, os, torch
sys.path.insert(0, '.')
from overcooked_ai_amd.vec_env import VecOvercookedEnv
dev = torch.device('cuda:0')
env = VecOvercookedEnv("cramped_room", 65536, horizon=400, device=dev, auto_reset=True, seed=0)
mode = os.environ.get("MODE", "")
if mode: setattr(env, mode, True)
rew = torch.zeros((100, 65536, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((100, 65536), dtype=torch.uint8, device=dev)
for _ in range(12):
    env.rollout_random(100, rew, fl)
torch.cuda.synchronize()
