import sys, torch
sys.path.insert(0, '.')
from overcooked_ai_amd.vec_env import VecOvercookedEnv
dev = torch.device('cuda:0')
env = VecOvercookedEnv("asymmetric_advantages", 65536, horizon=400, device=dev, auto_reset=True, seed=1)
env.rollout_random(150)
for dt in (torch.uint8, torch.float32):
    obs = torch.empty((65536, 2, 9, 5, 26), dtype=dt, device=dev)
    for _ in range(20):
        env.encode_lossless(dt, out=obs)
torch.cuda.synchronize()
