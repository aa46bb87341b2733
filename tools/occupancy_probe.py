import sys, torch
sys.path.insert(0, '.')
from sparc_amd import WireEDMEnv
for n, k, l in ((16384, 9, 8), (16384, 4, 8), (16384, 9, 4)):
    env = WireEDMEnv(num_envs=n, device="cuda:0"); env.set_kernel(k, l); env.reset(seed=1)
    env.step_many(env.make_action(), 10); torch.cuda.synchronize()
    print(env._backend.last_kernel(), "occupancy API blocks/CU:", env._backend.last_occupancy())
