#!/usr/bin/env python
"""Throughput of a batch whose environments terminate at different times (in-kernel autoreset on): the bench batch with
a cutting target a few sparks ahead of the initial gap, so that in steady state every launch sees a fraction of the
environments reach their target and stay frozen until the next launch re-initialises them.
usage: python tools/terminating_batch.py [launches] [target_ahead_um]"""
import sys
import time

sys.path.insert(0, ".")
import torch

from sparc_amd import EnvironmentConfig, WireEDMEnv, WireModuleParameters

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ahead = float(sys.argv[2]) if len(sys.argv) > 2 else 0.002
n = 65536
cfg = EnvironmentConfig(target_cutting_distance=50.0 + ahead)
env = WireEDMEnv(num_envs=n, device="cuda:0", config=cfg, wire_params=WireModuleParameters(segment_len=0.625),
                 autoreset=True, reward="progress")
env.reset(seed=7)
act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
for _ in range(20):
    env.step_many(act, 1000)
torch.cuda.synchronize()
e0 = int(env.state.episode.sum())
t0 = time.perf_counter()
for _ in range(launches):
    env.step_many(act, 1000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"terminating batch (target {ahead} um ahead): {n * 1000 * launches / dt:.3e} env-steps/s ({dt / launches * 1e3:.2f} ms per launch), "
      f"{(int(env.state.episode.sum()) - e0) / n / launches:.3f} resets per env per launch, kernel {env._backend.last_kernel()}")
