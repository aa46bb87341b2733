#!/usr/bin/env python
"""Time fused launches of an arbitrary (num_envs, segment_len, kernel, lanes) shape (diagnostic):
    python tools/probe_shape.py 8192 0.4 3 16"""
import sys
sys.path.insert(0, ".")
import torch
from sparc_amd import WireEDMEnv, WireModuleParameters
n, seg, variant, lanes = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
env = WireEDMEnv(num_envs=n, device="cuda:0", wire_params=WireModuleParameters(segment_len=seg))
env.set_kernel(variant, lanes)
env.reset(seed=1234)
act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
for _ in range(3):
    env.step_many(act, 1000)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(10):
    env.step_many(act, 1000)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(f"N={n} S={env.n_segments} {env._backend.last_kernel()}: {ms:.3f} ms per 1000 us, {n * 1000 / ms * 1e3:.4g} env-steps/s")
