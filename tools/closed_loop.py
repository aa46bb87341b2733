#!/usr/bin/env python
"""Closed-loop side measurement: the reference's driver scenario (experiments/run_simulation.py:
gap or voltage controller recomputed on the device at every control step) at the bench batch size.
usage: python tools/closed_loop.py [gap|voltage] [intervals] [config3|config4] [approach_ms] [kernel variant] [lanes]"""
import sys
import time

sys.path.insert(0, ".")
import torch

from sparc_amd import GapController, VoltageController, WireEDMEnv, WireModuleParameters, run_controlled

kind = sys.argv[1] if len(sys.argv) > 1 else "gap"
intervals = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wl = sys.argv[3] if len(sys.argv) > 3 else "config3"
n, wire = (65536, WireModuleParameters(segment_len=0.625)) if wl == "config3" else (32768, WireModuleParameters())
env = WireEDMEnv(num_envs=n, device="cuda:0", wire_params=wire)
if len(sys.argv) > 5:
    env.set_kernel(int(sys.argv[5]), int(sys.argv[6]) if len(sys.argv) > 6 else 0)
env.reset(seed=1)
env.state.workpiece_position = 70.0      # run_simulation.py:199-201
env.state.wire_position = 10.0
env.state.target_position = 5000.0
ctl = GapController() if kind == "gap" else VoltageController(30.0)
approach = int(sys.argv[4]) if len(sys.argv) > 4 else 20
run_controlled(env, ctl, approach * 1000 + 1)   # approach phase: the controller closes the 60 um gap
torch.cuda.synchronize()
s0 = int(env.state.spark_count.sum())
t0 = time.perf_counter()
done = run_controlled(env, ctl, intervals * 1000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gap = (env.state.workpiece_position - env.state.wire_position)
print(f"{kind} controller, {wl}: {n * done / dt:.3e} env-steps/s over {done} us ({dt / intervals * 1e3:.2f} ms per control interval), "
      f"{(int(env.state.spark_count.sum()) - s0) / n / (done / 1000):.1f} sparks per env per ms, mean gap {float(gap.mean()):.2f} um, "
      f"kernel {env._backend.last_kernel()}")

# where a control interval's time goes: the launch (HIP events on the launch stream) vs everything else
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(intervals)]
action = ctl(env)
torch.cuda.synchronize()
t0 = time.perf_counter()
for a, b in ev:
    a.record()
    env.step_many(action, 1000)
    b.record()
    action = ctl(env)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
kms = sum(a.elapsed_time(b) for a, b in ev) / intervals
print(f"  per control interval: launch {kms:.3f} ms (HIP events), wall {dt / intervals * 1e3:.3f} ms -> "
      f"controller + host {dt / intervals * 1e3 - kms:.3f} ms")
