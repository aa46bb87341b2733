#!/usr/bin/env python
"""Where the time of tests/test_gpu_parity.py::test_negative_plasma_heat_every_kernel_matches_oracle goes, per kernel variant
(GPU launch, single step, CPU oracle, comparison).  Diagnostic."""
import sys, time, torch
sys.path.insert(0, '.')
from sparc_amd import WireEDMEnv, WireModuleParameters, EnvironmentConfig
from tests._oracle_backend import OracleBackend
from tests._compare import block_diffs
kw = dict(wire_params=WireModuleParameters(segment_len=0.2, plasma_efficiency=-0.2), config=EnvironmentConfig(target_cutting_distance=5000.0))
gpu = WireEDMEnv(num_envs=200, device="cuda:0", **kw)
cpu = WireEDMEnv(num_envs=200, device="cpu", backend=OracleBackend, **kw)
T = time.perf_counter
for variant, lanes in [(1, 0), (5, 0), (6, 0), (6, 4), (6, 16), (2, 0), (3, 1), (3, 2), (3, 4), (3, 8), (3, 16), (4, 1), (4, 2), (4, 4), (4, 8), (10, 0), (2, 4), (2, 16), (0, 0), (7, 0), (8, 0), (9, 4), (9, 8), (11, 4), (11, 8), (11, 16)]:
    gpu.set_kernel(variant, lanes)
    for env in (gpu, cpu):
        env.reset(seed=515)
        env.state.workpiece_position = 18.0; env.state.wire_position = 10.0; env.state.target_position = 5000.0
    torch.cuda.synchronize(); t0 = T()
    try:
        gpu.step_many(gpu.make_action(0.05, 80.0, 17, 3.0, 15.0), 700)
        torch.cuda.synchronize(); t1 = T()
        gpu.step(gpu.make_action(0.05, 80.0, 17, 3.0, 15.0))
        torch.cuda.synchronize(); t2 = T()
    except Exception as exc:
        print(variant, lanes, "unsupported"); continue
    cpu.step_many(cpu.make_action(0.05, 80.0, 17, 3.0, 15.0), 701); t3 = T()
    d = block_diffs(gpu.state.clone_blocks(), cpu.state.clone_blocks(), 200); t4 = T()
    print(f"{variant} {lanes}: fused {t1 - t0:.3f} s, single {t2 - t1:.3f} s ({gpu._backend.last_kernel().split('<<<')[0]}), oracle {t3 - t2:.3f} s, compare {t4 - t3:.3f} s, diffs {len(d)}", flush=True)
