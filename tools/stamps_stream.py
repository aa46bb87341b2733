#!/usr/bin/env python
"""Phase stamps of wedm_step_stream from a -DWEDM_STAMPS build (diagnostic, never the shipped library).
usage: WEDM_HIP_LIB=build/ablate/libwedm_STAMPS.so python tools/stamps_stream.py [num_envs] [config3|config4|config2] [lanes]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import torch
from sparc_amd import WireEDMEnv, WireModuleParameters
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = sys.argv[2] if len(sys.argv) > 2 else "config3"
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
wire = WireModuleParameters(segment_len=0.625) if wl == "config3" else WireModuleParameters()
env = WireEDMEnv(num_envs=n, device="cuda:0", wire_params=wire)
env.set_kernel(6, lanes)
env.reset(seed=1234)
act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
nblk = (n * lanes + 255) // 256
buf = torch.zeros(nblk * 4 * 12, dtype=torch.int64, device="cuda")
L = env._backend._L
L.wedm_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.wedm_debug_set_stamp_buffer(env._backend._ctx, C.c_void_p(buf.data_ptr()))
for _ in range(20):
    env.step(act)
buf.zero_()
env.step(act)
torch.cuda.synchronize()
print(env._backend.last_kernel())
raw = buf.cpu().numpy().reshape(nblk * 4, 12).astype(np.float64)
names = ["kernel arguments here", "peak-current table requested, geometry constants here", "state rows requested", "state rows landed", "wire words requested", "first prelude done", "wire in LDS", "walk done",
         "patches + reduce + epilogue done", "stores issued", "stores landed"]
idx = [10, 11, 8, 9, 0, 2, 1, 3, 4, 5, 6]
prev = raw[:, 7]
for i, nm in zip(idx, names):
    d = raw[:, i] - prev
    print(f"  +{np.median(d):8.0f} (p10 {np.percentile(d, 10):8.0f} p90 {np.percentile(d, 90):8.0f} max {d.max():8.0f})  {nm}")
    prev = raw[:, i]
life = raw[:, 6] - raw[:, 7]
print(f"  wave lifetime: median {np.median(life):.0f} p90 {np.percentile(life, 90):.0f} max {life.max():.0f} cycles")
# chip-wide skew: meaningful with a -DWEDM_STAMPS -DWEDM_STAMPS_REAL build only (one 100 MHz clock, 10 ns per tick)
start, end = raw[:, 7], raw[:, 6]
t0 = start.min()
print("  wave starts after the first (p10/p50/p90/max):", np.percentile(start - t0, [10, 50, 90, 100]).round())
print("  wave ends after the first start (p1/p10/p50/p90/p99/max):", np.percentile(end - t0, [1, 10, 50, 90, 99, 100]).round())
