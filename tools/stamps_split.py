#!/usr/bin/env python
"""Latency chain of the single-microsecond split kernel from a -DWEDM_STAMPS build (diagnostic).
usage: WEDM_HIP_LIB=build/ablate/libwedm_STAMPS.so python tools/stamps_split.py [num_envs] [config3|config4]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import torch
from sparc_amd import WireEDMEnv, WireModuleParameters
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = sys.argv[2] if len(sys.argv) > 2 else "config3"
wire = WireModuleParameters(segment_len=0.625) if wl == "config3" else WireModuleParameters()
env = WireEDMEnv(num_envs=n, device="cuda:0", wire_params=wire)
env.set_kernel(5, 0)
env.reset(seed=1234)
act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
nblk = (n + 63) // 64
buf = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device="cuda")
L = env._backend._L
L.wedm_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.wedm_debug_set_stamp_buffer(env._backend._ctx, C.c_void_p(buf.data_ptr()))
for _ in range(20):
    env.step(act)
buf.zero_()
env.step(act)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(nblk, 4, 8).astype(np.float64)
t0 = raw[:, :, 7].min()            # first wave to enter the kernel
names = ["loop top (env loaded)", "prelude done", "barrier 1", "walk done", "barrier 2", "loop exit", "stored"]
w0 = raw[:, 0, :]                  # wave 0 of every block
entry = w0[:, 7] - t0
print(f"{wl} N={n}: {nblk} blocks; block entry time after kernel start: min {entry.min():.0f} median {np.median(entry):.0f} max {entry.max():.0f} (100 MHz ticks? see below)")
prev = w0[:, 7]
for i, nm in enumerate(names):
    d = w0[:, i] - prev
    print(f"  wave 0  +{np.median(d):8.0f} (max {d.max():8.0f})  {nm}")
    prev = w0[:, i]
w1 = raw[:, 1, :]
print(f"  wave 1 walk (barrier 1 -> walk done): median {np.median(w1[:, 3] - w1[:, 2]):.0f}")
print(f"  whole kernel (first entry -> last store): {raw[:, :, 6].max() - t0:.0f} ticks")
