#!/usr/bin/env python
"""Phase stamps of wedm_step_split2 from a -DWEDM_STAMPS build (diagnostic, never the shipped library).
usage: WEDM_HIP_LIB=build/ablate/libwedm_STAMPS.so python tools/stamps_split2.py [num_envs] [config3|config4] [variant]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import torch
from sparc_amd import WireEDMEnv, WireModuleParameters
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = sys.argv[2] if len(sys.argv) > 2 else "config3"
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 6
wire = WireModuleParameters(segment_len=0.625) if wl == "config3" else WireModuleParameters()
env = WireEDMEnv(num_envs=n, device="cuda:0", wire_params=wire)
env.set_kernel(variant, 0)
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
print(env._backend.last_kernel())
raw = buf.cpu().numpy().reshape(nblk, 4, 8).astype(np.float64)
t0 = raw[:, :, 7].min()
entry = raw[:, 0, 7] - t0
print(f"{wl} N={n}: {nblk} blocks; block entry after kernel start: min {entry.min():.0f} median {np.median(entry):.0f} "
      f"p90 {np.percentile(entry, 90):.0f} max {entry.max():.0f} cycles")
def chain(w, names, idx):
    prev = raw[:, w, 7]
    for nm, i in zip(names, idx):
        d = raw[:, w, i] - prev
        print(f"  wave {w}  +{np.median(d):8.0f} (p90 {np.percentile(d, 90):8.0f} max {d.max():8.0f})  {nm}")
        prev = raw[:, w, i]
chain(0, ["state loaded", "prelude done", "barrier 1 passed", "barrier 2 passed", "epilogue done", "stored"], [0, 1, 2, 3, 4, 5])
chain(1, ["rows requested", "at barrier 1", "barrier 1 passed", "rows landed", "cells done, stores issued", "barrier 2 passed", "stores landed"], [0, 1, 2, 3, 4, 5, 6])
last = max(raw[:, 0, 5].max(), raw[:, 1:, 6].max())
print(f"  whole kernel (first entry -> last store): {last - t0:.0f} cycles")
# early blocks (first round) vs late blocks
order = np.argsort(raw[:, 0, 7])
half = len(order) // 2
for nm, sel in (("first half of the blocks to start", order[:half]), ("second half", order[half:])):
    print(f"  {nm}: entry median {np.median(raw[sel, 0, 7] - t0):.0f}, block lifetime median {np.median(raw[sel, 0, 5] - raw[sel, 0, 7]):.0f}")
