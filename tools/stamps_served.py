#!/usr/bin/env python
"""Where the served kernel's waves spend their time, and which blocks share a CU (diagnostic build only):
    python tools/build_variant.py SVSTAMPS --only-part 3 -DWEDM_STAMPS   (part 0 must come from a -DWEDM_STAMPS build too: see below)
    WEDM_HIP_LIB=build/ablate/libwedm_SVSTAMPS.so python tools/stamps_served.py <lanes> <num_envs> [gap_um]"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from sparc_amd import WireEDMEnv

lanes, n = int(sys.argv[1]), int(sys.argv[2])
env = WireEDMEnv(num_envs=n, device="cuda:0")
env.set_kernel(9, lanes)
env.reset(seed=1234)
if len(sys.argv) > 3:
    env.state.wire_position = 10.0
    env.state.workpiece_position = 10.0 + float(sys.argv[3])
    env.state.target_position = 5000.0
act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
WW = 3  # walker waves per block (wedm_served.h)
nblk = (n + WW * 64 // lanes - 1) // (WW * 64 // lanes)
buf = torch.zeros(nblk * (WW + 1) * 12, dtype=torch.int64, device="cuda")
L = env._backend._L
L.wedm_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.wedm_debug_set_stamp_buffer(env._backend._ctx, C.c_void_p(buf.data_ptr()))
env.step_many(act, 1000)
buf.zero_()
env.step_many(act, 1000)
torch.cuda.synchronize()
r = buf.cpu().numpy().reshape(nblk, WW + 1, 12)
hw, xcc = r[:, :, 0], r[:, :, 1] & 0xF
cu = ((xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))[:, 0]
simd = (hw >> 4) & 3
t0, t1 = r[:, 0, 2].astype(np.float64) / 100.0, r[:, :, 3].max(axis=1).astype(np.float64) / 100.0  # us
print(f"{env._backend.last_kernel()}: {nblk} blocks on {len(set(cu.tolist()))} distinct CUs; occupancy API {env._backend.last_occupancy()} blocks/CU")
print(f"block lifetime us: min {np.min(t1 - t0):.0f} median {np.median(t1 - t0):.0f} max {np.max(t1 - t0):.0f}; launch span {t1.max() - t0.min():.0f} us")
# how many blocks of the same CU overlap in time
over = []
for c in set(cu.tolist()):
    idx = np.nonzero(cu == c)[0]
    ev = sorted([(t0[i], 1) for i in idx] + [(t1[i], -1) for i in idx])
    cur = best = 0
    for _, d in ev:
        cur += d
        best = max(best, cur)
    over.append(best)
print("blocks resident together per CU: " + ", ".join(f"{k}: {over.count(k)} CUs" for k in sorted(set(over))))
print("SIMD of the waves of block 0:", simd[0].tolist(), " of block 1:", simd[1].tolist() if nblk > 1 else "-")
for name, sel in (("walker", slice(0, WW)), ("scalar", slice(WW, WW + 1))):
    wait, tot = r[:, sel, 4].astype(np.float64), r[:, sel, 5].astype(np.float64)
    print(f"{name} waves: loop {tot.mean() / 1000:.0f} cycles per step, of which spinning {wait.mean() / 1000:.0f} ({100 * wait.sum() / tot.sum():.0f} %)")
w, sc = r[:, :WW, 7:10].astype(np.float64).mean(axis=(0, 1)) / 1000, r[:, WW, 7:10].astype(np.float64).mean(axis=0) / 1000
print(f"walker phases, cycles per step: mailbox + halos + patched cells from old values {w[0]:.0f}, tiles {w[1]:.0f}, patches + reduction + publication {w[2]:.0f}")
print(f"scalar phases, cycles per step: prelude + publication {sc[0]:.0f}, previous monitor (incl. its wait) {sc[1]:.0f}, proof + rest of the epilogue {sc[2]:.0f}")
print(f"steps the scalar wave ran ahead: {r[:, WW, 6].mean():.0f} of 1000")
