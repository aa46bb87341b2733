#!/usr/bin/env python
"""Phase breakdown of the fused kernels from a -DWEDM_STAMPS build (diagnostic only).
usage: WEDM_HIP_LIB=build/ablate/libwedm_STAMPS.so python tools/stamps.py <kernel> <lanes> [config3|config4|config2] [num_envs] [gap_um]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from sparc_amd import WireEDMEnv, WireModuleParameters
kernel, lanes = int(sys.argv[1]), int(sys.argv[2])
wl = sys.argv[3] if len(sys.argv) > 3 else "config3"
wire = WireModuleParameters(segment_len=0.625) if wl == "config3" else WireModuleParameters()
n = int(sys.argv[4]) if len(sys.argv) > 4 else {"config3": 65536, "config4": 32768, "config2": 4096}.get(wl, 65536)
env = WireEDMEnv(num_envs=n, device="cuda:0", wire_params=wire)
env.set_kernel(kernel, lanes)
env.reset(seed=1234)
if len(sys.argv) > 5:
    env.state.wire_position = 10.0
    env.state.workpiece_position = 10.0 + float(sys.argv[5])
    env.state.target_position = 5000.0
act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
buf = torch.zeros(n * 64, dtype=torch.int64, device="cuda")
L = env._backend._L
L.wedm_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.wedm_debug_set_stamp_buffer(env._backend._ctx, C.c_void_p(buf.data_ptr()))
env.step_many(act, 1000)
buf.zero_()
env.step_many(act, 1000)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
nblk = int(env._backend.last_kernel().split("<<<")[1].split(",")[0])
b = raw[: nblk * 16].reshape(-1, 4)
t = raw[nblk * 16: nblk * 16 + nblk * 24].reshape(-1, 6)
t = t[t[:, 3:].sum(axis=1) > 0]
if len(t):
    tm = t.mean(axis=0)
    if kernel == 4:  # the packed kernel uses these slots for the prelude: quiet fast path vs general path
        print("   prelude: quiet path on %.0f %% of the steps at %.0f cycles, general path %.0f %% at %.0f cycles" % (
            tm[3] / 10, tm[0] / max(tm[3], 1), tm[4] / 10, tm[1] / max(tm[4], 1)))
    else:
        print("   tiles per step: N %.2f B %.2f S %.2f ; cycles per tile: N %.0f B %.0f S %.0f" % (
            tm[3] / 1000, tm[4] / 1000, tm[5] / 1000, tm[0] / max(tm[3], 1), tm[1] / max(tm[4], 1), tm[2] / max(tm[5], 1)))
b = b[b.sum(axis=1) > 0]
m = b.mean(axis=0) / 1000.0
print(f"kernel {kernel} lanes {lanes} {wl}: waves {len(b)}  cycles/step: prelude {m[0]:.0f}  walk {m[1]:.0f}  patches+reduce {m[2]:.0f}  epilogue {m[3]:.0f}  total {m.sum():.0f}  ({env._backend.last_kernel()})")
