"""A backend that drives ``sparc_amd.WireEDMEnv`` with the CPU oracle on CPU tensors.

TEST SEAM ONLY: it lives under tests/, is injected explicitly
(``WireEDMEnv(..., device="cpu", backend=OracleBackend)``) and is never selected by the
product, whose only backend is the HIP library.  It lets the CPU suite exercise the
host logic (action plumbing, state views, sharding) and gives the GPU tests the
reference result for the same SoA blocks.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from oracle import oracle as orc
from sparc_amd import _abi


class OracleBackend:
    name = "oracle"
    math_mode = orc.MATH_PORTABLE
    stencil_mode = orc.STENCIL_F32
    n_threads = 0

    def __init__(self, params, num_envs, n_seg_max, device):
        assert device.type == "cpu", "the oracle backend works on host memory"
        self.params = params
        self.num_envs = num_envs
        self.n_seg_max = n_seg_max
        self.state = None
        self.geom = _abi.GeomPtrs(None, None)
        self._L = orc.lib()
        self._trace, self._trace_us, self._trace_count = None, 0, 0

    def bind_state(self, ptrs):
        self.state = ptrs

    def bind_geometry(self, ptrs):
        self.geom = ptrs

    def reset(self, mask_ptr, seed, reseed, fresh=False):
        rc = self._L.wedm_oracle_reset_batch(C.byref(self.params), C.byref(self.state), self.num_envs,
                                             mask_ptr, seed & (2**64 - 1), (1 if reseed else 0) | (2 if fresh else 0))
        assert rc == 0, rc
        # like wedm_reset: all n_seg_max temperature rows at the spool temperature, obs zeroed
        stride = self.state.stride
        T = np.ctypeslib.as_array(C.cast(self.state.T, C.POINTER(C.c_float)),
                                  shape=(_abi.t_quads(self.n_seg_max), stride, 4))  # quad-interleaved (include/wedm_hip.h)
        obs = np.ctypeslib.as_array(C.cast(self.state.obs, C.POINTER(C.c_float)),
                                    shape=(self.params.obs_dim, stride))
        if mask_ptr is None:
            T[:, : self.num_envs] = np.float32(self.params.spool_T)
            obs[:, : self.num_envs] = 0
        else:
            m = np.ctypeslib.as_array(C.cast(mask_ptr, C.POINTER(C.c_uint8)), shape=(self.num_envs,)).astype(bool)
            T[:, : self.num_envs][:, m] = np.float32(self.params.spool_T)
            obs[:, : self.num_envs][:, m] = 0

    def _run(self, n_substeps, action):
        stencil = max(int(self.stencil_mode), int(self.params.stencil_mode))
        rc = self._L.wedm_oracle_step_batch(C.byref(self.params), C.byref(self.state), C.byref(self.geom),
                                            C.byref(action), self.num_envs, self.n_seg_max, n_substeps,
                                            self.math_mode, stencil, self.n_threads)
        assert rc == 0, rc

    def step(self, n_substeps, action):
        """wedm_step, including the sampling schedule of wedm_bind_trace (include/wedm_hip.h):
        sample m (microseconds since the bind) is taken when m % every == 0, into ring slot
        (m / every - 1) % capacity; rows packed in ascending row order."""
        tr = self._trace
        if tr is None:
            self._run(n_substeps, action)
            return
        left = n_substeps
        while left > 0:
            chunk = min(left, tr.every - self._trace_us % tr.every)
            self._run(chunk, action)
            left -= chunk
            self._trace_us += chunk
            if self._trace_us % tr.every == 0:
                self._sample(self._trace_count % tr.capacity)
                self._trace_count += 1

    def _block(self, ptr, ctype, rows):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(rows, self.state.stride))

    def _sample(self, slot):
        tr = self._trace
        lo, cnt = tr.env_lo, tr.env_count
        for ptr, mask, src, ctype, nrows in (
                (tr.f64, tr.f64_mask, self.state.f64, C.c_double, _abi.F64_COUNT),
                (tr.i32, tr.i32_mask, self.state.i32, C.c_int32, _abi.I32_COUNT),
                (tr.i8, tr.i8_mask, self.state.i8, C.c_int8, _abi.I8_COUNT)):
            rows = [r for r in range(nrows) if (mask >> r) & 1]
            if not rows:
                continue
            dst = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(tr.capacity, len(rows), cnt))
            dst[slot] = self._block(src, ctype, nrows)[rows, lo:lo + cnt]
        if tr.T:
            dst = np.ctypeslib.as_array(C.cast(tr.T, C.POINTER(C.c_float)), shape=(tr.capacity, self.n_seg_max, cnt))
            nq = _abi.t_quads(self.n_seg_max)
            T = np.ctypeslib.as_array(C.cast(self.state.T, C.POINTER(C.c_float)), shape=(nq, self.state.stride, 4))
            # trace rings stay [capacity][n_seg_max][env_count]
            dst[slot] = T[:, lo:lo + cnt].transpose(0, 2, 1).reshape(4 * nq, cnt)[: self.n_seg_max]

    def bind_trace(self, desc):
        self._trace, self._trace_us, self._trace_count = desc, 0, 0

    def bind_rng_replay(self, table_ptr, n_steps):
        raise NotImplementedError("the CPU seam replays recorded draws per environment through oracle.Env.rng (RNG_REPLAY)")

    def trace_samples(self):
        return self._trace_count

    def set_kernel(self, variant):
        pass

    def last_kernel(self):
        return "oracle"

    def close(self):
        pass
