"""Loader of ``libwedm_hip.so`` — the only compute path of this package.

There is deliberately no fallback: if the HIP library is missing or no gfx950
device is visible, construction fails with an explicit error instead of silently
running something else."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import _abi

# WEDM_HIP_LIB lets the profiling scripts load an instrumented build of the SAME library;
# it never selects a different implementation.
import os

LIB_PATH = Path(os.environ.get("WEDM_HIP_LIB") or (Path(__file__).resolve().parent / "libwedm_hip.so"))


class WedmError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"{_abi.STATUS_NAMES.get(code, code)}: {text}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """dlopen the library and declare every entry point of include/wedm_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  sparc_amd has no CPU or PyTorch fallback."
        )
    L = C.CDLL(str(LIB_PATH))
    L.wedm_abi_version.restype = C.c_int32
    L.wedm_sizeof_params.restype = C.c_int64
    if L.wedm_abi_version() != _abi.ABI_VERSION:
        raise ImportError(f"libwedm_hip.so has ABI {L.wedm_abi_version()}, python expects {_abi.ABI_VERSION}")
    if L.wedm_sizeof_params() != C.sizeof(_abi.Params):
        raise ImportError("struct wedm_params layout mismatch between libwedm_hip.so and sparc_amd._abi")
    ctx = C.c_void_p
    L.wedm_create.argtypes = [C.POINTER(_abi.Params), C.c_int32, C.c_int32, C.POINTER(ctx)]
    L.wedm_create.restype = C.c_int32
    L.wedm_destroy.argtypes = [ctx]
    L.wedm_destroy.restype = C.c_int32
    L.wedm_bind_state.argtypes = [ctx, C.POINTER(_abi.StatePtrs)]
    L.wedm_bind_state.restype = C.c_int32
    L.wedm_bind_geometry.argtypes = [ctx, C.POINTER(_abi.GeomPtrs)]
    L.wedm_bind_geometry.restype = C.c_int32
    L.wedm_reset.argtypes = [ctx, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p]
    L.wedm_reset.restype = C.c_int32
    L.wedm_step.argtypes = [ctx, C.c_int32, C.POINTER(_abi.ActionPtrs), C.c_void_p]
    L.wedm_step.restype = C.c_int32
    L.wedm_bind_trace.argtypes = [ctx, C.POINTER(_abi.TraceDesc)]
    L.wedm_bind_trace.restype = C.c_int32
    L.wedm_bind_rng_replay.argtypes = [ctx, C.c_void_p, C.c_int64]
    L.wedm_bind_rng_replay.restype = C.c_int32
    L.wedm_trace_samples.argtypes = [ctx]
    L.wedm_trace_samples.restype = C.c_int64
    L.wedm_set_kernel.argtypes = [ctx, C.c_int32]
    L.wedm_set_kernel.restype = C.c_int32
    L.wedm_set_lanes.argtypes = [ctx, C.c_int32]
    L.wedm_set_lanes.restype = C.c_int32
    L.wedm_last_error.argtypes = [ctx]
    L.wedm_last_error.restype = C.c_char_p
    L.wedm_last_kernel.argtypes = [ctx]
    L.wedm_last_kernel.restype = C.c_char_p
    L.wedm_last_occupancy.argtypes = [ctx]
    L.wedm_last_occupancy.restype = C.c_int32
    L.wedm_debug_math.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.wedm_debug_math.restype = C.c_int32
    L.wedm_debug_poison_lds.argtypes = [C.c_float, C.c_void_p]
    L.wedm_debug_poison_lds.restype = C.c_int32
    L.wedm_build_id.argtypes = []
    L.wedm_build_id.restype = C.c_char_p
    _lib = L
    return L


EXPORTS = (
    "wedm_abi_version", "wedm_create", "wedm_destroy", "wedm_bind_state", "wedm_bind_geometry",
    "wedm_reset", "wedm_step", "wedm_bind_trace", "wedm_bind_rng_replay", "wedm_trace_samples", "wedm_set_kernel", "wedm_set_lanes", "wedm_last_kernel", "wedm_last_error", "wedm_last_occupancy",
    "wedm_sizeof_params", "wedm_debug_math", "wedm_debug_poison_lds", "wedm_build_id",
)


def build_id() -> str:
    """`wedm_build_id()` of the loaded library: the fingerprint measurement records are bound to (include/wedm_hip.h)."""
    return load().wedm_build_id().decode()


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


class HipBackend:
    """Thin owner of one ``wedm_ctx``.  All memory stays with the caller (torch)."""

    name = "hip"

    def __init__(self, params: _abi.Params, num_envs: int, n_seg_max: int, device):
        import torch

        if device.type != "cuda":
            raise RuntimeError(
                f"sparc_amd runs on an AMD GPU only (device={device}); there is no CPU path. "
                "Use device='cuda' on an MI355X."
            )
        self._L = load()
        self._torch = torch
        self.device = device
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._ctx = C.c_void_p()
        with torch.cuda.device(device):
            rc = self._L.wedm_create(C.byref(params), num_envs, n_seg_max, C.byref(self._ctx))
        if rc != _abi.OK:
            raise WedmError(rc, (self._L.wedm_last_error(None) or b"").decode())

    def _check(self, rc: int) -> None:
        if rc != _abi.OK:
            raise WedmError(rc, (self._L.wedm_last_error(self._ctx) or b"").decode())

    def _stream(self):
        # raw handle of torch's current stream on this device (the public accessor builds a
        # Stream object per call: several microseconds on the per-microsecond stepping path)
        raw = self._raw_stream
        if raw is not None:
            return C.c_void_p(raw(self.device.index))
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def bind_state(self, ptrs: _abi.StatePtrs) -> None:
        self._check(self._L.wedm_bind_state(self._ctx, C.byref(ptrs)))

    def bind_geometry(self, ptrs: _abi.GeomPtrs) -> None:
        self._check(self._L.wedm_bind_geometry(self._ctx, C.byref(ptrs)))

    def _on_device(self):
        """The library launches on whatever device is current and refuses a handle of another one
        (wedm_step / wedm_reset return WEDM_ERR_BAD_ARG): make this environment's device current
        for the call when it is not (an environment on cuda:1 in a process whose current device is
        cuda:0).  The common case — already current — costs one integer compare."""
        torch = self._torch
        if torch.cuda.current_device() == self.device.index:
            return _NO_GUARD
        return torch.cuda.device(self.device)

    def reset(self, mask_ptr, seed: int, reseed: bool, fresh: bool = False) -> None:
        """`fresh`: WEDM_RESET_FRESH -- module-private state too, whatever `reset_semantics` (a handle's first reset)."""
        with self._on_device():
            flags = (1 if reseed else 0) | (2 if fresh else 0)
            self._check(self._L.wedm_reset(self._ctx, mask_ptr, seed & (2**64 - 1), flags, self._stream()))

    def step(self, n_substeps: int, action: _abi.ActionPtrs) -> None:
        with self._on_device():
            self._check(self._L.wedm_step(self._ctx, n_substeps, C.byref(action), self._stream()))

    def bind_trace(self, desc) -> None:
        """`desc` is an `_abi.TraceDesc` or None (unbind)."""
        self._check(self._L.wedm_bind_trace(self._ctx, C.byref(desc) if desc is not None else None))

    def bind_rng_replay(self, table_ptr, n_steps: int) -> None:
        self._check(self._L.wedm_bind_rng_replay(self._ctx, table_ptr, int(n_steps)))

    def trace_samples(self) -> int:
        return int(self._L.wedm_trace_samples(self._ctx))

    def set_kernel(self, variant: int) -> None:
        self._check(self._L.wedm_set_kernel(self._ctx, variant))

    def set_lanes(self, lanes: int) -> None:
        self._check(self._L.wedm_set_lanes(self._ctx, lanes))

    def last_kernel(self) -> str:
        return (self._L.wedm_last_kernel(self._ctx) or b"").decode()

    def last_occupancy(self) -> int:
        """Blocks per CU the occupancy API admits for the last launch's kernel, block size and LDS (diagnostic)."""
        return int(self._L.wedm_last_occupancy(self._ctx))

    def build_id(self) -> str:
        return build_id()

    def close(self) -> None:
        if self._ctx:
            self._L.wedm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
