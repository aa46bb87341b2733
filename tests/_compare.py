"""Bit-exact comparison of two sets of SoA state blocks with readable diagnostics."""
from __future__ import annotations

import torch

from sparc_amd import _abi

_ROW_NAMES = {
    "f64": [f.name for f in _abi.F64], "i32": [f.name for f in _abi.I32], "i8": [f.name for f in _abi.I8],
    "obs": list(_abi.OBS_NAMES), "stats": [f.name for f in _abi.STAT], "reward": ["reward"],
}


def block_diffs(got, want, n, *, skip_rows=(), T_rows=None):
    """Return a list of human-readable mismatches between two ``clone_blocks()`` dicts
    over the first ``n`` environments.  NaN == NaN.  Empty list = bit-identical."""
    out = []
    for k in ("i32", "i8", "f64", "T", "obs", "stats", "reward"):
        if k not in got or k not in want:
            continue
        a, b = got[k][:, :n], want[k][:, :n]
        if k == "T":  # quad-interleaved block [seg / 4][env][4] -> one row per segment (padding cells included)
            a, b = (x.permute(0, 2, 1).reshape(-1, x.shape[1]) for x in (a, b))
            if T_rows is not None:
                a, b = a[:T_rows], b[:T_rows]
        if a.is_floating_point():
            neq = ~((a == b) | (a.isnan() & b.isnan()))
        else:
            neq = a != b
        if not bool(neq.any()):
            continue
        rows = torch.nonzero(neq.any(dim=1)).flatten().tolist()
        for r in rows:
            name = _ROW_NAMES[k][r] if k in _ROW_NAMES and r < len(_ROW_NAMES[k]) else f"row{r}"
            if (k, name) in skip_rows or name in skip_rows:
                continue
            envs = torch.nonzero(neq[r]).flatten()
            e = int(envs[0])
            out.append(f"{k}.{name}: {envs.numel()} envs differ; env {e}: got {a[r, e].item()!r} want {b[r, e].item()!r}")
    return out


def assert_blocks_equal(got, want, n, **kw):
    diffs = block_diffs(got, want, n, **kw)
    assert not diffs, "\n".join(diffs[:25])
