"""Sharding of the batched environment over the GPUs of one node.

Environments are independent, so the batch is cut into contiguous
ranges, one process per GPU, and NOTHING is exchanged in the physics.  The Philox
counter carries the GLOBAL environment id (``env_id_offset``), so a shard reproduces
exactly the trajectories the same environments have in a single-process run.  The one
collective is an all-gather of the control-step observations (and done flags) over
RCCL/xGMI — ``torch.distributed`` backend "nccl" on ROCm; "gloo" works for CPU tests.

The reference has no distributed code; this module has no counterpart there.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _abi
from .envs.wire_edm import DeviceAction, WireEDMEnv


def _slice_leaf(x, lo: int, hi: int, total: int):
    """Per-environment leaves given for the GLOBAL batch are cut to this rank's range."""
    if x is None:
        return None
    if torch.is_tensor(x):
        return x.reshape(-1)[lo:hi] if x.numel() == total else x
    arr = np.asarray(x)
    return arr.reshape(-1)[lo:hi] if arr.size == total and total > 1 else x


class ShardedWireEDMEnv:
    """``global_num_envs`` environments spread over the ranks of ``process_group``.

    Each rank owns environments ``[rank * n_local, (rank + 1) * n_local)``.  ``reset`` /
    ``step`` / ``step_many`` act on the local shard (same tuples as ``WireEDMEnv``);
    ``gather_obs`` / ``gather_done`` return the whole batch on every rank.
    """

    def __init__(self, global_num_envs: int, *, process_group: Optional[Any] = None, device: Any = None,
                 workpiece_height=None, wire_diameter=None, **env_kwargs):
        if not dist.is_initialized():
            raise RuntimeError("ShardedWireEDMEnv needs torch.distributed.init_process_group first")
        self.group = process_group
        self.rank = dist.get_rank(process_group)
        self.world_size = dist.get_world_size(process_group)
        if global_num_envs % self.world_size:
            raise ValueError("global_num_envs must be divisible by the number of ranks")
        self.global_num_envs = int(global_num_envs)
        self.num_envs = self.global_num_envs // self.world_size
        self.lo, self.hi = self.rank * self.num_envs, (self.rank + 1) * self.num_envs
        self.env = WireEDMEnv(
            num_envs=self.num_envs, device=device, env_id_offset=self.lo,
            workpiece_height=_slice_leaf(workpiece_height, self.lo, self.hi, self.global_num_envs),
            wire_diameter=_slice_leaf(wire_diameter, self.lo, self.hi, self.global_num_envs),
            **env_kwargs)
        self.device = self.env.device
        self.state = self.env.state
        # outputs are the rank-major concatenation of the inputs (the layout every backend accepts)
        self._obs_all = torch.empty((self.world_size * _abi.OBS_DIM, self.num_envs), dtype=torch.float32,
                                    device=self.device)
        self._done_all = torch.empty((self.world_size * self.num_envs,), dtype=torch.uint8, device=self.device)

    # ------------------------------------------------------------------ local physics
    def reset(self, *, seed: Optional[int] = None, options=None):
        if options and options.get("mask") is not None:
            options = dict(options, mask=_slice_leaf(options["mask"], self.lo, self.hi, self.global_num_envs))
        return self.env.reset(seed=seed, options=options)

    def make_action(self, servo=0.0, target_voltage=80.0, current_mode=5, ON_time=3.0, OFF_time=80.0) -> DeviceAction:
        cut = lambda x: _slice_leaf(x, self.lo, self.hi, self.global_num_envs)  # noqa: E731
        return self.env.make_action(cut(servo), cut(target_voltage), cut(current_mode), cut(ON_time), cut(OFF_time))

    def _local_action(self, action):
        if isinstance(action, DeviceAction):
            return action
        gc = action["generator_control"]
        return self.make_action(action["servo"], gc["target_voltage"], gc["current_mode"], gc["ON_time"], gc["OFF_time"])

    def step(self, action):
        return self.env.step(self._local_action(action))

    def step_many(self, action, n_substeps: int):
        return self.env.step_many(self._local_action(action), n_substeps)

    def step_control(self, action, gather: bool = True):
        """One control interval on every shard, then (optionally) the observation all-gather."""
        out = self.env.step_control(self._local_action(action))
        if gather:
            return (self.gather_obs(),) + tuple(out[1:])
        return out

    # ------------------------------------------------------------------ the one collective
    def gather_obs(self) -> torch.Tensor:
        """``float32[global_num_envs, 8]`` on every rank (all-gather over xGMI)."""
        local = self.env.state.obs[:, : self.num_envs].contiguous()
        dist.all_gather_into_tensor(self._obs_all, local, group=self.group)
        return self._obs_all.view(self.world_size, _abi.OBS_DIM, self.num_envs).permute(0, 2, 1).reshape(
            self.global_num_envs, _abi.OBS_DIM)

    def gather_done(self) -> torch.Tensor:
        local = self.env.state.done.to(torch.uint8).contiguous()
        dist.all_gather_into_tensor(self._done_all, local, group=self.group)
        return self._done_all.bool()

    def close(self) -> None:
        self.env.close()


class PipelinedObsGather:
    """The observation all-gather taken off the critical path.

    After control interval k the local observations are snapshotted into a staging buffer (a
    2-MB device copy on the compute stream) and all-gathered ASYNCHRONOUSLY (the process group
    runs the collective on its own stream), while the compute stream goes straight on to interval
    k+1 — whose control step overwrites the live observation block, hence the snapshot.
    `post()` after every `step_many`; `result()` waits for the newest gather.  Same bytes over
    xGMI per control step as the blocking form, overlapped with the next launch."""

    def __init__(self, obs_local: torch.Tensor, world_size: int, group: Optional[Any] = None):
        self.obs_local = obs_local                        # [obs_dim, n_local] view of the live block
        self.stage = torch.empty_like(obs_local, memory_format=torch.contiguous_format)
        self.out = torch.empty((world_size * obs_local.shape[0], obs_local.shape[1]), dtype=obs_local.dtype,
                               device=obs_local.device)
        self.group = group
        self._work = None

    def post(self) -> None:
        if self._work is not None:
            self._work.wait()          # the previous gather has consumed the staging buffer
        self.stage.copy_(self.obs_local)
        self._work = dist.all_gather_into_tensor(self.out, self.stage, group=self.group, async_op=True)

    def result(self) -> torch.Tensor:
        """Rank-major ``[world * obs_dim, n_local]`` block of the newest posted gather."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self.out
