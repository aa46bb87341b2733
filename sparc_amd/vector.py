"""Gymnasium-VectorEnv-style adapter with automatic reset.

The reference leaves observation and reward as TODOs (`envs/wire_edm.py:100-101,181-187`) and
has no vector API; this adapter gives RL code the usual contract on top of the batched
environment: `step()` advances one control interval (one fused launch), returns
`(obs, reward, terminated, truncated, info)` with a leading batch dimension, and environments
that terminated are reset at the start of the next `step()` ("next-step" autoreset), each with a
fresh Philox episode stream.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from . import _abi
from .envs.wire_edm import WireEDMEnv


def progress_reward(env: WireEDMEnv, prev: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Default reward of the adapter (the reference's `_calculate_reward` is a TODO returning 0.0,
    envs/wire_edm.py:185-187): micrometres cut during the control interval, minus 10 when the wire
    broke or collided in it, float32 per environment."""
    st = env.state
    cut = (st.workpiece_position - prev["workpiece_position"]).to(torch.float32)
    return cut - 10.0 * st.is_wire_broken.to(torch.float32)


class WireEDMVectorEnv:
    """Next-step autoreset (Gymnasium's ``AutoresetMode.NEXT_STEP``): the `step()` after the one that
    reported ``terminated`` / ``truncated`` for an environment starts a new episode for it.

    When the wrapped environment was built with ``autoreset=True`` the reset happens INSIDE the step
    kernel (`wedm_params.autoreset`, include/wedm_hip.h): an environment found terminated when the
    launch begins is re-initialised by that launch and stepped on, and with ``reward="progress"`` the
    kernel also writes the reward — `step()` is then exactly one kernel launch and no device-to-host
    read.  Otherwise (or with a callable reward) the adapter falls back to a masked `reset` launch and
    a torch expression."""

    def __init__(self, env: WireEDMEnv, *, max_episode_steps: Optional[int] = None, autoreset: bool = True,
                 reward=None):
        """``reward``: None keeps the environment's own reward (the reference's constant 0.0, or the
        in-kernel progress reward if the environment was built with ``reward="progress"``);
        ``"progress"`` selects `progress_reward` (computed in the kernel when the environment supports
        it); a callable ``f(env, prev) -> float32[N]`` receives the environment after the control
        interval and ``prev = {"workpiece_position": ...}`` snapshotted before it (all on the device)."""
        self.env = env
        self.num_envs = env.num_envs
        self.single_action_space = env.single_action_space
        self.single_observation_space = env.single_observation_space
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.autoreset = bool(autoreset)
        self.max_episode_steps = max_episode_steps
        self._in_kernel_reset = self.autoreset and bool(getattr(env, "autoreset", False))
        if getattr(env, "autoreset", False) and not self.autoreset:
            raise ValueError("the environment resets terminated environments in the kernel (autoreset=True): "
                             "the adapter cannot switch that off")
        if reward == "progress" and getattr(env, "reward_kind", None) == "progress":
            reward = None  # the kernel writes it
        self._reward_fn = progress_reward if reward == "progress" else reward
        if self._reward_fn is not None and not callable(self._reward_fn):
            raise ValueError("reward must be None, 'progress' or a callable")
        self._need_reset = torch.zeros(self.num_envs, dtype=torch.bool, device=env.device)
        self.episode_count = torch.zeros(self.num_envs, dtype=torch.int64, device=env.device)

    def reset(self, *, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        obs, info = self.env.reset(seed=seed, options=options)
        self._need_reset.zero_()
        return obs.clone(), info

    def step(self, action):
        """One control interval for every environment (1000 us by default)."""
        if self.autoreset:
            if self._in_kernel_reset:
                # terminated environments are reset by the launch itself; a truncated one is handed to it
                # through its DONE flag (device-side, no synchronisation)
                if self.max_episode_steps is not None:
                    self.env.state.done.logical_or_(self._need_reset)
            elif bool(self._need_reset.any().item()):
                self.env.reset(options={"mask": self._need_reset})  # same key, next episode stream
            self.episode_count += self._need_reset.to(torch.int64)
        prev = {"workpiece_position": self.env.state.workpiece_position.clone()} if self._reward_fn is not None else None
        if prev is not None and self._in_kernel_reset:  # what the kernel's reset will make of the marked environments
            prev["workpiece_position"] = torch.where(self._need_reset, torch.full_like(
                prev["workpiece_position"], float(self.env.config.initial_gap)), prev["workpiece_position"])
        obs, reward, terminated, truncated, info = self.env.step_control(action)
        if self._reward_fn is not None:
            reward = self._reward_fn(self.env, prev)
        else:
            reward = reward.clone()
        terminated = terminated.clone()
        if self.max_episode_steps is not None:
            truncated = (self.env.state.time >= self.max_episode_steps) & ~terminated
        else:
            truncated = truncated.clone()
        self._need_reset = terminated | truncated
        info = dict(info)
        info["episode"] = self.episode_count
        return obs.clone(), reward, terminated, truncated, info

    def close(self) -> None:
        self.env.close()

    @property
    def obs_names(self):
        return _abi.OBS_NAMES
