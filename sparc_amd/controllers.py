"""On-device controllers and the control-loop driver.

The reference's smoke-run driver (`experiments/run_simulation.py:24-56,255-297`) evaluates a
proportional gap controller on the host after every control step.  Here the same law is
evaluated on the GPU from the state tensors — no host round-trip per control interval — and
returns a `DeviceAction` that is handed to the next fused launch.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .envs.wire_edm import DeviceAction, WireEDMEnv


class GapController:
    """P controller on the gap, both control modes (`create_gap_controller`,
    experiments/run_simulation.py:24-56).  Bit-compatible with the reference: the servo command
    is rounded to float32 exactly where the reference builds `np.array([delta], dtype=np.float32)`."""

    def __init__(self, desired_gap: float = 5.0, target_voltage: float = 80.0, current_mode: int = 7,
                 ON_time: float = 2.0, OFF_time: float = 33.0):
        self.desired_gap = float(desired_gap)
        self.target_voltage, self.current_mode = float(target_voltage), int(current_mode)
        self.ON_time, self.OFF_time = float(ON_time), float(OFF_time)
        self._cache = None

    def __call__(self, env: WireEDMEnv) -> DeviceAction:
        st = env.state
        error = (st.workpiece_position - st.wire_position) - self.desired_gap
        if env.mechanics_control_mode == "position":
            delta = error * 0.1
        else:
            delta = torch.clamp(error * 50.0, -1000.0, 1000.0)
        servo = delta.to(torch.float32).to(torch.float64).contiguous()
        if self._cache is None or self._cache[0] is not env:
            base = env.make_action(0.0, self.target_voltage, self.current_mode, self.ON_time, self.OFF_time)
            self._cache = (env, base)
        base = self._cache[1]
        return DeviceAction(servo, base.target_voltage, base.on_time, base.off_time, base.current_mode)


def run_controlled(env: WireEDMEnv, controller: Callable[[WireEDMEnv], DeviceAction], n_steps: int,
                   on_control_step: Optional[Callable[[WireEDMEnv, int], None]] = None) -> int:
    """The driver loop of experiments/run_simulation.py:241-297 with fused launches.

    The reference computes the first action before any step, latches it on the first control
    step (call number ``servo_interval + 1``), recomputes the action right after every control
    step and latches THAT one a whole interval later.  Launch lengths reproduce exactly this:
    ``servo_interval - time_since_servo + 1`` microseconds up to and including the next latch,
    then one ``servo_interval`` per launch.  Returns the number of microseconds run.
    """
    interval = env.servo_interval // env.dt
    tss = int(env.state.time_since_servo.max().item())  # one host read, before the loop
    action = controller(env)
    done = 0
    next_k = max(1, interval - tss + 1)  # microseconds up to and including the next latch
    while done < n_steps:
        k = min(next_k, n_steps - done)
        env.step_many(action, k)
        done += k
        if k < next_k:
            break  # n_steps ran out between two control steps
        action = controller(env)  # the launch ended on a control step (run_simulation.py:272-281)
        if on_control_step is not None:
            on_control_step(env, done)
        next_k = interval
    return done
