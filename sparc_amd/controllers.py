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


class VoltageController:
    """PI controller on the voltage averaged over the last millisecond
    (`create_voltage_controller`, experiments/run_simulation.py:58-109), one integrator per
    environment, evaluated on the GPU.

    The reference's driver appends `state.voltage` after every 1-us step and keeps the samples
    whose time is >= time - 1000, i.e. up to 1001 of them (run_simulation.py:262-270).  Here the
    step kernels record the voltage of every microsecond into a 1001-slot ring
    (`WireEDMEnv.bind_trace`) and the controller averages the ring at the control step.  Sum and
    division are float64; when the voltages are integers (the 0 / 24 / 80 V of the default
    generator settings) the sum is exact in any order and the command equals the reference's
    bit for bit, otherwise it may differ in the last bit before the float32 rounding of the
    servo leaf.  Before the first step the reference falls back to `state.voltage` (None -> 0)."""

    WINDOW = 1001

    def __init__(self, target_voltage: float = 30.0, *, Kp: float = 0.05, Ki: float = 0.1,
                 generator_voltage: float = 80.0, current_mode: int = 7, ON_time: float = 2.0, OFF_time: float = 33.0):
        self.target_voltage, self.Kp, self.Ki = float(target_voltage), float(Kp), float(Ki)
        self.generator_voltage, self.current_mode = float(generator_voltage), int(current_mode)
        self.ON_time, self.OFF_time = float(ON_time), float(OFF_time)
        self._env = None

    def bind(self, env: WireEDMEnv, trace=None):
        """Attach to `env`.  Pass an existing `DeviceTrace` that records ``"voltage"`` of every
        environment every microsecond with capacity >= 1001 to share it with a logger; otherwise
        the controller binds its own."""
        if trace is None:
            trace = env.bind_trace(["voltage"], every=1, capacity=self.WINDOW)
        if ("voltage" not in trace.signals or trace.every != 1 or trace.capacity < self.WINDOW
                or trace.env_lo != 0 or trace.env_count != env.num_envs):
            raise ValueError("VoltageController needs a trace of 'voltage' for all environments, every=1, capacity>=1001")
        self._env, self._trace = env, trace
        self.integral_error = torch.zeros(env.num_envs, dtype=torch.float64, device=env.device)
        self._base = env.make_action(0.0, self.generator_voltage, self.current_mode, self.ON_time, self.OFF_time)
        return self

    def average_voltage(self) -> torch.Tensor:
        n = min(self._trace.count, self.WINDOW)
        if n == 0:
            return self._env.state.voltage.clone()
        total = self._trace.read(last=n, names=["voltage"])["voltage"].sum(dim=0)
        # tensor / tensor: torch's GPU kernel for tensor / python-scalar multiplies by 1/n instead
        return total / torch.full_like(total, float(n))

    def __call__(self, env: WireEDMEnv) -> DeviceAction:
        if self._env is not env:
            self.bind(env)
        error = self.target_voltage - self.average_voltage()
        self.integral_error = torch.clamp(self.integral_error + error, -100.0, 100.0)
        pi_output = -(self.Kp * error + self.Ki * self.integral_error * 0.001)
        if env.mechanics_control_mode == "position":
            delta = torch.clamp(pi_output, -5.0, 5.0)
        else:
            delta = torch.clamp(pi_output * 100.0, -1000.0, 1000.0)
        servo = delta.to(torch.float32).to(torch.float64).contiguous()
        b = self._base
        return DeviceAction(servo, b.target_voltage, b.on_time, b.off_time, b.current_mode)


def run_controlled(env: WireEDMEnv, controller: Callable[[WireEDMEnv], DeviceAction], n_steps: int,
                   on_control_step: Optional[Callable[[WireEDMEnv, int], None]] = None, logger=None) -> int:
    """The driver loop of experiments/run_simulation.py:241-297 with fused launches.

    The reference computes the first action before any step, latches it on the first control
    step (call number ``servo_interval + 1``), recomputes the action right after every control
    step and latches THAT one a whole interval later.  Launch lengths reproduce exactly this:
    ``servo_interval - time_since_servo + 1`` microseconds up to and including the next latch,
    then one ``servo_interval`` per launch.  Returns the number of microseconds run.

    ``logger``: a `SimulationLogger`.  With the ``control_step`` frequency it samples the state
    after every launch that ended on a control step; with ``every_step`` / ``interval`` it must be
    attached to the environment's device trace (`logger.attach(env)`), whose ring is drained after
    every launch — the per-microsecond log of run_simulation.py:257 without leaving the GPU.
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
        if logger is not None:
            logger.collect_launch(env.state, control_step=(k == next_k))
        if k < next_k:
            break  # n_steps ran out between two control steps
        action = controller(env)  # the launch ended on a control step (run_simulation.py:272-281)
        if on_control_step is not None:
            on_control_step(env, done)
        next_k = interval
    return done
