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

    The reference's driver appends `state.voltage` after every step and keeps the samples whose time
    is >= time - 1000 us (run_simulation.py:262-270): with the default 1000-us servo interval that is
    this control interval's samples plus the previous control step's — up to 1001 of them.  The step
    kernels keep exactly that as a running float64 sum in step order and publish it at every control
    step (rows WEDM_F_VOLT_ACC / WEDM_F_VOLT_SUM, include/wedm_hip.h), so the controller reads one
    float64 per environment; no per-microsecond ring is written or re-read.  Sum and division are
    float64; when the voltages are integers (the 0 / 24 / 80 V of the default generator settings) the
    sum is exact in any order and the command equals the reference's `np.mean` bit for bit, otherwise
    it may differ in the last bit before the float32 rounding of the servo leaf.  Before the first
    control step the reference falls back to `state.voltage` (None -> 0); the published sum is 0 then.

    Servo intervals that divide 1000 us are handled by adding the last ``1000 / interval`` published
    sums (their shared end points counted once; environments must then run in lock-step, i.e. without
    per-environment resets).  For any other interval pass ``use_ring=True``: the controller then
    averages a per-microsecond voltage trace (`WireEDMEnv.bind_trace`), as in round 1."""

    WINDOW_US = 1000  # run_simulation.py:267 `cutoff_time = env.state.time - 1000.0`

    def __init__(self, target_voltage: float = 30.0, *, Kp: float = 0.05, Ki: float = 0.1,
                 generator_voltage: float = 80.0, current_mode: int = 7, ON_time: float = 2.0, OFF_time: float = 33.0,
                 use_ring: bool = False):
        self.target_voltage, self.Kp, self.Ki = float(target_voltage), float(Kp), float(Ki)
        self.generator_voltage, self.current_mode = float(generator_voltage), int(current_mode)
        self.ON_time, self.OFF_time = float(ON_time), float(OFF_time)
        self.use_ring = bool(use_ring)
        self._env = None

    def bind(self, env: WireEDMEnv, trace=None):
        """Attach to `env`.  ``trace`` (ring mode only): an existing `DeviceTrace` that records
        ``"voltage"`` of every environment every step with capacity >= the window, to share it with
        a logger; passing one selects the ring mode."""
        if getattr(env, "autoreset", False) and not (self.use_ring or trace is not None):
            # The running-sum form decides "no control step yet" from the batch-wide step count and adds sums published in
            # lock-step: an environment re-initialised inside a launch (VOLT_SUM = VOLT_ACC = 0, its control steps shifted
            # against the batch) would be averaged over sums that are not its own and wind up its integrator.
            raise ValueError("VoltageController's running-sum form needs environments in lock-step: it cannot drive an "
                             "environment built with autoreset=True (use use_ring=True, or reset from the host)")
        self._env = env
        steps = -(-env.servo_interval // env.dt)              # physics steps between two control steps
        period_us = steps * env.dt
        self._window = self.WINDOW_US // env.dt + 1           # samples with time >= t - 1000
        if trace is not None:
            self.use_ring = True
        if not self.use_ring and self.WINDOW_US % period_us:
            raise ValueError(f"the control period ({period_us} us) does not divide the {self.WINDOW_US}-us averaging "
                             "window: use VoltageController(use_ring=True)")
        self._trace = None
        if self.use_ring:
            if trace is None:
                trace = env.bind_trace(["voltage"], every=1, capacity=self._window)
            if ("voltage" not in trace.signals or trace.every != 1 or trace.capacity < self._window
                    or trace.env_lo != 0 or trace.env_count != env.num_envs):
                raise ValueError(f"VoltageController needs a trace of 'voltage' for all environments, every=1, "
                                 f"capacity>={self._window}")
            self._trace = trace
        self._steps, self._m = steps, max(1, self.WINDOW_US // period_us)
        self._sums, self._ends = [], []                       # last m published sums / control-step voltages
        self._seen_steps = -1
        self.integral_error = torch.zeros(env.num_envs, dtype=torch.float64, device=env.device)
        self._base = env.make_action(0.0, self.generator_voltage, self.current_mode, self.ON_time, self.OFF_time)
        return self

    def average_voltage(self) -> torch.Tensor:
        """Mean of the samples `create_voltage_controller` would receive now (call it right after a
        control step, as the driver does)."""
        env = self._env
        if self._trace is not None:
            n = min(self._trace.count, self._window)
            if n == 0:
                return env.state.voltage.clone()
            total = self._trace.read(last=n, names=["voltage"])["voltage"].sum(dim=0)
            # tensor / tensor: torch's GPU kernel for tensor / python-scalar multiplies by 1/n instead
            return total / torch.full_like(total, float(n))
        st = env.state
        if env.steps_since_reset <= self._steps:               # no control step yet: `state.voltage` (None -> 0)
            return st.voltage.clone()
        if self._m == 1:
            total = st.voltage_sum_at_control_step
            return total / torch.full_like(total, float(self._steps + 1))
        if env.steps_since_reset != self._seen_steps:           # a new control step has been published
            self._seen_steps = env.steps_since_reset
            self._sums.append(st.voltage_sum_at_control_step.clone())
            self._ends.append(st.voltage.clone())
            del self._sums[:-self._m], self._ends[:-self._m]
        total = self._sums[0].clone()
        for k in range(1, len(self._sums)):
            total = total + self._sums[k] - self._ends[k - 1]   # the shared end point is in both sums
        n = len(self._sums) * self._steps + 1
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
    ``ceil((servo_interval - time_since_servo) / dt) + 1`` physics steps up to and including the next
    latch, then ``ceil(servo_interval / dt)`` per launch (``n_steps`` and the return value count
    physics steps of ``config.dt`` microseconds each).

    ``logger``: a `SimulationLogger`.  With the ``control_step`` frequency it samples the state
    after every launch that ended on a control step; with ``every_step`` / ``interval`` it must be
    attached to the environment's device trace (`logger.attach(env)`), whose ring is drained after
    every launch — the per-microsecond log of run_simulation.py:257 without leaving the GPU.
    """
    dt = env.dt
    interval = -(-env.servo_interval // dt)             # physics steps between two latches: ceil(servo_interval / dt)
    tss = int(env.state.time_since_servo.max().item())  # [us] one host read, before the loop
    action = controller(env)
    done = 0
    next_k = max(1, -(-(env.servo_interval - tss) // dt) + 1)  # steps up to and including the next latch
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
