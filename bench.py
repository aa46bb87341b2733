#!/usr/bin/env python
"""Headline benchmark: env-steps/s of the batched Wire-EDM step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One bench "step" = one control interval = ``--substeps`` (default 1000) physics
microseconds for every environment, i.e. one fused kernel launch per rank, followed
(N > 1) by the RCCL all-gather of the control-step observations.  Workload at any N:
BASELINE.json configs[2] per GPU (num_envs 65 536, 128-segment wire grid,
segment_len 0.625), fresh reset, constant quickstart action ->
weak scaling.  ``--workload config2`` selects num_envs 4096 / 400 segments instead.

Prints ONE JSON line (rank 0).

``roofline`` names the resource that binds the dominant kernel.  The fused launches keep the wire
on the chip for 1000 us (in the lanes' registers: wedm_step_regs, the kernel of the headline batch;
in LDS: the kernels of longer wires and smaller batches), so HBM sees each byte once per launch
(~0.3 % of peak) and the kernel is bound by vector-ALU ISSUE: ``achieved`` = wave-level VALU instructions per launch (SQ_INSTS_VALU from a
separate rocprofv3 --pmc pass of this same command, recorded in profiles/valu.json) / the kernel
duration measured live with HIP events on the launch stream; ``peak`` = 256 CUs x 4 SIMDs x
2.4 GHz / 2 cycles per wave64 instruction = 1.2288e12 wave-instr/s (MI355X_MICROARCH.md: a wave64
VALU instruction occupies its SIMD-32 for 2 cycles).  The same object carries the physical HBM
fraction (PMC bytes / duration / 8 TB/s), the useful-FP32 fraction ((14 S + 120) flop per env-step
against 157.3 TFLOP/s) and, labelled as an equivalent, the SURVEY.md §8d algorithmic-HBM figure
B(S) = 8 S + 208 bytes per env-step (what an unfused one-launch-per-microsecond implementation
would have to move).  ``side`` holds side measurements of the same build and run: the reference's
own one-launch-per-microsecond cadence (there HBM IS the roof), a densely sparking start (15 um
gap), the closed loop of the reference's driver with its PI voltage controller on the device, an
autoreset batch, the same batch with frozen environments, a policy in the loop (a fresh dict of device
tensors per control step through WireEDMVectorEnv.step), and the other single-GPU BASELINE workloads
(configs[1], the per-GPU shards of configs[3] and configs[4]) each with its own ``roofline`` block.
With N > 1 ranks ``per_rank`` lists every rank's own kernel and wall time per step (min / max / slowest rank).
``cpu_baseline`` times the CPU oracle (oracle/, OpenMP) on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 per CU, 2.4 GHz max clock, a wave64 VALU instruction issues over 2 cycles
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0  # wave-level VALU instructions per second


def useful_flop_per_env_step(n_seg: int) -> int:
    """~14 flop per wire cell + ~120 flop of scalar physics per env-step."""
    return 14 * n_seg + 120


def algorithmic_bytes_per_env_step(n_seg: int) -> int:
    """read+write T (2*4*S) + read+write the 96-B scalar block + 16 B action."""
    return 8 * n_seg + 208


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--substeps", type=int, default=1000, help="physics microseconds per bench step")
    ap.add_argument("--workload", choices=["config3", "config2", "config4", "config5"], default="config3")
    ap.add_argument("--num-envs", type=int, default=0, help="override environments per GPU")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 global-memory, 2 LDS any-geometry (packed), 9 served packed, 10 LDS any-geometry cell by cell, 11 served any-geometry, 12 served registers, 3 LDS fused, 4 LDS fused + packed f32, 5 global-memory split, 6 stream (single microseconds), 7 registers (one or two lanes per environment), 8 wide registers (4 / 8 / 16 lanes per environment)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per environment in kernel 3 (0 auto)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements (1-us cadence, dense sparking, closed loop)")
    ap.add_argument("--gap", type=float, default=None,
                    help="side measurement: start from this gap [um] instead of the reset state's 50 um (a 15 um gap "
                         "sparks about every 90 us per environment, so the general scalar path runs on most steps)")
    ap.add_argument("--trace", choices=["off", "voltage", "signals"], default="off",
                    help="side measurement: cost of the in-kernel signal trace (every microsecond, all environments): "
                         "'voltage' = the 1 ms ring the voltage controller needs, 'signals' = the 11 scalar signals "
                         "of the reference's logger")
    ap.add_argument("--trace-every", type=int, default=1, help="sample period of --trace in microseconds")
    ap.add_argument("--stencil-dtype", choices=["float32", "float64"], default="float32",
                    help="side measurement: the wire stencil as Numba types it (float64 expressions rounded at each float32 store)")
    ap.add_argument("--traffic", type=float, default=None,
                    help="measured HBM bytes per launch from a separate rocprofv3 --pmc pass (else null)")
    return ap.parse_args()


def workload(args):
    from sparc_amd import WireModuleParameters

    if args.workload == "config3":
        n, wire, name = 65536, WireModuleParameters(segment_len=0.625), "BASELINE configs[2]"
    elif args.workload == "config2":
        n, wire, name = 4096, WireModuleParameters(), "BASELINE configs[1]"
    elif args.workload == "config4":
        n, wire, name = 32768, WireModuleParameters(), "BASELINE configs[3] (per-GPU shard)"
    else:
        n, wire, name = 16384, WireModuleParameters(), "BASELINE configs[4] (per-GPU shard, per-env geometry)"
    if args.num_envs:
        n = args.num_envs
    return n, wire, name


def config5_draws(n_global, lo, hi):
    """per-env workpiece_height ~ U[10,30] mm, wire_diameter and
    current_mode from fixed sets, drawn host-side from numpy.default_rng(2024)."""
    import numpy as np

    rng = np.random.default_rng(2024)
    h = rng.uniform(10.0, 30.0, n_global)
    d = rng.choice([0.10, 0.15, 0.20, 0.25, 0.30], n_global)
    mode = rng.choice([1, 3, 5, 7, 9, 11, 13, 15, 17], n_global).astype(np.int32)
    return h[lo:hi], d[lo:hi], mode[lo:hi]


def cpu_baseline(wire_params, n_envs, n_sub, target_seconds):
    """Time the CPU oracle (oracle/, the checker) on a bounded sample of the same workload:
    the same batch, as many control intervals as fit in ~target_seconds, all host cores."""
    from oracle import oracle as orc
    from sparc_amd import WireEDMEnv
    from tests._oracle_backend import OracleBackend

    threads = int(orc.lib().wedm_oracle_max_threads())
    n_envs = min(n_envs, 65536)
    env = WireEDMEnv(num_envs=n_envs, device="cpu", backend=OracleBackend, wire_params=wire_params)
    env.reset(seed=1234)
    act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
    t0 = time.perf_counter()
    env.step_many(act, n_sub)  # also the probe that sizes the sample
    probe = time.perf_counter() - t0
    intervals = max(1, min(200, int(target_seconds / max(probe, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(intervals):
        env.step_many(act, n_sub)
    dt = time.perf_counter() - t0
    return {
        "value": n_envs * n_sub * intervals / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
        "sample": f"{n_envs} envs x {intervals} control intervals x {n_sub} us of the same workload "
                  f"(after one warm-up interval), OpenMP static over envs, {dt:.1f} s",
    }


def recorded(kernel_name, file_name):
    """Row of profiles/<file_name> whose kernel_prefix matches the kernel that ran (separate rocprofv3
    --pmc passes of this same command, see tools/profile_round.sh), or None."""
    path = ROOT / "profiles" / file_name
    if not path.exists():
        return None
    try:
        for row in json.loads(path.read_text()):
            if kernel_name.startswith(row["kernel_prefix"]):
                return row
    except Exception:
        return None
    return None


def recorded_for_this_build(kernel_name, file_name):
    """(row, None) when profiles/<file_name> holds counters of the kernel that ran, counted on THIS build of the library
    (the row's `build_id` equals `wedm_build_id()` of the loaded library: sha256 over the kernel sources and flags);
    (None, reason) otherwise -- a kernel edit that keeps its name and launch geometry must not price the new duration
    with the old instruction count."""
    from sparc_amd import _lib

    row = recorded(kernel_name, file_name)
    if row is None:
        return None, f"no recorded pass matches this kernel (profiles/{file_name})"
    have, want = row.get("build_id"), _lib.build_id()
    if have != want:
        return None, (f"profiles/{file_name} was counted on build {have}, the loaded library is build {want}: "
                      "re-run tools/profile_round.sh on this build")
    return row, None


def measured_traffic(kernel_name):
    """(HBM bytes per launch, None) from the separate rocprofv3 --pmc passes recorded under profiles/
    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE), or (None, reason)
    when no recorded pass of this build matches the kernel that ran."""
    row, why = recorded_for_this_build(kernel_name, "traffic.json")
    return (row["hbm_bytes_per_launch"], None) if row else (None, why)


def resident_waves_cap(backend):
    """Waves of the last launch's kernel that can be resident per SIMD: blocks per CU by the occupancy API x waves per block / 4
    (the served kernels run three blocks of four waves per CU, the register / LDS kernels two)."""
    try:
        block = int(backend.last_kernel().split("<<<")[1].split(">>>")[0].split(",")[1])
        occ = backend.last_occupancy()
        return max(1.0, occ * (block // 64) / 4.0) if occ > 0 else 2.0
    except Exception:
        return 2.0


def roofline_block(kernel_name, kernel_ms, n_envs, n_sub, n_seg, traffic_override=None, resident_cap=2.0):
    """The `roofline` object for one kernel launch of `n_envs` x `n_sub` env-steps that took `kernel_ms`."""
    t = kernel_ms * 1e-3
    env_steps = n_envs * n_sub
    alg_bytes = env_steps * algorithmic_bytes_per_env_step(n_seg)
    traffic, traffic_why = (traffic_override, None) if traffic_override is not None else measured_traffic(kernel_name)
    alg = {"GB/s": alg_bytes / t / 1e9, "frac_of_hbm_peak": alg_bytes / t / 1e9 / HBM_PEAK_GBS,
           "bytes_per_launch": alg_bytes,
           "note": "B(S) = 8 S + 208 bytes per env-step (SURVEY.md §8d) x env-steps per launch"}
    fp32 = {"TFLOP/s": env_steps * useful_flop_per_env_step(n_seg) / t / 1e12, "peak": FP32_VECTOR_PEAK_TFLOPS}
    fp32["frac"] = fp32["TFLOP/s"] / FP32_VECTOR_PEAK_TFLOPS
    hbm = None
    if traffic is not None:
        hbm = {"GB/s": traffic / t / 1e9, "peak": HBM_PEAK_GBS, "frac": traffic / t / 1e9 / HBM_PEAK_GBS,
               "bytes_per_launch": traffic}
    if n_sub == 1:
        # one launch per microsecond: every byte crosses HBM once per launch -> HBM is the roof, priced with the
        # algorithmic bytes as SURVEY.md §8d defines them
        out = {"bound": "hbm", "achieved": alg["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": alg["frac_of_hbm_peak"], "traffic": traffic, "kernel_ms": kernel_ms,
               "algorithmic_bytes_per_launch": alg_bytes, "hbm_physical": hbm, "fp32_useful": fp32}
        if traffic is None:
            out["traffic_note"] = traffic_why
        return out
    valu, valu_why = recorded_for_this_build(kernel_name, "valu.json")
    out = {"bound": "valu-issue", "unit": "wave-instr/s", "peak": VALU_ISSUE_PEAK, "traffic": traffic,
           "kernel_ms": kernel_ms, "hbm_physical": hbm, "fp32_useful": fp32, "algorithmic_hbm_equivalent": alg}
    if traffic is None:
        out["traffic_note"] = traffic_why
    if valu is not None:
        insts = valu["valu_insts_per_launch"] * (env_steps / valu["env_steps_per_launch"])
        out["achieved"] = insts / t
        out["frac"] = out["achieved"] / VALU_ISSUE_PEAK
        out["valu_insts_per_env_step"] = valu["valu_insts_per_launch"] / valu["env_steps_per_launch"]
        out["source"] = valu["source"]
        if "sq_active_inst_valu" in valu and "sq_wave_cycles" in valu and "sq_waves" in valu:
            # SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave's VALU instruction occupies its SIMD (packed-f32 and
            # f64 instructions hold it twice as long as the 2-cycle f32 instruction the issue peak assumes);
            # SQ_WAVE_CYCLES counts the quad-cycles the launch's waves were resident.  Both from the SAME counter pass,
            # so their ratio is the pipe occupancy at the clock the chip really ran at, whatever that was.
            # waves RESIDENT per SIMD: what the occupancy API admits for the kernel that ran (`resident_cap`: 2 for the register /
            # LDS kernels, 3 for the served kernels); a launch with more waves than that runs them in rounds
            resident = min(max(1.0, valu["sq_waves"] / 1024.0), resident_cap)
            simd_cycles = valu["sq_wave_cycles"] / resident                  # quad-cycles a SIMD was occupied by the launch
            scale = env_steps / valu["env_steps_per_launch"]
            out["valu_pipe_busy"] = {
                "frac": valu["sq_active_inst_valu"] / simd_cycles,
                "measured_clock_GHz": valu["sq_wave_cycles"] * scale * 4.0 / (1024.0 * resident) / t / 1e9,
                "note": "SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / resident waves per SIMD), one counter pass; the clock is "
                        "SQ_WAVE_CYCLES x 4 / waves / the live kernel duration (the 2.4 GHz of the issue peak is the "
                        "data-sheet maximum, under this load the chip runs lower)"}
    else:
        out["achieved"] = None
        out["frac"] = None
        out["source"] = valu_why
    where = "in the lanes' registers" if "wedm_step_regs" in kernel_name else "in LDS"
    out["note"] = (f"the fused launch keeps the wire {where} for all its microseconds: HBM sees each byte once per launch, "
                   "the kernel is bound by VALU issue; frac = wave-level VALU instructions per second / (1024 SIMDs x 2.4 GHz / 2)")
    if "wedm_step_regs_wide" in kernel_name:
        # the batch gives the chip at most one wave per SIMD: nothing overlaps with a wave's own dependent chain
        out["note"] += ("; this batch is one round of blocks at ONE wave per SIMD, so the launch lasts as long as a wave's "
                        "dependent chain per microsecond (prelude -> walk -> epilogue: DESIGN.md 4.1b) -- the occupancy of the "
                        "VALU pipe (valu_pipe_busy) and the algorithmic-HBM equivalent say how much of the chip that leaves idle")
    return out


def time_launches(env, act, n_sub, steps, warmup):
    """(seconds per launch, kernel name): HIP events on the launch stream around `steps` back-to-back launches."""
    import torch

    for _ in range(warmup):
        env.step_many(act, n_sub)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        env.step_many(act, n_sub)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / steps, env._backend.last_kernel()


def side_measurements(n_local, wire, S, device):
    """Five side measurements of the same build at the bench batch (single GPU only):
      * the reference's own cadence, one launch per microsecond (wedm_step(n_substeps=1));
      * a densely sparking start (15 um gap: ~5.6 sparks per environment per ms instead of ~0.7);
      * the closed loop of experiments/run_simulation.py with ITS PI voltage controller evaluated on the device
        from the kernel-side running voltage sum (steady state after a 100 ms approach);
      * a batch with in-launch autoreset whose environments keep terminating at different times;
      * the same batch without autoreset: terminated environments stay frozen among the live ones."""
    import torch

    from sparc_amd import VoltageController, WireEDMEnv, run_controlled

    out = []
    env = WireEDMEnv(num_envs=n_local, device=device, wire_params=wire)
    env.reset(seed=1234)
    act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
    sec, kname = time_launches(env, act, 1, 1000, 100)
    out.append({"name": "one launch per microsecond (the reference's step() cadence)", "value": n_local / sec,
                "unit": "env-steps/s", "kernel": kname, "roofline": roofline_block(kname, sec * 1e3, n_local, 1, S)})
    env.reset(seed=1234)
    env.state.wire_position = 10.0
    env.state.workpiece_position = 25.0
    env.state.target_position = 5000.0
    s0 = int(env.state.spark_count.sum().item())
    sec, kname = time_launches(env, act, 1000, 8, 2)
    sparks = (int(env.state.spark_count.sum().item()) - s0) / n_local / 10.0
    out.append({"name": "fresh reset moved to a 15 um gap (dense sparking)", "value": n_local * 1000 / sec,
                "unit": "env-steps/s", "kernel": kname, "sparks_per_env_per_ms": sparks,
                "roofline": roofline_block(kname, sec * 1e3, n_local, 1000, S)})
    env.reset(seed=1)
    env.state.workpiece_position = 70.0      # experiments/run_simulation.py:199-201
    env.state.wire_position = 10.0
    env.state.target_position = 5000.0
    ctl = VoltageController(30.0)
    # ONE driver loop: a 100 ms approach (the controller needs ~90 ms to close the 60 um gap), then 10 timed control
    # intervals, bracketed by HIP events recorded from the loop's own control-step hook (two calls of run_controlled in a
    # row would evaluate the controller twice for the sample between them: one extra PI-integrator update).
    marks = {}

    def mark(e, done):
        if done in (100 * 1000 + 1, 110 * 1000 + 1):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks[done] = (ev, e.state.spark_count.sum())  # (device-side sum: no host sync inside the loop)

    run_controlled(env, ctl, 110 * 1000 + 1, on_control_step=mark)
    torch.cuda.synchronize()
    (ev0, s0), (ev1, s1) = marks[100 * 1000 + 1], marks[110 * 1000 + 1]
    dt = ev0.elapsed_time(ev1) * 1e-3
    gap = float((env.state.workpiece_position - env.state.wire_position).mean().item())
    out.append({"name": "closed loop: the reference driver's PI voltage controller on the device, steady state",
                "value": n_local * 10 * 1000 / dt, "unit": "env-steps/s", "kernel": env._backend.last_kernel(),
                "ms_per_control_interval": dt / 10 * 1e3, "mean_gap_um": gap,
                "sparks_per_env_per_ms": (int(s1.item()) - int(s0.item())) / n_local / 10.0,
                "timing": "HIP events around 10 control intervals of one driver loop, incl. the controller's torch ops"})
    env.close()
    # a training-style batch: in-launch autoreset and a cutting target a few sparks ahead, so that in steady state a sixth
    # of the environments terminates in every launch and waits, frozen, for the next launch's re-initialisation
    from sparc_amd import EnvironmentConfig

    env = WireEDMEnv(num_envs=n_local, device=device, wire_params=wire, autoreset=True, reward="progress",
                     config=EnvironmentConfig(target_cutting_distance=50.002))
    env.reset(seed=7)
    act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
    for _ in range(20):
        env.step_many(act, 1000)
    torch.cuda.synchronize()
    e0 = int(env.state.episode.sum().item())
    t0 = time.perf_counter()
    for _ in range(20):
        env.step_many(act, 1000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.append({"name": "autoreset batch: environments terminate at different times (cutting target 0.002 um ahead)",
                "value": n_local * 20 * 1000 / dt, "unit": "env-steps/s", "kernel": env._backend.last_kernel(),
                "resets_per_env_per_launch": (int(env.state.episode.sum().item()) - e0) / n_local / 20.0,
                "timing": "wall clock around 20 launches of 1000 us"})
    env.close()
    # the same batch WITHOUT autoreset: terminated environments stay frozen until the host resets them.  The register
    # kernel walks under the mask of the live lanes (frozen lanes keep their registers); a handle on the LDS kernels
    # moves by itself to the instantiation whose tile code tolerates frozen lanes (the first launch that finds one reports
    # it through a host-visible word); before, such a wave fell back to the per-cell predicated path.
    env = WireEDMEnv(num_envs=n_local, device=device, wire_params=wire, config=EnvironmentConfig(target_cutting_distance=50.002))
    env.reset(seed=7)
    act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
    for _ in range(4):
        env.step_many(act, 1000)
    torch.cuda.synchronize()
    frozen0 = int(env.state.done.sum().item())
    sec, kname = time_launches(env, act, 1000, 8, 1)
    frozen1 = int(env.state.done.sum().item())
    live = n_local - 0.5 * (frozen0 + frozen1)
    out.append({"name": "no autoreset, terminated environments stay frozen in the batch (cutting target 0.002 um ahead)",
                "value": live * 1000 / sec, "unit": "env-steps/s (live environments only)", "kernel": kname,
                "all_lanes_equivalent": n_local * 1000 / sec, "frozen_fraction": 0.5 * (frozen0 + frozen1) / n_local,
                "kernel_ms": sec * 1e3})
    env.close()
    return out


def make_workload_env(name, device, rank=0, world=1, n_override=0, **kw):
    """(env, action mode(s), n_local, label) of one BASELINE workload as `main()` builds it for a rank."""
    from sparc_amd import EnvironmentConfig, WireEDMEnv

    class A:  # the fields `workload()` reads
        workload, num_envs = name, n_override

    n_local, wire, label = workload(A)
    if name == "config5":
        h, d, mode = config5_draws(world * n_local, rank * n_local, (rank + 1) * n_local)
        env = WireEDMEnv(num_envs=n_local, device=device, wire_params=wire, env_id_offset=rank * n_local,
                         workpiece_height=h, wire_diameter=d, config=EnvironmentConfig(target_cutting_distance=5000.0), **kw)
    else:
        mode = 5
        env = WireEDMEnv(num_envs=n_local, device=device, wire_params=wire, env_id_offset=rank * n_local, **kw)
    return env, mode, n_local, label


def other_config_lines(device, stencil_dtype="float32"):
    """The other single-GPU BASELINE workloads, timed in the SAME driver run as the headline (their published numbers
    used to exist only as builder runs): configs[1] (4 096 x 400), the per-GPU shard of configs[3] (32 768 x 400) and the
    per-GPU shard of configs[4] (16 384 environments with per-environment geometry), fused 1000-us launches, each with its
    own `roofline` block priced from the profiles/ rows of this build.  ``stencil_dtype="float64"``: the same workloads, and
    configs[2] itself, with the stencil as Numba types the reference's @njit kernel (wire.py:58-123)."""
    import torch

    out = []
    f64 = stencil_dtype != "float32"
    for name, steps in ((("config3", 5),) if f64 else ()) + (("config2", 10), ("config4", 4 if f64 else 6), ("config5", 4 if f64 else 6)):
        env, mode, n_local, label = make_workload_env(name, device, stencil_dtype=stencil_dtype)
        env.reset(seed=1234)
        act = env.make_action(0.1, 80.0, mode, 3.0, 80.0)
        sec, kname = time_launches(env, act, 1000, steps, 2)
        typing = "; stencil_dtype=float64 (float64 expressions rounded at each float32 store, as Numba types wire.py:58-123)" if f64 else ""
        line = {"name": f"{label}: num_envs={n_local}, n_segments={env.n_segments}, fresh reset(seed=1234), fused 1000-us launches{typing}",
                "workload": name + ("_f64" if f64 else ""), "value": n_local * 1000 / sec, "unit": "env-steps/s", "kernel": kname, "kernel_ms": sec * 1e3,
                "steps": steps, "occupancy_blocks_per_cu": env._backend.last_occupancy(),
                "roofline": roofline_block(kname, sec * 1e3, n_local, 1000, env.n_segments, resident_cap=resident_waves_cap(env._backend)),
                "check": {"envs_done": int(env.state.done.sum().item()), "sparks": int(env.state.spark_count.sum().item())}}
        out.append(line)
        env.close()
        del env
        torch.cuda.empty_cache()
    return out


def policy_in_the_loop_line(n_local, wire, device):
    """The contract an RL loop uses (wire_edm.py:116-121,162-170): per control step a FRESH dict of device tensors --
    the servo command a small torch function of the observation, the current modes drawn on the device -- through
    `WireEDMVectorEnv.step` on the autoreset batch of the fourth side line.  No device-to-host read anywhere (the modes
    are validated on the device); what it costs over the constant pre-built action is the policy's own torch ops."""
    import torch

    from sparc_amd import EnvironmentConfig, WireEDMEnv, WireEDMVectorEnv

    env = WireEDMEnv(num_envs=n_local, device=device, wire_params=wire, autoreset=True, reward="progress",
                     config=EnvironmentConfig(target_cutting_distance=50.002))
    vec = WireEDMVectorEnv(env)
    obs, _ = vec.reset(seed=7)
    valid = torch.tensor([5, 5, 5, 3, 7], dtype=torch.int32, device=device)
    gen = torch.Generator(device=device).manual_seed(11)
    volt = torch.full((n_local,), 80.0, device=device)

    def policy(o):
        gap = o[:, 0]
        return {"servo": torch.clamp(0.1 + 0.002 * (gap - 50.0), -1.0, 1.0),
                "generator_control": {"target_voltage": volt, "ON_time": torch.full((n_local,), 3.0, device=device),
                                      "OFF_time": torch.full((n_local,), 80.0, device=device),
                                      "current_mode": valid[torch.randint(0, 5, (n_local,), device=device, generator=gen)]}}

    for _ in range(20):
        obs, reward, term, trunc, info = vec.step(policy(obs))
    torch.cuda.synchronize()
    e0 = int(env.state.episode.sum().item())
    prev = torch.cuda.get_sync_debug_mode()
    torch.cuda.set_sync_debug_mode("error")  # a synchronising call inside the loop fails the bench, loudly
    try:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            obs, reward, term, trunc, info = vec.step(policy(obs))
        b.record()
    finally:
        torch.cuda.set_sync_debug_mode(prev)
    torch.cuda.synchronize()
    dt = a.elapsed_time(b) * 1e-3
    env.check_errors()
    line = {"name": "policy in the loop: a fresh dict of device tensors per control step through WireEDMVectorEnv.step "
                    "(servo = f(obs), current modes drawn on the device), autoreset batch",
            "value": n_local * 20 * 1000 / dt, "unit": "env-steps/s", "kernel": env._backend.last_kernel(),
            "ms_per_control_interval": dt / 20 * 1e3,
            "resets_per_env_per_launch": (int(env.state.episode.sum().item()) - e0) / n_local / 20.0,
            "timing": "HIP events around 20 control intervals under torch's sync-debug mode 'error' (no host read in the loop)"}
    env.close()
    return line


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from sparc_amd import WireEDMEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # WEDM_BENCH_FORCE_DIST=1 exercises the RCCL code path even with one rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("WEDM_BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=device)

    n_local, wire, wl_name = workload(args)
    n_sub = args.substeps
    if args.workload == "config5":
        from sparc_amd import EnvironmentConfig

        h, d, mode = config5_draws(world * n_local, rank * n_local, (rank + 1) * n_local)
        env = WireEDMEnv(num_envs=n_local, device=device, wire_params=wire, env_id_offset=rank * n_local,
                         workpiece_height=h, wire_diameter=d, stencil_dtype=args.stencil_dtype,
                         config=EnvironmentConfig(target_cutting_distance=5000.0))
    else:
        mode = 5
        env = WireEDMEnv(num_envs=n_local, device=device, wire_params=wire, env_id_offset=rank * n_local,
                         stencil_dtype=args.stencil_dtype)
    env.set_kernel(args.kernel, args.lanes)
    if args.trace == "voltage":
        env.bind_trace(["voltage"], every=args.trace_every, capacity=1001)
    elif args.trace == "signals":  # experiments/run_simulation.py:127-139 (scalar signals)
        env.bind_trace(["time", "voltage", "current", "wire_position", "wire_velocity", "workpiece_position",
                        "target_delta", "debris_concentration", "flow_rate", "is_short_circuit"], every=args.trace_every, capacity=1000)
    env.reset(seed=1234)
    if args.gap is not None:
        env.state.wire_position = 10.0
        env.state.workpiece_position = 10.0 + args.gap
        env.state.target_position = 5000.0
    act = env.make_action(0.1, 80.0, mode, 3.0, 80.0)
    S = env.n_segments
    obs_local = env.state.obs[:, :n_local]
    gather = None
    if use_dist:  # observations of every shard, once per control step, over xGMI, overlapped with the next launch
        from sparc_amd.parallel import PipelinedObsGather

        gather = PipelinedObsGather(obs_local, world)

    def one_step():
        env.step_many(act, n_sub)
        if use_dist:
            gather.post()

    for _ in range(args.warmup):
        one_step()

    # ---- timed region: exactly K steps between barrier + synchronize
    # HIP events on torch's current stream == the stream wedm_step launches on.  One pair per step;
    # with many short steps (single-microsecond launches) one pair around the whole region, because
    # two event records per 30-us launch would themselves slow the stream down.
    per_step_events = args.steps <= 200
    n_ev = args.steps if per_step_events else 1
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not per_step_events:
        starts[0].record()
    for i in range(args.steps):
        if per_step_events:
            starts[i].record()
        env.step_many(act, n_sub)
        if per_step_events:
            ends[i].record()
        if use_dist:
            gather.post()
    if not per_step_events:
        ends[0].record()
    gathered = gather.result() if use_dist else None  # the last gather is inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / args.steps
    # every rank's own kernel time and wall time (the first multi-GPU run should show stragglers, not only the maximum)
    per_rank = None
    if use_dist:
        mine = torch.tensor([kernel_ms, elapsed_local * 1e3 / args.steps], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        km = [float(x[0].item()) for x in allr]
        per_rank = {"kernel_ms": km, "kernel_ms_min": min(km), "kernel_ms_max": max(km), "slowest_rank": km.index(max(km)),
                    "wall_ms_per_step": [float(x[1].item()) for x in allr]}

    done = int(env.state.done.sum().item())
    broken = int(env.state.is_wire_broken.sum().item())
    reached = int(env.state.is_target_distance_reached.sum().item())
    sparks = int(env.state.spark_count.sum().item())
    if rank == 0:
        total_env_steps = world * n_local * n_sub * args.steps
        value = total_env_steps / elapsed
        kname = env._backend.last_kernel()
        out = {
            "metric": f"env-steps/sec at batch {world * n_local}", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 wire temperature + f64 scalar state", "data": "synthetic",
            "config": {
                "workload": f"{wl_name}: num_envs={n_local} per GPU, n_segments={S}, fresh reset(seed=1234), "
                            f"constant action servo 0.1 / 80 V / I5 / ON 3 / OFF 80",
                "substeps_per_step": n_sub, "global_num_envs": world * n_local,
                "parallelism": (f"env-sharded x{world}: one process per GPU over RCCL, rank r steps environments "
                                f"[r*{n_local}, (r+1)*{n_local}) with env_id_offset r*{n_local} (global ids in the Philox "
                                "counter), no data-path collective; obs all-gather per control step (async, overlapped "
                                "with the next launch)") if world > 1 else "single GPU",
                "ranks": world, "env_id_offsets": [r * n_local for r in range(world)],
                "kernel": kname, "build_id": env._backend.build_id(), "occupancy_blocks_per_cu": env._backend.last_occupancy(),
                **({"trace": args.trace} if args.trace != "off" else {}),
                **({"stencil_dtype": args.stencil_dtype} if args.stencil_dtype != "float32" else {}),
                **({"initial_gap_um": args.gap} if args.gap is not None else {}),
            },
            "roofline": roofline_block(kname, kernel_ms, n_local, n_sub, S, args.traffic, resident_cap=resident_waves_cap(env._backend)),
            **({"per_rank": per_rank} if per_rank is not None else {}),
            "check": {"envs_done": done, "envs_wire_broken": broken, "envs_target_reached": reached, "sparks": sparks,
                      **({"note": "per-environment draws that pair the thinnest wires (0.10 mm) with the highest current modes "
                                  "overheat the wire (Tmax > breaking temperature, wire.py:376-388) within the window: the "
                                  "physics of those draws, not a bench artefact; such an environment is frozen from then on "
                                  "and still counted in `value` (the kernel carries its lanes)"}
                         if args.workload == "config5" and done else {})},
        }
        if world == 1 and not args.no_side and args.workload == "config3" and args.trace == "off" and args.gap is None \
                and not args.num_envs and args.kernel == 0 and args.stencil_dtype == "float32":
            out["side"] = side_measurements(n_local, wire, S, device)
            out["side"].append(policy_in_the_loop_line(n_local, wire, device))
            out["side"] += other_config_lines(device)
            out["side"] += other_config_lines(device, "float64")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wire, n_local, n_sub, args.cpu_seconds)
            out["cpu_baseline"]["reference_python"] = (
                "1 209 env-steps/s, 1 core: the Python reference itself (Numba stubbed), measured in the build "
                "container (BASELINE.md); it cannot travel to the GPU box")
        print(json.dumps(out))
    if use_dist:
        if rank == 0 and gathered is not None:  # the gathered block really is the local observations
            assert torch.equal(gathered[rank * obs_local.shape[0]: (rank + 1) * obs_local.shape[0]], obs_local), \
                "all-gather returned wrong data"
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
