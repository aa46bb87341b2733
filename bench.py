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

Prints ONE JSON line (rank 0).  ``roofline`` prices the dominant kernel against the
8 TB/s HBM peak with the ALGORITHMIC bytes of SURVEY.md §8d, B(S) = 8*S + 208 per
env-step; ``cpu_baseline`` times the CPU oracle (oracle/, OpenMP) on this host.
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
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 global-memory, 2 LDS predicated, 3 LDS fused, 4 LDS fused + packed f32, 5 global-memory split")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per environment in kernel 3 (0 auto)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gap", type=float, default=None,
                    help="side measurement: start from this gap [um] instead of the reset state's 50 um (a 15 um gap "
                         "sparks about every 90 us per environment, so the general scalar path runs on most steps)")
    ap.add_argument("--trace", choices=["off", "voltage", "signals"], default="off",
                    help="side measurement: cost of the in-kernel signal trace (every microsecond, all environments): "
                         "'voltage' = the 1 ms ring the voltage controller needs, 'signals' = the 11 scalar signals "
                         "of the reference's logger")
    ap.add_argument("--trace-every", type=int, default=1, help="sample period of --trace in microseconds")
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


def measured_traffic(kernel_name):
    """HBM bytes per launch from the separate rocprofv3 --pmc passes recorded under profiles/
    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE); null when
    no recorded pass matches the kernel that ran."""
    path = ROOT / "profiles" / "traffic.json"
    if not path.exists():
        return None
    try:
        for row in json.loads(path.read_text()):
            if kernel_name.startswith(row["kernel_prefix"]):
                return row["hbm_bytes_per_launch"]
    except Exception:
        return None
    return None


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
                         workpiece_height=h, wire_diameter=d,
                         config=EnvironmentConfig(target_cutting_distance=5000.0))
    else:
        mode = 5
        env = WireEDMEnv(num_envs=n_local, device=device, wire_params=wire, env_id_offset=rank * n_local)
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
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / args.steps

    done = int(env.state.done.sum().item())
    sparks = int(env.state.spark_count.sum().item())
    if rank == 0:
        total_env_steps = world * n_local * n_sub * args.steps
        value = total_env_steps / elapsed
        bytes_per_launch = n_local * n_sub * algorithmic_bytes_per_env_step(S)
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec at batch 65536", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 wire temperature + f64 scalar state", "data": "synthetic",
            "config": {
                "workload": f"{wl_name}: num_envs={n_local} per GPU, n_segments={S}, fresh reset(seed=1234), "
                            f"constant action servo 0.1 / 80 V / I5 / ON 3 / OFF 80",
                "substeps_per_step": n_sub, "global_num_envs": world * n_local,
                "parallelism": (f"env-sharded x{world}, obs all-gather per control step (async, overlapped with the next "
                                "launch)") if world > 1 else "single GPU",
                "kernel": env._backend.last_kernel(),
                **({"trace": args.trace} if args.trace != "off" else {}),
                **({"initial_gap_um": args.gap} if args.gap is not None else {}),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": args.traffic if args.traffic is not None else measured_traffic(env._backend.last_kernel()),
                "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                "note": "algorithmic bytes B(S)=8S+208 per env-step x envs x substeps per launch; the fused "
                        "kernel keeps T in LDS, so physical HBM traffic is ~1/substeps of this",
            },
            "check": {"envs_done": done, "sparks": sparks},
        }
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
