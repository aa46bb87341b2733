"""world_size-2 and world_size-8 tests of the sharded environment on CPU (gloo): shards reproduce the
single-process batch exactly and the observation all-gather returns the whole batch."""
from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_GLOBAL = 24


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_single():
    from sparc_amd import WireEDMEnv
    from tests._oracle_backend import OracleBackend

    env = WireEDMEnv(num_envs=N_GLOBAL, device="cpu", backend=OracleBackend)
    env.reset(seed=2024)
    env.state.workpiece_position = 25.0
    env.state.wire_position = 10.0
    env.state.target_position = 5000.0
    act = env.make_action(servo=torch.linspace(-0.2, 0.4, N_GLOBAL).double(), current_mode=[1, 3, 5, 7, 9, 11] * 4)
    for _ in range(2):
        env.step_control(act)
    return env


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparc_amd.parallel import ShardedWireEDMEnv
        from tests._oracle_backend import OracleBackend

        torch.set_num_threads(1)
        env = ShardedWireEDMEnv(N_GLOBAL, device="cpu", backend=OracleBackend)
        env.reset(seed=2024)
        env.state.workpiece_position = 25.0
        env.state.wire_position = 10.0
        env.state.target_position = 5000.0
        act = env.make_action(servo=torch.linspace(-0.2, 0.4, N_GLOBAL).double(), current_mode=[1, 3, 5, 7, 9, 11] * 4)
        from sparc_amd.parallel import PipelinedObsGather

        pipe = PipelinedObsGather(env.state.obs[:, : env.num_envs], world)
        for _ in range(2):
            obs_all, *_ = env.step_control(act)
            pipe.post()                       # the bench's overlapped form of the same gather
        piped = pipe.result().view(world, -1, env.num_envs).permute(0, 2, 1).reshape(N_GLOBAL, -1)
        assert torch.equal(piped, obs_all)
        done_all = env.gather_done()
        blocks = env.state.clone_blocks()
        n = env.num_envs
        out[rank] = {"obs_all": obs_all.clone(), "done_all": done_all.clone(), "lo": env.lo,
                     "f64": blocks["f64"][:, :n], "i32": blocks["i32"][:, :n], "T": blocks["T"][:, :n]}
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8])   # 8: the rank layout of the driver's one-node scaling run
def test_sharding_matches_single_process(world):
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = {k: v for k, v in out.items()}
    ref = _run_single()
    n = N_GLOBAL // world
    for rank in range(world):
        r = res[rank]
        sl = slice(r["lo"], r["lo"] + n)
        a, b = r["f64"], ref.state.f64[:, sl]
        assert bool(((a == b) | (a.isnan() & b.isnan())).all()), f"rank {rank}: float64 state differs"
        assert torch.equal(r["i32"], ref.state.i32[:, sl])
        assert torch.equal(r["T"], ref.state.T[:, sl])
        assert torch.equal(r["obs_all"], ref.state.obs[:, :N_GLOBAL].t())   # every rank holds the whole batch
        assert torch.equal(r["done_all"], ref.state.done)
    assert int(ref.state.spark_count.sum()) > 0
