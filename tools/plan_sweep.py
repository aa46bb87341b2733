#!/usr/bin/env python
"""Does `wedm_step`'s automatic kernel choice pick the fastest kernel?  Fused launches (1000 us) over a grid of batch sizes
and wire lengths, the automatic plan against every forced variant that accepts the shape (run ON THE GPU BOX):

    python tools/plan_sweep.py > profiles/r4/plan_sweep.txt          # the whole grid
    python tools/plan_sweep.py 16384,20480 128                       # some batch sizes / wire lengths
    python tools/plan_sweep.py --f64 > profiles/r4/plan_sweep_f64.txt # the stencil in Numba's typing (stencil_mode 1)

`sweep_point()` is also what tests/test_gpu_parity.py::test_automatic_plan_is_within_reach_of_the_best_forced_kernel uses."""
import sys

sys.path.insert(0, ".")

N_GRID = (2048, 4096, 8192, 16384, 20480, 32768, 65536, 131072)
S_GRID = (128, 200, 256, 400, 512)
# (kernel, lanes): the fused-launch kernels of uniform geometry; 0 = the kernel's own lane choice
CANDIDATES = ((3, 0), (3, 8), (3, 16), (4, 0), (4, 4), (4, 8), (7, 0), (8, 0), (9, 8), (9, 4))
CANDIDATES_F64 = ((3, 0), (3, 8), (3, 16), (7, 0), (7, 1), (8, 0), (8, 8), (8, 16))   # what accepts stencil_mode 1
N_GRID_F64 = (4096, 16384, 32768, 65536)


def segment_len(n_seg):
    return 80.0 / n_seg  # default workpiece height 20 mm + two 30 mm buffers (wire.py:144-149)


def time_shape(n, n_seg, kernel, lanes, launches=3, stencil_dtype="float32"):
    """(ms per 1000-us launch, kernel name) or (None, reason)."""
    import torch

    from sparc_amd import WireEDMEnv, WireModuleParameters
    from sparc_amd._lib import WedmError

    env = WireEDMEnv(num_envs=n, device="cuda:0", wire_params=WireModuleParameters(segment_len=segment_len(n_seg)), stencil_dtype=stencil_dtype)
    assert env.n_segments == n_seg, (env.n_segments, n_seg)
    try:
        env.set_kernel(kernel, lanes)
        env.reset(seed=1234)
        act = env.make_action(0.1, 80.0, 5, 3.0, 80.0)
        try:
            env.step_many(act, 1000)
        except WedmError as exc:
            return None, str(exc).split(":")[-1].strip()[:60]
        times = []
        for _ in range(launches):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            env.step_many(act, 1000)
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        return sorted(times)[len(times) // 2], env._backend.last_kernel().split("<<<")[0]
    finally:
        env.close()
        del env
        torch.cuda.empty_cache()


def sweep_point(n, n_seg, launches=3, stencil_dtype="float32"):
    """{'auto': (ms, name), 'best': (ms, name, (kernel, lanes)), 'all': {...}} for one shape."""
    auto = time_shape(n, n_seg, 0, 0, launches, stencil_dtype)
    rows = {}
    for kernel, lanes in (CANDIDATES if stencil_dtype == "float32" else CANDIDATES_F64):
        ms, name = time_shape(n, n_seg, kernel, lanes, launches, stencil_dtype)
        if ms is not None:
            rows[(kernel, lanes)] = (ms, name)
    best_key = min(rows, key=lambda k: rows[k][0])
    return {"auto": auto, "best": (*rows[best_key], best_key), "all": rows}


def main():
    f64 = "--f64" in sys.argv
    if f64:
        sys.argv.remove("--f64")
    dtype = "float64" if f64 else "float32"
    ns = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (N_GRID_F64 if f64 else N_GRID)
    ss = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else S_GRID
    from sparc_amd import _lib

    print(f"# automatic plan vs forced kernels, fused launches of 1000 us, fresh reset(seed=1234), stencil_dtype {dtype}; build {_lib.build_id()}")
    print(f"# {'N':>7s} {'S':>4s}  {'auto: kernel':34s} {'ms':>8s}  {'best forced: kernel':34s} {'ms':>8s}  auto/best")
    worst = 0.0
    for s in ss:
        for n in ns:
            if n * s * 4 > 3.0e9:
                continue
            r = sweep_point(n, s, stencil_dtype=dtype)
            (ams, aname), (bms, bname, bkey) = r["auto"], r["best"]
            ratio = ams / bms
            worst = max(worst, ratio)
            flag = "  <-- cliff" if ratio > 1.05 else ""
            print(f"  {n:7d} {s:4d}  {aname:34s} {ams:8.3f}  {bname + ' ' + str(bkey):34s} {bms:8.3f}  {ratio:5.3f}{flag}", flush=True)
            print("      " + "  ".join(f"{k}:{v[0]:.3f}" for k, v in sorted(r["all"].items())), flush=True)
    print(f"# worst auto / best: {worst:.3f}")


if __name__ == "__main__":
    main()
