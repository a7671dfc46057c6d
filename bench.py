#!/usr/bin/env python3
"""bench.py — UAV-steps/sec of the fused UavSystem::makeStep() kernel on N MI355X (one process per GPU).

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched through torch.distributed.run.
A "step" is one makeStep(dt = 1 ms) of every UAV of the rank's shard (one kernel launch over the whole batch, state
resident in HBM).  Workload at every N: BASELINE.json configs[2] — 100 000 x500 UAVs per GPU, actuator-level references,
no collisions (weak scaling: UAVs are independent, no data-path collective).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DT = 0.001
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s achievable)

# algorithmic bytes per UAV-step: one read + one write of everything the step must touch, FP64
BYTES_PER_UAV_STEP = {
    "actuator": (25 + 3 + 4 + 1) * 8 + 4 + (25 + 3) * 8,            # 492 B  (n_motors = 4)
    "position": (25 + 3 + 1 + 24 + 4) * 8 + 4 + (25 + 3 + 24) * 8,  # 876 B
}


def pmc_traffic(args, n):
    """HBM-side bytes per launch from the committed rocprofv3 PMC summary of this same command (profiles/), corrected as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 on gfx950 (confirmed by the calibration rows of that summary for this
    8-B/lane SoA pattern), WRITE_SIZE exact, both x1024.  None when no summary matches the configuration."""
    stem = {"actuator": "step_kernel", "position": "position_cascade"}.get(args.workload)
    size = f"{n // 1000}k" if n < 1_000_000 else f"{n // 1_000_000}M"
    path = os.path.join(ROOT, "profiles", f"r01_{stem}_{size}_{args.arith}_summary.json")
    if stem is None or args.substeps != 1 or not os.path.exists(path):
        return None, None
    pmc = json.load(open(path)).get("pmc", {})
    if "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
        return None, None
    return (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--uavs", type=int, default=100_000, help="UAVs per GPU")
    ap.add_argument("--volume-per-uav", type=float, default=64.0, help="collision workloads: m^3 of air space per UAV")
    ap.add_argument("--workload", choices=["actuator", "position", "position+collisions"], default="actuator")
    ap.add_argument("--arith", choices=["literal", "fast"], default="fast",
                    help="fast: FMA + rsqrt arithmetic (within 1e-6 of the reference, the production flavour); literal: reference op order")
    ap.add_argument("--substeps", type=int, default=1, help="makeStep rounds fused per launch (state kept in registers)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def make_inputs(n, workload, seed, volume_per_uav=64.0):
    from mrs_multirotor_simulator_amd import synthetic  # numpy only: the GPU legs of the bench never touch oracle/ or tests/
    rng = np.random.default_rng(seed)
    if workload == "actuator":
        st = synthetic.random_state(rng, n, 4)
        cmd = rng.uniform(0.35, 0.60, (n, 4))
    else:
        side = (volume_per_uav * n) ** (1.0 / 3.0)  # 64 m^3 per UAV by default (BASELINE config 4)
        st = synthetic.random_state(rng, n, 4, tilted=True)
        st["x"] = rng.uniform(0, 1, (n, 3)) * [side * 2, side * 2, side / 4] + [0, 0, 5]
        cmd = np.concatenate([st["x"] + rng.uniform(-5, 5, (n, 3)), rng.uniform(-3.14, 3.14, (n, 1))], axis=1)
    return st, cmd


def cpu_baseline(args, st, cmd):
    """The oracle (scalar C restatement of the reference, 1 thread like the reference's serial loop) timed on a bounded
    sample of the same workload."""
    import helpers
    from oracle import oracle_swarm as O
    n = min(args.uavs, 20_000)
    o = O.OracleSwarm(n)
    po = helpers.oracle_params("x500", ground_enabled=True)
    o.construct(0, n, po)
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(o, nm)(0, n)
    o.set_state(0, n, st["x"][:n], st["v"][:n], st["R"][:n], st["omega"][:n], st["motor_rpm"][:n])
    o.set_input(0, n, O.ACTUATOR_CMD if args.workload == "actuator" else O.POSITION_CMD, cmd[:n])
    coll = args.workload.endswith("collisions")
    o.step_n(DT, 2)
    steps, t0 = 0, time.perf_counter()
    while True:
        if coll:
            o.step(DT)
            o.handle_collisions(True, False, 100.0)
            steps += 1
        else:
            o.step_n(DT, 5)
            steps += 5
        el = time.perf_counter() - t0
        if el > args.cpu_seconds:
            break
    out = {"value": n * steps / el, "unit": "UAV-steps/s", "cores": 1, "kind": "port",
           "sample": f"{n} UAVs x {steps} steps of the same workload, oracle/uav_oracle.c -O2 -ffp-contract=off, 1 thread "
                     "(the reference's loop is serial, src/multirotor_simulator.cpp:211-213)"}
    if coll and O.ref_lib() is not None:
        # the reference's OWN broadphase on the same positions: nanoflann build + one radius search per UAV
        # (oracle/_ref, compiled from the reference tree) — "kind": "reference" for this part of the tick
        import ctypes as C
        pts = np.ascontiguousarray(o.get_state()["x"])
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 2.0:
            O.ref_lib().ref_nf_build_and_count(pts.ctypes.data_as(C.POINTER(C.c_double)), n, 3.0, 10)
            reps += 1
        out["collision_broadphase_reference"] = {"kind": "reference", "ms_per_tick": (time.perf_counter() - t0) / reps * 1e3,
                                                 "sample": f"nanoflann kd-tree build + {n} radius searches, 1 thread"}
    if not coll:  # generous upper bound for a CPU implementation: pthreads over UAVs on every host core
        cores = os.cpu_count() or 1
        k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < max(3.0, args.cpu_seconds / 3):
            o.step_n(DT, 10, cores)
            k += 10
        out["all_cores"] = {"value": n * k / (time.perf_counter() - t0), "cores": cores}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import mrs_multirotor_simulator_amd as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    torch.cuda.set_device(local)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched through torch.distributed.run
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    n = args.uavs
    st, cmd = make_inputs(n, args.workload, seed=3 + rank, volume_per_uav=args.volume_per_uav)
    sw = M.Swarm(n, device=local, arith=M.ARITH_FAST if args.arith == "fast" else M.ARITH_LITERAL)
    sw.construct(0, n, M.model_params("x500", ground_enabled=True))
    sw.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    sw.set_input(0, n, M.ACTUATOR_CMD if args.workload == "actuator" else M.POSITION_CMD, cmd)
    coll = args.workload.endswith("collisions")

    sharded = None
    native_comm = False
    force_comm = os.environ.get("MRS_NATIVE_RCCL") == "force"  # rehearse the multi-rank code path with a one-rank communicator
    if coll and (world > 1 or force_comm):  # BASELINE configs[4]: index shards + ONE all-gather of 48-B records per tick (RCCL over xGMI)
        if os.environ.get("MRS_NATIVE_RCCL", "1") != "0":
            # the library issues the all-gather itself on the swarm's stream (mrs_swarm_tick_sharded_n): K ticks = one host call
            from mrs_multirotor_simulator_amd.swarm import rccl_unique_id
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid = torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8).cuda()
            if use_dist:
                dist.broadcast(uid, 0)
            sw.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()), n * world)
            native_comm = True
            sharded = sw
        else:  # the same exchange driven from Python through torch.distributed (sharded.py)
            from mrs_multirotor_simulator_amd.sharded import GpuEngine, ShardedSwarm
            dev = torch.device("cuda", local)
            sharded = ShardedSwarm(n * world, GpuEngine(sw, dev), dev)

    def run(k):
        if native_comm:
            sw.tick_sharded_n(DT, k, True, False, 100.0)
        elif sharded is not None:
            sharded.tick_n(DT, k, True, False, 100.0)
        elif coll:
            sw.tick_n(DT, k, True, False, 100.0)
        else:
            sw.step_n(DT, k, args.substeps)

    def sync_local():
        sw.synchronize()
        torch.cuda.synchronize()

    def barrier():
        sync_local()
        if use_dist:
            dist.barrier()
        sync_local()

    run(args.warmup)
    barrier()
    if use_dist:
        barrier()  # the first RCCL barrier of a process sets up its communicator (milliseconds): keep that out of the start skew
    sw.set_profiling(0 if sharded is not None else 1)  # one hipEvent pair around the timed region, on the swarm's stream
    t0 = time.perf_counter()
    run(args.steps)
    # closing bracket: every rank's K steps are complete at its own synchronize — that instant ends ITS interval; the barrier that
    # follows (an RCCL kernel + host round trip of ~1-2 ms, not part of the workload) and the MAX over ranks close the job's interval
    sync_local()
    el = time.perf_counter() - t0
    barrier()
    kern_ms, n_launch = sw.last_step_kernel_ms() if sharded is None else (el / args.steps * 1e3, args.steps)
    sw.set_profiling(0)
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    assert np.all(np.isfinite(sw.get_state(0, 64)["x"]))
    if rank == 0:
        key = "actuator" if args.workload == "actuator" else "position"
        # one launch reads and writes the state once, however many sub-steps it fuses: no roofline credit for fusion
        alg_bytes = BYTES_PER_UAV_STEP[key] * n
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args, n)
        # the launcher's choice (step_device.inc): buffer-addressed columns below 4 GiB of state, three-wave variant beyond 2 waves/SIMD,
        # non-temporal accesses for model-only steps of small swarms
        npad = (n + 63) // 64 * 64
        kernel_name = ("mrs_uav_model_step" if args.workload == "actuator" else "mrs_uav_step") + ("_multi" if args.substeps > 1 else "")
        if 86 * npad * 8 < 2 ** 32:
            fast1 = args.substeps == 1 and args.arith == "fast"
            kernel_name += "_buf" + ("_w3" if (fast1 and npad // 64 > 2048) else "_nt" if (fast1 and npad // 64 <= (1900 if args.workload == "actuator" else 900)) else "")
        kernel_name += "_" + args.arith
        # swarm_host.hip issues a run of steps without collisions as two half-swarm launches per step on two streams
        launches_per_step = 1
        if not coll and os.environ.get("MRS_SPLIT_STREAMS", "1") != "0" and npad // 64 >= 1024 and -(-args.steps // args.substeps) >= 4:
            launches_per_step = 2
        if traffic is not None:
            traffic *= launches_per_step  # the PMC summary is per dispatch; `achieved` and `traffic` are both per step
        out = {
            "metric": "UAV-steps/sec (whole node) at 1000 Hz sim-dt", "value": world * n * args.steps / el, "unit": "UAV-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: {n} x500 UAVs per GPU, {args.workload} references, dt=1 ms, RK4, ground on"
                       if args.workload == "actuator" else f"{n} x500 UAVs per GPU, {args.workload}, dt=1 ms",
                       "uavs_per_gpu": n, "arith": args.arith, "substeps_per_launch": args.substeps,
                       "parallelism": (f"{world} index shards, one all-gather of 48 B/UAV per tick" if (coll and (world > 1 or native_comm))
                                       else f"{world} independent shard(s), no collective")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel_name if not coll
                         else "whole tick: " + kernel_name + " + collision pass (time per tick, bytes of the step only)", "kernel_avg_ms": kern_ms, "launches": n_launch,
                         "algorithmic_bytes_per_uav_step": BYTES_PER_UAV_STEP[key],
                         "concurrent_launches_per_step": launches_per_step,
                         "method": "one hipEvent pair around the timed region on the swarm's stream (the second stream is joined before the closing "
                                   "event): elapsed / steps, inter-launch gaps included; with two concurrent half-swarm launches per step `achieved` "
                                   "is the bytes of both over that time, and each launch lasts about one step period (rocprofv3 per-dispatch average)"},
        }
        if coll:
            out["roofline"]["note"] = "per-kernel times of the tick: profiles/r01_collision_tick_100k_kernel_stats.csv"
            ticks, searches = sw.collision_stats()
            out["config"]["collision_ticks"], out["config"]["neighbour_searches"] = int(ticks), int(searches)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, st, cmd)
        print(json.dumps(out), flush=True)
    if native_comm:
        sw.comm_destroy()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
