#!/usr/bin/env python3
"""bench.py — UAV-steps/sec of the fused UavSystem::makeStep() kernel on N MI355X (one process per GPU).

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 the driver launches it through torch.distributed.run,
and a PLAIN `python bench.py --gpus N` starts those N ranks itself (a torch.distributed.run child, before this process touches a GPU).
A "step" is one makeStep(dt = 1 ms) of every UAV of the rank's shard (one kernel launch over the whole batch, state resident in
HBM).  Headline workload at every N: BASELINE.json configs[2] — 100 000 x500 UAVs per GPU, actuator-level references, no
collisions (weak scaling: UAVs are independent, no data-path collective).  For N > 1 the same JSON line carries a `config5`
sub-record: BASELINE configs[4], 1 000 000 UAVs with mutual collisions sharded over the N ranks, the collision exchange over RCCL.
Rank 0 prints ONE JSON line.

Timing: W warm-up steps, then regions of EXACTLY K steps, each bracketed by barrier + synchronize on both sides, repeated until 50 ms
have been measured.  `ms_per_step` / `value` = the median region by the WALL clock (time.perf_counter() between the two brackets,
MAX over ranks) — SURVEY §8d's definition; one region of the driver's K = 20 lasts 0.2 ms, a fifth of it host start-up and synchronize
latency, and that is included.  The DEVICE time of the same regions (hipEvents on the swarm's streams: before the first launch, after
the last launch of each stream) is in the line too (`device_ms_per_step`, `value_device_time`) and is what `roofline` is computed from.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DT = 0.001
# MRS_BENCH_REHEARSAL=1: the --gpus N path with every rank on cuda:0 and a gloo process group (NOTES["rehearsal"]) — a one-GPU box can
# run the code an 8-GPU node will run, so that the node is not its first execution
REHEARSAL = os.environ.get("MRS_BENCH_REHEARSAL") == "1"
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBS = 6300.0  # what a plain read+write stream reaches according to the same guide (measured here: 6.0-6.2 TB/s)
INFINITY_CACHE_BYTES = 256 * 2 ** 20
INFINITY_CACHE_GBS = 8600.0  # lower bound the same guide measured for reads served by the Infinity Cache (38 MB table, uniformly random rows)
STATE_BYTES_PER_UAV = 86 * 8 + 4  # swarm_layout.h: F_COUNT doubles + the flag word
HBM_STREAMING_UAVS = 4_000_000    # 2.8 GB of state, 1.65 GB moved per step: nothing is read twice out of a cache

# algorithmic bytes per UAV-step: one read + one write of everything the step must touch, FP64; n = n_motors
def bytes_per_uav_step(key, n_motors=4):
    n = n_motors
    if key == "actuator":  # read state 21+n, F_ext 3, cmd n, init_z 1 (+ flags); write state 21+n, imu 3
        return (21 + n + 3 + n + 1) * 8 + 4 + (21 + n + 3) * 8          # 492 B at n = 4
    return (21 + n + 3 + 1 + 24 + 4) * 8 + 4 + (21 + n + 3 + 24) * 8      # position cascade: 876 B at n = 4, 908 B at n = 6


# what the kernels really move per UAV-step: the v_prev, F_ext (while no force was ever applied) and init_z columns are elided
# (DESIGN §4): read x v R w (18) + cmd + rpm n + flags, write x v R w + imu 3 + rpm n — and the 24 PID doubles both ways
def bytes_moved_per_uav_step(key, n_motors=4):
    n = n_motors
    if key == "actuator":
        return (18 + n + n) * 8 + 4 + (18 + 3 + n) * 8                     # 412 B at n = 4 (PMC: 417)
    return (18 + 4 + n + 24) * 8 + 4 + (18 + 3 + n + 24) * 8              # 796 B at n = 4


BYTES_PER_UAV_STEP = {k: bytes_per_uav_step(k) for k in ("actuator", "position")}
BYTES_MOVED_PER_UAV_STEP = {k: bytes_moved_per_uav_step(k) for k in ("actuator", "position")}
# collision pass: a list tick reads the list head + count and writes the force for every UAV, and gathers the
# positions of the listed partners of the p UAVs that have any; a search tick is the SURVEY figure
COLLISION_BYTES = {"list_tick_per_uav": 28, "list_tick_per_uav_with_partner": 150, "search_tick_per_uav": 92, "search_tick_per_candidate": 24}


# What the fields of the line mean and how they are measured.  NOT printed in the line (the driver keeps the last 8 KB of stdout: the
# line must hold every number, so the prose lives here, in DESIGN.md §6 and behind `python bench.py --explain`).
NOTES = {
    "timing": "regions of exactly K steps, each bracketed by barrier + synchronize on both sides, in two passes: by time.perf_counter() "
              "(ms_per_step / value: wall clock, host start-up and synchronize latency of the region included, MAX over ranks, median over "
              "`regions` regions) and with a hipEvent pair around the region's launches (device_ms_per_step / value_device_time: what the "
              "roofline is computed from)",
    "roofline": "hipEvents around each timed region: one before the first launch, one per stream after its last launch (the later of the two "
                "ends the region; joining the streams afterwards is bookkeeping): elapsed / steps, inter-launch gaps included, median over the "
                "regions; with two concurrent half-swarm launches per step `achieved` is the bytes of both over that time.  `achieved` prices "
                "the ALGORITHMIC bytes (SURVEY 8d); `moved_GBps` the bytes the kernel really moves (elided v_prev / F_ext / init_z columns), "
                "and `frac_of_achievable` holds those against the 6.3 TB/s a read+write stream reaches.  `traffic` = rocprofv3 --pmc "
                "FETCH_SIZE x 2 + WRITE_SIZE per step (separate passes, child runs before this process touches the GPU).  In the "
                "`infinity-cache-resident` regime the state (touched_bytes) fits the 256 MiB Infinity Cache, so HBM bandwidth is not the "
                "operative limit there (`bound`); the `hbm_streaming` sub-record is the same kernel on 4 M UAVs, where every byte comes from HBM",
    "roofline_collision": "whole tick = fused step + collision launch, plus the neighbour search amortised over the ticks between two searches, "
                          "over (step bytes + list-tick bytes); search_* = one search on its own (k_pack_insert<1> + k_query2<1, 3>, 16 searches "
                          "back to back between two hipEvents) against SURVEY 8d's (92 + 24 k) B per UAV, k measured on the run's positions; "
                          "per-kernel times: profiles/r05_collision_tick_*",
    "sub_records": "hbm_streaming: the headline kernel on 4 M UAVs; config4: BASELINE configs[3] (100 000 UAVs, position references + "
                   "collisions + ground) as mrs_swarm_tick_n runs it, and (spawned_in_cell_order) the same swarm spawned in the order mrs_cell_order suggests; literal: the bit-faithful flavour on the headline workload; config2: "
                   "BASELINE configs[1] (400 f550 on the tmux grid); io_tick: config 3 with the publisher payload of every UAV downloaded and a "
                   "command block uploaded every tick (SURVEY 8f rank 2), serial and pipelined; sharded_rank_standin: one rank of 8 x 125 000 "
                   "alone on the GPU behind a fixed-latency stand-in collective (NOT a multi-GPU measurement; split_10us_plus_bytes_at_300GBps: the "
                   "stand-in also charges every collective its bytes, (world - 1) blocks at 300 GB/s; halo_searches / halo_repeats: searches "
                   "that exchanged the records inside another rank's box instead of all records / had to be repeated on all of them); "
                   "config5: BASELINE configs[4] through mrs_swarm_tick_sharded_n on the ranks of this run",
    "rehearsal": "MRS_BENCH_REHEARSAL=1: the --gpus N code path (torch.distributed.run child, one process per rank, process group, MAX over "
                 "ranks, config-5 leg, guarded peer-window child run) with every rank on cuda:0 and a gloo process group; the config-5 exchange "
                 "is a host all-gather over gloo or the peer windows over IPC.  Exercises the code, measures nothing: n_devices says 1",
}


def _round_floats(o, digits=7):
    """nested records carry floats at 7 significant digits (the top-level value / ms_per_step stay exact)"""
    if isinstance(o, float):
        return float(f"{o:.{digits}g}") if np.isfinite(o) else o
    if isinstance(o, dict):
        return {k: _round_floats(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round_floats(v, digits) for v in o]
    return o


def compact_line(out):
    """the ONE line: top-level scalars as they are, every nested record with rounded floats"""
    return {k: (_round_floats(v) if isinstance(v, (dict, list)) else v) for k, v in out.items()}


def pmc_traffic(args, n, workload=None):
    """HBM-side bytes per launch from a committed rocprofv3 PMC summary of this same command (profiles/), corrected as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 on gfx950 (confirmed by the calibration rows of that summary for this
    8-B/lane SoA pattern), WRITE_SIZE exact, both x1024.  None when no summary matches the configuration."""
    stem = {"actuator": "step_kernel", "position": "position_cascade", "position+collisions": "collision_tick"}.get(workload or args.workload)
    size = f"{n // 1000}k" if n < 1_000_000 else f"{n // 1_000_000}M"
    if stem is None or args.substeps != 1:
        return None, None
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{stem}_{size}_{args.arith}_summary.json")
        if not os.path.exists(path):
            continue
        pmc = json.load(open(path)).get("pmc", {})
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            return (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


def live_traffic(args, uavs=None, workload=None):
    """HBM-side bytes per dispatch of the step kernel, MEASURED in this run: two short child runs of the same workload under
    `rocprofv3 --pmc` — FETCH_SIZE and WRITE_SIZE in separate passes, kernel-trace / stats off, as MI355X_MICROARCH.md prescribes —
    corrected as the guide says (FETCH_SIZE x2 on gfx950, both x1024).  The children are ordinary child processes (`-- python
    bench.py --pmc-child ...`, the program itself after `--`); any failure returns None and the committed profile is used instead."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    vals = {}
    uavs = args.uavs if uavs is None else uavs
    workload = args.workload if workload is None else workload
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mrs_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
               "--uavs", str(uavs), "--workload", workload, "--arith", args.arith, "--substeps", str(args.substeps),
               "--volume-per-uav", str(args.volume_per_uav), "--steps", "64", "--warmup", "16"]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-200:]}"
            # (the dominant kernel only: a collision workload also runs search kernels and, on its first ticks, plain step kernels)
            rows = [row for row in csv.DictReader(open(files[0])) if row["Counter_Name"] == ctr and row["Kernel_Name"].startswith("mrs_uav_")]
            names = {}
            for row in rows:
                names[row["Kernel_Name"]] = names.get(row["Kernel_Name"], 0) + 1
            top = max(names, key=names.get) if names else None
            acc = [float(row["Counter_Value"]) for row in rows if row["Kernel_Name"] == top]
            if not acc:
                return None, f"no step-kernel rows for {ctr}"
            vals[ctr] = sum(acc) / len(acc)
        except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
            return None, f"rocprofv3 --pmc {ctr}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "live rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE in separate passes)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--uavs", type=int, default=100_000, help="UAVs per GPU")
    ap.add_argument("--volume-per-uav", type=float, default=64.0, help="collision workloads: m^3 of air space per UAV")
    ap.add_argument("--workload", choices=["actuator", "position", "position+collisions"], default="actuator")
    ap.add_argument("--order", choices=["random", "xcell", "morton"], default="random",
                    help="collision workloads: UAV indices unrelated to positions (the generator's order), or sorted at spawn by x-major "
                         "list cells / by a Morton key of the list cells (the caller's indices then follow space)")
    ap.add_argument("--arith", choices=["literal", "fast"], default="fast",
                    help="fast: FMA + rsqrt arithmetic (within 1e-6 of the reference, the production flavour); literal: reference op order")
    ap.add_argument("--substeps", type=int, default=1, help="makeStep rounds fused per launch (state kept in registers)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-measure-ms", type=float, default=50.0, help="regions of --steps steps are repeated until this much was timed")
    ap.add_argument("--config5", choices=["auto", "on", "off"], default="auto",
                    help="the 1 000 000-UAV collision leg (BASELINE configs[4]); auto = at every N for the default workload (one rank: the same "
                         "path on a one-rank communicator — the figure the N-rank value is divided by)")
    ap.add_argument("--sub-records", choices=["auto", "on", "off"], default="auto",
                    help="hbm_streaming (4 M UAVs) and config4 (100 k UAVs with collisions) sub-records; auto = in the default N=1 run")
    ap.add_argument("--traffic", choices=["live", "profile", "off"], default="live",
                    help="roofline.traffic: live = two rocprofv3 --pmc child runs of the same workload (N=1 only), profile = the committed summary")
    ap.add_argument("--explain", action="store_true", help="print what the fields of the line mean and how they are measured, then exit")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the short run rocprofv3 wraps for --traffic live
    ap.add_argument("--config5-timeout", type=float, default=240.0, help="seconds after which the config-5 leg is given up (the headline line is printed regardless)")
    ap.add_argument("--config5-uavs", type=int, default=1_000_000, help="UAVs of the config-5 leg, all ranks together")
    ap.add_argument("--config5-shards", choices=["slabs", "index"], default="slabs", help="x-sorted slabs (boundary sets stay small) or index ranges")
    ap.add_argument("--config5-transport", choices=["rccl", "peer"], default="rccl",
                    help="collective backend of the config-5 leg: RCCL all-gather, or the library's peer-window exchange (direct writes into the "
                         "peers' device memory over xGMI, IPC handles carried by torch.distributed; never run across devices yet: opt-in)")
    ap.add_argument("--config5-peer", choices=["auto", "on", "off"], default="auto",
                    help="a second, GUARDED config-5 record with the peer-window exchange (`config5_peer`), run in child processes after the RCCL leg so "
                         "that whatever happens to it cannot touch the line's other records or the exit status; auto = whenever N > 1")
    ap.add_argument("--only-config5", action="store_true", help=argparse.SUPPRESS)  # the child run behind config5_peer: that leg alone, one JSON object
    ap.add_argument("--config5-exchange", choices=["export", "full"], default="export",
                    help="export: boundary UAVs only between two searches; full: all 48-B records on every tick")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` started plainly: run the N ranks as a torch.distributed.run CHILD (this process has not touched
    the GPU: device_count() does not initialise it on this image) and leave with its exit code."""
    import torch
    have = torch.cuda.device_count()
    if have < (1 if REHEARSAL else args.gpus):
        raise SystemExit(f"bench.py --gpus {args.gpus}: this machine shows {have} GPU(s)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def spatial_order(x, order, cell=2.25):
    """permutation that sorts UAVs by the cell of their position: x-major (cx, cy, cz) or a Morton key of the three cell indices"""
    c = np.floor((x - x.min(axis=0)) / cell).astype(np.int64)
    if order == "xcell":
        return np.lexsort((c[:, 2], c[:, 1], c[:, 0]))
    from mrs_multirotor_simulator_amd import cell_order  # the library's own helper for callers that choose their spawn order (mrs_cell_order)
    return cell_order(x, cell)


def make_inputs(n, workload, seed, volume_per_uav=64.0, n_motors=4, order="random"):
    from mrs_multirotor_simulator_amd import synthetic  # numpy only: the GPU legs of the bench never touch oracle/ or tests/
    rng = np.random.default_rng(seed)
    if workload == "config2":
        # BASELINE configs[1]: the tmux/standalone_400_uavs spawn grid — 20 x 20, 4 m pitch, z = 0, heading 0 — and the
        # goals of its goto.py: x, y ~ U(-40, 40), z ~ U(2, 20), heading ~ U(-3.14, 3.14), default_rng(400)
        rng = np.random.default_rng(400)
        side = int(round(n ** 0.5))
        assert side * side == n, "config 2 is a square grid"
        gx, gy = np.meshgrid(np.arange(side) * 4.0, np.arange(side) * 4.0, indexing="ij")
        st = {"x": np.stack([gx.ravel(), gy.ravel(), np.zeros(n)], axis=1), "heading": np.zeros(n)}
        cmd = np.concatenate([rng.uniform(-40, 40, (n, 2)), rng.uniform(2, 20, (n, 1)), rng.uniform(-3.14, 3.14, (n, 1))], axis=1)
        return st, cmd
    if workload == "actuator":
        st = synthetic.random_state(rng, n, n_motors)
        cmd = rng.uniform(0.35, 0.60, (n, n_motors))
    else:
        side = (volume_per_uav * n) ** (1.0 / 3.0)  # 64 m^3 per UAV by default (BASELINE config 4)
        st = synthetic.random_state(rng, n, n_motors, tilted=True)
        st["x"] = rng.uniform(0, 1, (n, 3)) * [side * 2, side * 2, side / 4] + [0, 0, 5]
        cmd = np.concatenate([st["x"] + rng.uniform(-5, 5, (n, 3)), rng.uniform(-3.14, 3.14, (n, 1))], axis=1)
        if order != "random":
            perm = spatial_order(st["x"], order)
            st = {k: v[perm] for k, v in st.items()}
            cmd = cmd[perm]
    return st, cmd


def cpu_baseline(args, st, cmd, workload=None, uavs=None, airframe="x500", seconds=None):
    """The oracle (scalar C restatement of the reference, 1 thread like the reference's serial loop) timed on a bounded
    sample of the same workload, built for this host with the flags SURVEY §8d names."""
    from oracle import oracle_swarm as O
    if O._lib is None:
        flags = O.use_native()  # -O3 -march=native -ffp-contract=off, compiled on this machine (falls back to the portable -O2 build)
        cpu_baseline.flags = flags
    flags = getattr(cpu_baseline, "flags", O.PORTABLE_FLAGS)
    import helpers
    workload = args.workload if workload is None else workload
    seconds = args.cpu_seconds if seconds is None else seconds
    n = min(args.uavs if uavs is None else uavs, 20_000)
    o = O.OracleSwarm(n)
    po = helpers.oracle_params(airframe, ground_enabled=True)
    if workload == "config2":
        o.construct(0, n, po, st["x"][:n], st["heading"][:n])
    else:
        o.construct(0, n, po)
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(o, nm)(0, n)
    if workload != "config2":
        o.set_state(0, n, st["x"][:n], st["v"][:n], st["R"][:n], st["omega"][:n], st["motor_rpm"][:n])
    o.set_input(0, n, O.ACTUATOR_CMD if workload == "actuator" else O.POSITION_CMD, cmd[:n])
    coll = workload.endswith("collisions")
    o.step_n(DT, 2)
    chunk = 5 if n >= 2000 else 500
    steps, t0 = 0, time.perf_counter()
    while True:
        if coll:
            o.step(DT)
            o.handle_collisions(True, False, 100.0)
            steps += 1
        else:
            o.step_n(DT, chunk)
            steps += chunk
        el = time.perf_counter() - t0
        if el > seconds:
            break
    out = {"value": n * steps / el, "unit": "UAV-steps/s", "cores": 1, "kind": "port", "compiler_flags": "gcc " + flags,
           "sample": f"{n} {airframe} UAVs x {steps} steps of the same workload, oracle/uav_oracle.c, 1 thread"}
    if coll and O.ref_lib() is not None:
        # the reference's OWN broadphase on the same positions: nanoflann build + one radius search per UAV
        # (oracle/_ref, compiled from the reference tree) — "kind": "reference" for this part of the tick
        import ctypes as C
        pts = np.ascontiguousarray(o.get_state()["x"])
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 2.0:
            O.ref_lib().ref_nf_build_and_count(pts.ctypes.data_as(C.POINTER(C.c_double)), n, 3.0, 10)
            reps += 1
        ms = (time.perf_counter() - t0) / reps * 1e3
        out["collision_broadphase_reference"] = {"kind": "reference", "ms_per_tick": ms, "uavs": n, "us_per_uav_tick": ms * 1e3 / n,
                                                 "sample": f"the reference's own nanoflann (oracle/_ref): build + {n} radius searches"}
    if not coll and workload != "config2":  # generous upper bound for a CPU implementation: pthreads over UAVs on every host core
        cores = os.cpu_count() or 1
        k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < max(3.0, seconds / 3):
            o.step_n(DT, 10, cores)
            k += 10
        out["all_cores"] = {"value": n * k / (time.perf_counter() - t0), "cores": cores}
    return out


class Ranks:
    """what a leg needs of the process group: rank/world, barrier, MAX over ranks"""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
        self.rehearsal = REHEARSAL and self.world > 1
        if self.rehearsal:
            self.local = 0  # every rank on the one device
        if self.local >= torch.cuda.device_count():
            raise SystemExit(f"rank {self.rank}: local rank {self.local} has no GPU ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(self.local)
        self.use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched through torch.distributed.run
        if self.use_dist:
            if self.rehearsal:  # (RCCL refuses two ranks on one device)
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local))

    def barrier(self, sync_local):
        sync_local()
        if self.use_dist:
            self.dist.barrier()
        sync_local()

    def max_over_ranks(self, x):
        if not self.use_dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.rehearsal else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.use_dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed_regions(R, run, sync_local, steps, warmup, min_ms, after_region=None):
    """W warm-up steps, then regions of exactly `steps` steps (barrier + synchronize on both sides, MAX over ranks) until min_ms have
    been timed.  Returns the per-region seconds."""
    run(warmup)
    R.barrier(sync_local)
    if R.use_dist:
        R.barrier(sync_local)  # the first RCCL barrier of a process sets up its communicator (milliseconds): keep that out of the start skew
    times, extra = [], []
    while True:
        R.barrier(sync_local)
        t0 = time.perf_counter()
        run(steps)
        # closing bracket: every rank's K steps are complete at its own synchronize — that instant ends ITS interval; the barrier
        # of the next region (an RCCL kernel + host round trip of ~1-2 ms, not part of the workload) follows, MAX over ranks closes it
        sync_local()
        el = time.perf_counter() - t0
        if after_region is not None:
            extra.append(after_region())
        times.append(R.max_over_ranks(el))  # identical on every rank: all ranks take the same number of regions
        if sum(times) * 1e3 >= min_ms or len(times) >= 400:
            break
    return times, extra


def kernel_name_of(n, workload, arith, substeps):
    """the launcher's choice (step_device.inc): buffer-addressed columns below 4 GiB of state, three-wave variant beyond 2 waves/SIMD,
    non-temporal accesses for small swarms and for swarms far beyond the Infinity Cache"""
    key = "actuator" if workload == "actuator" else "position"
    npad = (n + 63) // 64 * 64
    if workload.endswith("collisions"):
        return "mrs_uav_step_coll" + ("_buf" if 86 * npad * 8 < 2 ** 32 else "") + "_" + arith
    name = ("mrs_uav_model_step" if workload == "actuator" else "mrs_uav_step") + ("_multi" if substeps > 1 else "")
    if 86 * npad * 8 < 2 ** 32:
        fast1 = substeps == 1 and arith == "fast"
        hbm_stream = fast1 and npad * BYTES_MOVED_PER_UAV_STEP[key] >= 1.4e9  # far beyond the Infinity Cache: two-wave non-temporal kernel
        name += "_buf" + ("_nt" if (hbm_stream or (fast1 and npad // 64 <= (1900 if workload == "actuator" else 900)))
                          else "_w3" if (fast1 and npad // 64 > 2048) else "")
    return name + "_" + arith


def mean_search_candidates(x, inv_cell):
    """k-bar of SURVEY §8d's collision formula, MEASURED on the positions of the run: per UAV, the other UAVs in the 27 cells around
    its own — the records a search fetches (the tag filter of collide.hip k_query admits exactly the members of the probed cells)."""
    c = np.floor(np.asarray(x) * inv_cell).astype(np.int64) + (1 << 20)
    key = (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
    uniq, cnt = np.unique(key, return_counts=True)
    total = np.zeros(len(key))
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                k2 = key + ((dx << 42) + (dy << 21) + dz)
                at = np.searchsorted(uniq, k2)
                at[at >= len(uniq)] = len(uniq) - 1
                total += np.where(uniq[at] == k2, cnt[at], 0)
    return float(total.mean() - 1.0)  # (minus the UAV itself)


def step_leg(args, R, n, workload, steps, warmup, traffic=(None, "not requested"), min_ms=None, seed=3, airframe="x500", arith=None, substeps=None):
    """One timed workload on this rank's GPU: `n` UAVs of one airframe, regions of exactly `steps` steps (ticks).  Returns (record, st,
    cmd); the record is built on rank 0 only."""
    import mrs_multirotor_simulator_amd as M
    from mrs_multirotor_simulator_amd import airframes
    torch = R.torch
    arith = args.arith if arith is None else arith
    substeps = args.substeps if substeps is None else substeps
    n_motors = airframes.AIRFRAMES[airframe]["n_motors"]
    st, cmd = make_inputs(n, workload, seed=seed + R.rank, volume_per_uav=args.volume_per_uav, n_motors=n_motors, order=args.order)
    sw = M.Swarm(n, device=R.local, arith=M.ARITH_FAST if arith == "fast" else M.ARITH_LITERAL)
    if workload == "config2":
        sw.construct(0, n, M.model_params(airframe, ground_enabled=True), st["x"], st["heading"])
    else:
        sw.construct(0, n, M.model_params(airframe, ground_enabled=True))
        sw.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    sw.set_input(0, n, M.ACTUATOR_CMD if workload == "actuator" else M.POSITION_CMD, cmd)
    coll = workload.endswith("collisions")

    def run(k):
        if coll:
            sw.tick_n(DT, k, True, False, 100.0)
        else:
            sw.step_n(DT, k, substeps)

    def sync_local():
        sw.synchronize()
        torch.cuda.synchronize()

    if args.pmc_child:  # under rocprofv3 --pmc: just run the launches
        run(warmup)
        run(steps)
        sync_local()
        return None, st, cmd
    # two passes over regions of exactly `steps` steps: with the library's hipEvent pair around each region (device time: roofline), and
    # without it (wall clock: `value` — the events, and the synchronisation that reads them, are measurement, not workload)
    budget = args.min_measure_ms if min_ms is None else min_ms
    sw.set_profiling(1)  # one hipEvent pair around every step_n / tick_n call, on the swarm's stream
    _, ev = timed_regions(R, run, sync_local, steps, warmup, budget / 2, after_region=sw.last_step_kernel_ms)
    sw.set_profiling(0)
    times, _ = timed_regions(R, run, sync_local, steps, 0, budget / 2)
    x_end = sw.get_state(0, n)["x"] if coll else sw.get_state(0, 64)["x"]
    assert np.all(np.isfinite(x_end[:64]))
    kern_ms = float(np.median([e[0] for e in ev]))
    n_launch = ev[0][1]
    # `value` is the contract's figure: UAVs x steps / WALL seconds of regions of exactly K steps, each bracketed by
    # barrier + synchronize on both sides, MAX over ranks, median over the regions — host start-up and synchronize latency of the
    # region included (at the driver's K = 20 a region lasts 0.2 ms, about a fifth of it that latency).  The device time of the
    # same regions (hipEvents on the swarm's streams) is reported next to it and is what the roofline is computed from.
    wall = float(np.median(times))
    el = R.max_over_ranks(kern_ms * n_launch * 1e-3)
    out = None
    if R.rank == 0:
        world = R.world
        key = "actuator" if workload == "actuator" else "position"
        # one launch reads and writes the state once, however many sub-steps it fuses: no roofline credit for fusion
        b_alg, b_mov = bytes_per_uav_step(key, n_motors), bytes_moved_per_uav_step(key, n_motors)
        alg_bytes = b_alg * n
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        moved = b_mov * n / (kern_ms * 1e-3) / 1e9
        tr, tr_src = traffic  # measured by main() before this process touched the GPU; (None, why) otherwise
        if tr is None and args.traffic != "off" and airframe == "x500" and arith == args.arith:
            if args.traffic == "live" and world == 1 and traffic[1] != "not requested":
                sys.stderr.write(f"bench.py: live PMC traffic unavailable for {n} UAVs / {workload} ({tr_src}); using the committed profile\n")
            tr, tr_src = pmc_traffic(args, n, workload)
        npad = (n + 63) // 64 * 64
        kernel_name = kernel_name_of(n, workload, arith, substeps)
        # tick_single.hip issues a run of steps without collisions as two half-swarm launches per step on two streams
        launches_per_step = 1
        if not coll and os.environ.get("MRS_SPLIT_STREAMS", "1") != "0" and npad // 64 >= 1024 and -(-steps // substeps) >= 4:
            launches_per_step = 2
        if tr is not None:
            tr *= launches_per_step  # the PMC figure is per dispatch; `achieved` and `traffic` are both per step
        touched = STATE_BYTES_PER_UAV * n
        resident = touched < INFINITY_CACHE_BYTES
        if workload == "actuator":
            wl = f"BASELINE configs[2]: {n} {airframe} UAVs per GPU, {workload} references, dt=1 ms, RK4, ground on"
        elif workload == "config2":
            wl = f"BASELINE configs[1]: {n} {airframe} on the tmux/standalone_400_uavs grid, position references (goto.py goals), dt=1 ms"
        elif coll:
            wl = f"BASELINE configs[3]: {n} {airframe} UAVs, position references + collisions + ground, {args.volume_per_uav:g} m^3 per UAV, dt=1 ms"
        else:
            wl = f"{n} {airframe} UAVs per GPU, {workload}, dt=1 ms"
        out = {
            "metric": "UAV-steps/sec (whole node) at 1000 Hz sim-dt", "value": world * n * steps / wall, "unit": "UAV-steps/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": wall / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "regions": len(times), "device_ms_per_step": el / steps * 1e3, "value_device_time": world * n * steps / el,
            "first_region_wall_ms_per_step": times[0] / steps * 1e3,
            "config": {"workload": wl, "uavs_per_gpu": n, "airframe": airframe, "n_motors": n_motors, "arith": arith,
                       "substeps_per_launch": substeps, "order": args.order if coll else "n/a", "parallelism": f"{world} independent shard(s), no collective on the data path"},
            # `peak` is the HBM3E spec figure in every regime (comparable across sizes); while the touched state fits the 256 MiB Infinity
            # Cache the operative limit is that cache, not HBM: `bound` says so, and the guide's measured lower bound for it is given
            "roofline": {"bound": "infinity-cache" if resident else "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_src,
                         "infinity_cache_peak": INFINITY_CACHE_GBS, "touched_bytes": touched,
                         "bytes_moved_per_uav_step": b_mov, "moved_GBps": moved,
                         "achievable_peak": HBM_ACHIEVABLE_GBS, "frac_of_achievable": moved / HBM_ACHIEVABLE_GBS,
                         "kernel": kernel_name if not coll
                         else kernel_name + " + search every ~27 ticks (time per tick, bytes of the step)",
                         "kernel_avg_ms": kern_ms, "launches": n_launch,
                         "algorithmic_bytes_per_uav_step": b_alg,
                         "concurrent_launches_per_step": launches_per_step},
        }
        if coll:
            ticks, searches = sw.collision_stats()
            out["config"]["collision_ticks"], out["config"]["neighbour_searches"] = int(ticks), int(searches)
            fused, stalls, replayed, ahead = sw.fused_stats()
            out["config"]["ticks_evaluated_by_the_next_step_launch"], out["config"]["stale_list_stalls"] = int(fused), int(stalls)
            out["config"]["launches_replayed"], out["config"]["searches_queued_ahead"] = int(replayed), int(ahead)
            p = 0.06  # fraction of UAVs with a listed partner at 64 m^3 per UAV
            cb = COLLISION_BYTES["list_tick_per_uav"] + COLLISION_BYTES["list_tick_per_uav_with_partner"] * p
            # the search itself (92 + 24 k-bar bytes per UAV): k-bar measured on the positions at the end of the run (list
            # cells of collide.hip: edge sqrt(3) + skin + 0.018 m), the two search kernels timed live, back to back, by hipEvents
            kbar = mean_search_candidates(x_end, 1.0 / (3.0 ** 0.5 + 0.5 + 0.0179491924))
            search_ms = sw.debug_search_ms(reps=16)
            search_bytes = (COLLISION_BYTES["search_tick_per_uav"] + COLLISION_BYTES["search_tick_per_candidate"] * kbar) * n
            out["roofline_collision"] = {
                "bound": "infinity-cache" if resident else "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "algorithmic_bytes_per_uav_list_tick": cb,
                "algorithmic_bytes_per_uav_search_tick": COLLISION_BYTES["search_tick_per_uav"] + COLLISION_BYTES["search_tick_per_candidate"] * kbar,
                "search_candidates_per_uav": kbar, "search_bytes": search_bytes, "search_ms": search_ms,
                "search_achieved": search_bytes / (search_ms * 1e-3) / 1e9, "search_frac": search_bytes / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "ticks_per_search": (int(ticks) / max(1, int(searches))),
                "achieved_whole_tick": (b_alg + cb) * n / (kern_ms * 1e-3) / 1e9,
                "frac_whole_tick": (b_alg + cb) * n / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del sw
    return out, st, cmd


SUB_KEYS = ("value", "unit", "steps", "warmup", "ms_per_step", "regions", "device_ms_per_step", "value_device_time", "config", "roofline", "roofline_collision", "cpu_baseline")
SUB_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "moved_GBps", "frac_of_achievable", "kernel", "kernel_avg_ms", "launches",
                     "concurrent_launches_per_step", "algorithmic_bytes_per_uav_step")
SUB_CONFIG_DROP = ("parallelism", "n_motors", "substeps_per_launch", "ticks_evaluated_by_the_next_step_launch")


def sub_record(rec):
    """the fields of a full record that a sub-record of the headline line keeps (numbers; the prose is NOTES)"""
    out = {k: rec[k] for k in SUB_KEYS if k in rec}
    if "roofline" in out:
        out["roofline"] = {k: out["roofline"][k] for k in SUB_ROOFLINE_KEYS if k in out["roofline"]}
    if "config" in out:
        out["config"] = {k: v for k, v in out["config"].items() if k not in SUB_CONFIG_DROP}
    if "cpu_baseline" in out:
        out["cpu_baseline"] = dict(out["cpu_baseline"])
        for k in ("compiler_flags", "unit"):
            out["cpu_baseline"].pop(k, None)
    return out


def mini_record(rec):
    """a sub-record cut down to what the line needs of the secondary workloads"""
    out = sub_record(rec)
    out.pop("unit", None), out.pop("regions", None), out.pop("value_device_time", None)
    out["config"] = {k: out["config"][k] for k in ("workload", "uavs_per_gpu", "arith") if k in out["config"]}
    out["roofline"] = {k: out["roofline"][k] for k in ("achieved", "frac", "kernel", "kernel_avg_ms") if k in out["roofline"]}
    return out


def sharded_rank_cost(n_per_rank=125_000, world=8, ticks=600, latency_us=20.0, split=True, warm=80, volume_per_uav=64.0):
    """Device time ONE rank of a `world`-rank config-5 run spends per tick: rank world/2 alone on this GPU with the library's measurement
    stand-in for the collective (mrs_swarm_comm_init_standin): every collective takes `latency_us` of stream time and the rank's
    neighbours in the slab order are periodic images of itself — boundary sets, boundary / interior launches, searches and buffer
    sizes of the real run; missing: the other ranks' physics and the wire.  split=False: round 2's serial protocol (MRS_SHARD_SPLIT=0)."""
    import mrs_multirotor_simulator_amd as M
    from mrs_multirotor_simulator_amd.sharded import shard_range
    rank, n_total = world // 2, n_per_rank * world
    st, cmd = make_inputs(n_total, "position+collisions", seed=5, volume_per_uav=volume_per_uav)
    order = M.slab_partition(st["x"], world)
    lo, hi = shard_range(n_total, world, rank)
    idx = order[lo:hi]
    width = float(st["x"][idx, 0].max() - st["x"][idx, 0].min()) * (1.0 + 1.0 / len(idx))
    before = os.environ.get("MRS_SHARD_SPLIT")
    if not split:
        os.environ["MRS_SHARD_SPLIT"] = "0"  # (read when the swarm is created)
    try:
        g = M.Swarm(hi - lo, arith=M.ARITH_FAST)
    finally:
        if not split:
            if before is None:
                os.environ.pop("MRS_SHARD_SPLIT", None)
            else:
                os.environ["MRS_SHARD_SPLIT"] = before
    g.construct(0, hi - lo, M.model_params("x500", ground_enabled=True))
    g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
    g.set_input(0, hi - lo, M.POSITION_CMD, cmd[idx])
    del st, cmd
    g.comm_init_standin(world, rank, n_total, latency_us, width)
    g.tick_sharded_n(DT, warm, True, False, 100.0)
    g.synchronize()
    s0, _ = g.split_stats()
    c0 = g.comm_info()
    t0 = time.perf_counter()
    g.tick_sharded_n(DT, ticks, True, False, 100.0)
    g.synchronize()
    el = time.perf_counter() - t0
    ci = g.comm_info()
    s1, nbnd = g.split_stats()
    _, on_halo, repeats, halo_cap = g.search_stats()
    out = {"us_per_tick": el / ticks * 1e6, "ticks": ticks, "rank": rank, "world": world, "uavs_per_rank": hi - lo, "collective_latency_us": latency_us,
           "form": "split" if split else "serial", "split_ticks": int(s1 - s0), "boundary_blocks": int(nbnd), "blocks": (hi - lo + 63) // 64,
           "export_set": int(ci["export_count"]), "export_capacity": int(ci["export_capacity"]), "searches": int(ci["searches"] - c0["searches"]),
           "replayed_noop_ticks": int(ci["noop_ticks"] - c0["noop_ticks"]),
           # (since the communicator was bound) searches on a halo exchange / repeated on all records; bytes a rank sends per tick and per search tick
           "halo_searches": int(on_halo), "halo_repeats": int(repeats), "bytes_per_tick": int(ci["bytes_per_tick"]), "bytes_per_search_tick": int(ci["bytes_per_rebuild"])}
    g.comm_destroy()
    del g
    return out


def sharded_rank_record(args):
    """sub-record of the default line: what one rank of an 8-rank config-5 run costs per tick at 10 and 20 us of collective latency
    (split form, and the serial form at 10 us) — the figures DESIGN §5 and BASELINE.md quote, reproducible from the driver's own run"""
    runs = [sharded_rank_cost(latency_us=10.0, ticks=400), sharded_rank_cost(latency_us=20.0, ticks=400), sharded_rank_cost(latency_us=10.0, ticks=400, split=False)]
    # the same with the collectives' BYTES charged as well (a fixed latency makes a search tick's blocks as cheap as an ordinary tick's
    # 54 KB): (world - 1) blocks received at 300 GB/s, a ring all-gather's bus bandwidth over xGMI.  Round 5's halo exchange of a search
    # is what this figure shows: 41.6 us with every record gathered (6 MB per rank), 39.1 with halos (~1 MB) — medians of five runs
    os.environ["MRS_STANDIN_GBPS"] = "300"
    try:
        wire = sharded_rank_cost(latency_us=10.0, ticks=400)
    finally:
        os.environ.pop("MRS_STANDIN_GBPS", None)
    keep = ("split_ticks", "boundary_blocks", "blocks", "export_set", "searches", "replayed_noop_ticks", "halo_searches", "halo_repeats", "bytes_per_tick", "bytes_per_search_tick")
    return {"workload": "rank 4 of 8 x 125000 UAVs of BASELINE configs[4] alone on the GPU, fixed-latency stand-in collective: NOT a multi-GPU measurement",
            "unit": "us per tick (wall clock, 400 ticks incl. searches)",
            "split_10us": runs[0]["us_per_tick"], "split_20us": runs[1]["us_per_tick"], "serial_10us": runs[2]["us_per_tick"],
            "split_10us_plus_bytes_at_300GBps": wire["us_per_tick"],
            "split_run": {k: runs[0][k] for k in keep}, "serial_run": {k: runs[2][k] for k in keep}}


def io_tick_record(args, n=100_000, ticks=150, warm=20):
    """The headline workload with the host in the loop EVERY tick (the reference publishes every UAV's odometry / IMU /
    range after every step, src/uav_system_ros.cpp:278-282, and its subscribers write commands at any time): a staged command block
    up (n x 4 doubles), one makeStep, the packed publisher payload of every UAV down (n x 136 B).  serial: commit, step, synchronous
    download, one after the other; pipelined: the download is started behind the step (mrs_swarm_get_outputs_async) and waited for
    after the NEXT tick has been queued, copies on a stream of their own.  The rows are pre-filled (what the subscribers would write);
    the payload is handed out, not read."""
    import mrs_multirotor_simulator_amd as M
    st, cmd = make_inputs(n, "actuator", seed=3)
    sw = M.Swarm(n, arith=M.ARITH_FAST if args.arith == "fast" else M.ARITH_LITERAL)
    sw.construct(0, n, M.model_params("x500", ground_enabled=True))
    sw.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    for _ in range(2):  # both row blocks
        sw.input_staging(n, 4)[:] = cmd
        sw.commit_input(0, n, M.ACTUATOR_CMD, 4)
    from mrs_multirotor_simulator_amd.swarm import OUTPUT_DTYPE
    out_bytes, in_bytes = n * OUTPUT_DTYPE.itemsize, n * 4 * 8

    def serial(k):
        for _ in range(k):
            sw.input_staging(n, 4)
            sw.commit_input(0, n, M.ACTUATOR_CMD, 4)
            sw.step(DT)
            sw.get_outputs_view()

    def pipelined(k):
        pending = None
        for _ in range(k):
            sw.input_staging(n, 4)
            sw.commit_input(0, n, M.ACTUATOR_CMD, 4)
            sw.step(DT)
            t = sw.get_outputs_async()
            if pending is not None:
                sw.outputs_wait(pending)
            pending = t
        sw.outputs_wait(pending)

    res = {}
    for name, fn in (("serial", serial), ("pipelined", pipelined)):
        fn(warm)
        sw.synchronize()
        t0 = time.perf_counter()
        fn(ticks)
        sw.synchronize()
        el = (time.perf_counter() - t0) / ticks
        res[name] = {"ms_per_tick": el * 1e3, "value": n / el, "d2h_GBps": out_bytes / el / 1e9, "pcie_GBps_both_ways": (out_bytes + in_bytes) / el / 1e9}
    del sw
    return {"workload": f"BASELINE configs[2] ({n} x500) + per tick: command block up, every UAV's publisher payload down",
            "unit": "UAV-steps/s", "ticks": ticks, "bytes_down_per_tick": out_bytes, "bytes_up_per_tick": in_bytes,
            "serial": res["serial"], "pipelined": res["pipelined"], "speedup": res["serial"]["ms_per_tick"] / res["pipelined"]["ms_per_tick"]}


def _gloo_allgather(R):
    """mrs_allgather_fn over the run's gloo group (rehearsal only): the launches queued so far finish, the bytes cross the processes"""
    import ctypes as C
    torch, dist = R.torch, R.dist
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def allgather(user, send, recv, nbytes, stream):
        try:
            if hip.hipStreamSynchronize(stream):
                return 1
            mine = np.empty(nbytes, dtype=np.uint8)
            if hip.hipMemcpy(mine.ctypes.data, send, nbytes, 2):  # device -> host
                return 1
            everyone = torch.empty(R.world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(everyone, torch.from_numpy(mine))
            return hip.hipMemcpy(recv, everyone.numpy().ctypes.data, R.world * nbytes, 1)  # host -> device
        except Exception as e:  # noqa: BLE001 - a lost peer: the library turns the status into MRS_ERR_*, the leg into config5.error
            sys.stderr.write(f"bench.py: rehearsal all-gather failed: {type(e).__name__}: {e}\n")
            return 1

    return allgather


def config5_leg(args, R):
    """BASELINE configs[4]: `--config5-uavs` UAVs (1 000 000) with mutual collisions, sharded over the ranks; the collision exchange is
    issued by the library itself on the swarm's stream (mrs_swarm_tick_sharded_n: RCCL bound at run time)."""
    import mrs_multirotor_simulator_amd as M
    from mrs_multirotor_simulator_amd.sharded import shard_range
    from mrs_multirotor_simulator_amd.swarm import rccl_unique_id
    torch, dist = R.torch, R.dist
    n_total = args.config5_uavs
    lo, hi = shard_range(n_total, R.world, R.rank)
    n = hi - lo
    st, cmd = make_inputs(n_total, "position+collisions", seed=5, volume_per_uav=args.volume_per_uav)  # every rank draws the same swarm
    # spatially coherent shards: x-sorted, equal-count slabs; `own` = the public indices this rank holds
    own = M.slab_partition(st["x"], R.world)[lo:hi] if args.config5_shards == "slabs" else np.arange(lo, hi)
    sw = M.Swarm(n, device=R.local, arith=M.ARITH_FAST if args.arith == "fast" else M.ARITH_LITERAL)
    sw.construct(0, n, M.model_params("x500", ground_enabled=True))
    sw.set_state(0, n, st["x"][own], st["v"][own], st["R"][own], st["omega"][own], st["motor_rpm"][own])
    sw.set_input(0, n, M.POSITION_CMD, cmd[own])
    del st, cmd
    transport = args.config5_transport if R.use_dist else "rccl"
    if args.config5_transport == "peer" and R.use_dist:
        _, handle = sw.peer_window_create(R.world, R.rank, n_total)
        handles = [None] * R.world
        dist.all_gather_object(handles, handle)
        sw.comm_init_peer(handles=handles)
    elif R.rehearsal:
        # rehearsal: RCCL refuses two ranks on one device — the collective is the caller-supplied one of the C ABI
        # (mrs_swarm_comm_init_custom): device -> host -> gloo all_gather -> device, blocking (tests/test_sharded_multiprocess_gpu.py)
        transport = "host all-gather over gloo (rehearsal)"
        sw.comm_init_custom(R.world, R.rank, n_total, _gloo_allgather(R))
    else:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if R.rank == 0:
            uid = torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8).cuda()
        if R.use_dist:
            dist.broadcast(uid, 0)
        sw.comm_init(R.world, R.rank, bytes(uid.cpu().numpy().tobytes()), n_total)
    sw.set_exchange(M.EXCHANGE_FULL_GATHER if args.config5_exchange == "full" else M.EXCHANGE_EXPORT_SETS)

    def run(k):
        sw.tick_sharded_n(DT, k, True, False, 100.0)

    def sync_local():
        sw.synchronize()
        torch.cuda.synchronize()

    # regions of at least 200 ticks: a search comes every ~28 ticks and a call starts with two ticks in the serial form, so
    # the driver's K = 20 would time the start-up of a call, not the tick
    steps, warmup = max(args.steps, 200), min(max(args.warmup, 30), 60)
    if os.environ.get("MRS_BENCH_KILL_RANK") == str(R.rank) and R.world > 1:  # test hook: this rank is lost inside the leg
        run(warmup)
        os.kill(os.getpid(), 9)
    times, _ = timed_regions(R, run, sync_local, steps, warmup, args.min_measure_ms)
    el = float(np.median(times))
    info = sw.comm_info()
    ticks, searches = sw.collision_stats()
    out = {"workload": f"BASELINE configs[4]: {n_total} x500 UAVs, position references + collisions, {R.world} shard(s)",
           "value": n_total * steps / el, "unit": "UAV-steps/s", "ms_per_tick": el / steps * 1e3, "n_total": n_total, "n_gpus": R.world,
           "steps": steps, "warmup": warmup, "regions": len(times), "scaling": "strong",
           "parallelism": info["parallelism"], "rccl_ranks": info["rccl_ranks"],
           "transport": transport, "n_devices": 1 if R.rehearsal else R.world,
           "shards": args.config5_shards,
           "collective_bytes_per_rank_per_tick": info["bytes_per_tick"], "collective_bytes_per_rank_per_search_tick": info["bytes_per_rebuild"],
           "export_set_of_rank0": info["export_count"], "export_capacity": info["export_capacity"], "uavs_per_rank": n,
           "sharded_ticks": info["ticks"], "search_ticks": info["searches"], "replayed_noop_ticks": info["noop_ticks"]}
    _, out["halo_searches"], out["halo_repeats"], _ = sw.search_stats()  # searches that exchanged halos instead of all records / had to be repeated on all
    if R.use_dist:
        dist.barrier()  # (peer windows: nobody unmaps a window a peer may still write into)
    sw.comm_destroy()
    del sw
    return out if R.rank == 0 else None


def peer_leg_in_children(args):
    """BASELINE configs[4] once more with the peer-window exchange (direct writes into the peers' device memory, no collective library in
    the tick) — in CHILD processes of rank 0, after the run's own process group is gone: the exchange has never run across devices, and
    a fault there must cost this record only.  Returns the child's config-5 record, or {"error": ...}."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--only-config5", "--config5-transport", "peer",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--arith", args.arith, "--config5-uavs", str(args.config5_uavs),
           "--config5-shards", args.config5_shards, "--config5-exchange", args.config5_exchange, "--volume-per-uav", str(args.volume_per_uav),
           "--min-measure-ms", str(args.min_measure_ms)]
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
            "MASTER_ADDR", "MASTER_PORT")
    env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.config5_timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {args.config5_timeout:.0f} s (child processes killed)", "transport": "peer"}
    except OSError as e:
        return {"error": f"{type(e).__name__}: {e}", "transport": "peer"}
    for ln in reversed(r.stdout.splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                break
    return {"error": f"child run ended with status {r.returncode} and no record: {r.stderr[-300:]}", "transport": "peer"}


def main():
    args = parse()
    if args.explain:
        print(json.dumps(NOTES, indent=1))
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    # stdout carries the ONE JSON line and nothing else: libraries that print banners there (RCCL's version block) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.only_config5:  # (the child run of peer_leg_in_children: one leg, one JSON object)
        R = Ranks()
        try:
            c5 = config5_leg(args, R)
        except Exception as e:  # noqa: BLE001 - the parent records it
            c5 = {"error": f"{type(e).__name__}: {e}", "transport": args.config5_transport}
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        if R.rank == 0:
            print(json.dumps(c5), flush=True)
        os._exit(0 if (c5 is None or "error" not in c5) else 3)  # (no waiting for ranks that may be stuck in a collective)
    subs = args.sub_records == "on" or (args.sub_records == "auto" and args.gpus == 1 and args.workload == "actuator"
                                        and args.uavs == 100_000 and args.substeps == 1 and not args.pmc_child)
    # roofline.traffic: the rocprofv3 --pmc child runs come FIRST, while this process has not initialised the GPU (children of a
    # process that holds the device are not started), and not at all when this run is itself being profiled
    live = {}
    if args.traffic == "live" and args.gpus == 1 and not args.pmc_child:
        profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
        jobs = [(args.uavs, args.workload)] + ([(HBM_STREAMING_UAVS, "actuator"), (100_000, "position+collisions")] if subs else [])
        for job in jobs:
            live[job] = (None, "this run is itself under a profiler") if profiled else live_traffic(args, *job)
    no_traffic = (None, "not requested")
    R = Ranks()
    if R.world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started {R.world} rank(s)")
    out, st, cmd = step_leg(args, R, args.uavs, args.workload, args.steps, args.warmup, traffic=live.get((args.uavs, args.workload), no_traffic))
    if args.pmc_child:
        R.close()
        return
    if R.rank == 0 and R.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, st, cmd)
    del st, cmd
    if subs:
        # the numbers the headline size cannot show (each a few seconds): the same kernel where every byte comes from HBM, and BASELINE
        # config 4 (collisions + ground) in the form tick_n runs it — fused step + collision launches, searches queued ahead
        rec, _, _ = step_leg(args, R, HBM_STREAMING_UAVS, "actuator", 200, 50, traffic=live.get((HBM_STREAMING_UAVS, "actuator"), no_traffic), min_ms=150.0)
        out["hbm_streaming"] = sub_record(rec)
        rec, st4, cmd4 = step_leg(args, R, 100_000, "position+collisions", 300, 100, traffic=live.get((100_000, "position+collisions"), no_traffic), min_ms=30.0)
        if R.rank == 0 and not args.no_cpu_baseline:  # incl. the one CPU number that is the reference's OWN code: its nanoflann broadphase
            rec["cpu_baseline"] = cpu_baseline(args, st4, cmd4, workload="position+collisions", uavs=100_000, seconds=min(args.cpu_seconds, 8.0))
        del st4, cmd4
        out["config4"] = sub_record(rec)
        # the same swarm spawned in the order mrs_cell_order suggests (indices follow space: VERDICT r4 item 1's data point)
        keep_order, args.order = args.order, "morton"
        rec, _, _ = step_leg(args, R, 100_000, "position+collisions", 300, 100, min_ms=30.0)
        args.order = keep_order
        if rec is not None:
            out["config4"]["spawned_in_cell_order"] = {"order": "morton (mrs_cell_order)", "ms_per_step": rec["ms_per_step"], "device_ms_per_step": rec["device_ms_per_step"],
                                                       "search_ms": rec["roofline_collision"]["search_ms"]}
        # the bit-faithful flavour (reference operation order, no FMA contraction) on the headline workload
        rec, _, _ = step_leg(args, R, 100_000, "actuator", 300, 50, min_ms=30.0, arith="literal")
        out["literal"] = mini_record(rec)
        # BASELINE configs[1]: 400 f550 hexarotors, position cascade, no collisions (launch-bound: one wave per seven SIMDs)
        rec, st2, cmd2 = step_leg(args, R, 400, "config2", 2000, 200, min_ms=30.0, airframe="f550")
        if R.rank == 0 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args, st2, cmd2, workload="config2", uavs=400, airframe="f550", seconds=min(args.cpu_seconds, 4.0))
        out["config2"] = mini_record(rec)
        # the same swarm with ten makeStep rounds per launch (mrs_swarm_step_n's substeps_per_launch: state kept in registers across the
        # rounds — legal while commands are constant and collisions are off, results identical); no roofline credit for the fusion
        rec, _, _ = step_leg(args, R, 400, "config2", 2000, 200, min_ms=30.0, airframe="f550", substeps=10)
        out["config2"]["fused_10_substeps_per_launch"] = {k: rec[k] for k in ("value", "ms_per_step", "device_ms_per_step")}
        out["sharded_rank_standin"] = sharded_rank_record(args)
        out["io_tick"] = io_tick_record(args)

    import threading
    emit_lock, emitted = threading.Lock(), []

    def emit(extra=None):
        """prints the ONE JSON line exactly once, whichever thread gets here first (main thread or the config-5 watchdog)"""
        with emit_lock:
            if emitted:
                return False
            emitted.append(True)
            if R.rank == 0 and extra:
                out.update(extra)
            sys.stdout.flush()
            os.dup2(json_fd, 1)
            if R.rank == 0:
                out["notes"] = "python bench.py --explain; DESIGN.md 6"
                if R.rehearsal:
                    out["rehearsal"] = "MRS_BENCH_REHEARSAL=1: every rank on cuda:0, gloo process group — code path only, NOT a multi-GPU measurement"
                    out["n_devices"] = 1
                line = compact_line(out)
                dropped = []
                # the driver keeps the last 8 KB of stdout: should the line ever outgrow that, the least important records go first
                # (and the line says which) rather than the head of the line being cut off
                for k in ("first_region_wall_ms_per_step", "literal", "config2", "sharded_rank_standin", "io_tick", "cpu_baseline"):
                    if len(json.dumps(line)) <= 7900:
                        break
                    if k in line:
                        del line[k]
                        dropped.append(k)
                        line["dropped_for_size"] = dropped
                print(json.dumps(line), flush=True)
            return True

    if args.config5 == "on" or (args.config5 == "auto" and not args.pmc_child and args.workload == "actuator"):
        # The headline line must not depend on this second leg: if it fails or does not come back (a rank lost, a collective that
        # never completes), every rank gives up after --config5-timeout, rank 0 prints the line with the failure recorded, and the
        # processes leave without waiting for each other.
        # A failed leg still prints the headline line (with config5.error) but the process then leaves with a NON-ZERO status, without
        # waiting for the other ranks and without re-executing anything: the launcher and the driver can tell it from success.
        def give_up(why):
            emit({"config5": {"error": why}})
            os._exit(3)

        watchdog = threading.Timer(args.config5_timeout, give_up, args=(f"no result within {args.config5_timeout:.0f} s",))
        watchdog.daemon = True
        watchdog.start()
        if R.world > 1:
            # torch.distributed.run answers a failed rank by sending SIGTERM to the others (SIGKILL 30 s later).  The main thread
            # may sit in a collective or a synchronize (no Python signal handler runs there): the C-level handler writes the signal
            # number into a pipe, a thread reads it and prints the headline line with the failure recorded
            import signal
            rd, wr = os.pipe()
            os.set_blocking(wr, False)
            signal.set_wakeup_fd(wr, warn_on_full_buffer=False)
            signal.signal(signal.SIGTERM, lambda *_: None)

            def on_signal():
                while True:
                    b = os.read(rd, 1)
                    if b and b[0] == signal.SIGTERM:
                        give_up("terminated by the launcher while the config-5 leg was running (another rank was lost)")

            threading.Thread(target=on_signal, daemon=True).start()
        try:
            c5 = config5_leg(args, R)
        except Exception as e:  # noqa: BLE001 - recorded in the line, the other ranks run into their own timeout
            give_up(f"{type(e).__name__}: {e}")
        watchdog.cancel()
        extra = {"config5": c5}
        peer = args.config5_peer == "on" or (args.config5_peer == "auto" and R.world > 1 and args.config5_transport == "rccl")
        if peer:
            # the run's own process group ends first (every rank's RCCL result is in hand); the guarded leg then runs in children of rank 0
            # and can only add a record — a failure there is written into it and changes neither the other records nor the exit status
            R.close()
            if R.rank == 0:
                extra["config5_peer"] = peer_leg_in_children(args)
        if not emit(extra):  # the watchdog fired between the leg's return and cancel(): its line (and exit status) stand
            time.sleep(60)
            os._exit(3)
        if not peer:
            R.close()
        return
    R.close()
    emit()


if __name__ == "__main__":
    main()
