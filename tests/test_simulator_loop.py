"""SURVEY 8f ranks 1 and 3 — the simulator loop around the hot path (include/mrs_multirotor_simulator/multirotor_simulator.hpp):
sim clock, tick order, input watchdog / hold mask, RTF telemetry, pacing, pause, randd.  The loop logic is unit-tested on the CPU
with a recording stand-in for the swarm; the hold mask itself is checked on the GPU against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import helpers
from helpers import RTOL_LITERAL, Pair

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = 0.001


def test_simulator_loop_logic_cpp():
    exe = os.path.join(ROOT, "tests", "cpp", "simulator_logic_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-DMRS_NO_EIGEN", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "simulator_logic_test.cpp"), "-o", exe, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    for tag in ("clock", "clock_same_rate", "watchdog", "watchdog_iterate", "rtf", "pacing", "randd"):
        assert f"ok {tag}" in out.stdout


def test_oracle_hold_mask_skips_the_model():
    """src/uav_system_ros.cpp:265: a UAV without input is not iterated when iterate_without_input is false."""
    from oracle import oracle_swarm as O
    o = O.OracleSwarm(3)
    o.construct(0, 3, helpers.oracle_params("x500"), np.array([[0, 0, 5.0]] * 3), np.zeros(3))
    o.set_input(0, 3, O.ACTUATOR_CMD, np.full((3, 4), 0.3))
    o.set_hold(1, 1, True)
    before = o.get_state()
    o.step_n(DT, 20)
    after = o.get_state()
    assert np.array_equal(after["x"][1], before["x"][1]) and np.array_equal(after["motor_rpm"][1], before["motor_rpm"][1])
    assert after["x"][0][2] < before["x"][0][2] and np.array_equal(after["x"][0], after["x"][2])
    o.set_hold(1, 1, False)
    o.step_n(DT, 20)
    assert o.get_state()["x"][1][2] < before["x"][1][2]


@pytest.mark.gpu
def test_hold_mask_matches_oracle(mrs, oracle):
    """UAVs on hold keep state, PIDs and IMU through steps and ticks; collisions still see them; release resumes."""
    rng = np.random.default_rng(31)
    n = 300
    p = Pair(mrs, n)
    pos = rng.uniform(0, 14, (n, 3)) + [0, 0, 10]
    p.construct(0, n, "x500", pos=pos, heading=rng.uniform(-3, 3, n))
    goals = np.concatenate([pos + rng.uniform(-3, 3, (n, 3)), rng.uniform(-3, 3, (n, 1))], axis=1)
    p.both("set_input", 0, n, oracle.POSITION_CMD, goals)
    p.step(DT, 30)
    held = np.zeros(n, bool)
    held[40:170] = True   # covers whole 64-UAV blocks and partial ones
    held[200] = True
    for lo, hi in ((40, 170), (200, 201)):
        p.both("set_hold", lo, hi - lo, True)
    frozen = p.g.get_state()
    for _ in range(40):
        p.o.step(DT)
        p.o.handle_collisions(True, False, 100.0)
    p.g.tick_n(DT, 40, True, False, 100.0)
    p.compare(RTOL_LITERAL, "with hold")
    now = p.g.get_state()
    for k in ("x", "v", "R", "omega", "motor_rpm"):
        assert np.array_equal(now[k][held], frozen[k][held]), k
    assert not np.array_equal(now["x"][~held], frozen["x"][~held])
    helpers.assert_close(p.g.get_external_force(), p.o.get_external_force(), 1e-12, "forces on and from held UAVs")
    assert np.abs(p.o.get_external_force()[held]).sum() > 0
    # the watchdog's other half: safe command, then release
    p.both("timeout_input", 40, 130)
    p.both("set_hold", 40, 130, False)
    p.both("set_hold", 200, 1, False)
    p.step(DT, 40)
    p.compare(RTOL_LITERAL, "after release")
    assert not np.array_equal(p.g.get_state()["x"][held], frozen["x"][held])


def _build_cpp(mrs, name):
    from mrs_multirotor_simulator_amd import swarm
    exe = os.path.join(ROOT, "tests", "cpp", name)
    libdir = os.path.dirname(swarm.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-DMRS_NO_EIGEN", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe, "-L", libdir, "-lmrs_swarm", "-lpthread",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_simulator_over_the_swarm_compiles(mrs):
    assert os.path.exists(_build_cpp(mrs, "simulator_gpu_test"))


@pytest.mark.gpu
def test_simulator_over_the_swarm_on_gpu(mrs):
    """MultirotorSimulator<UavSwarm>: frozen without input, only commanded UAVs move, timeout -> hover command + hold, pacing, and the
    pipelined publisher (every tick's payload once, one tick late, equal to the synchronous download of a twin swarm)."""
    out = subprocess.run([_build_cpp(mrs, "simulator_gpu_test")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    for tag in ("frozen_without_input", "only_commanded_uavs_move", "timeout_puts_on_hold", "paced", "pipelined_publisher"):
        assert f"ok {tag}" in out.stdout, out.stdout


def test_sharded_tick_host_compiles(mrs):
    assert os.path.exists(_build_cpp(mrs, "sharded_tick_test"))


@pytest.mark.gpu
def test_library_driven_sharded_tick_from_a_cpp_host(mrs):
    """mrs_swarm_comm_init / tick_sharded_n in a process without PyTorch: RCCL from the system loader path, one-rank communicator,
    same results as the local ticks, crash mode, destroy; then three slab shards of one swarm on std::threads over a loopback group
    (export-set exchange) against the whole swarm."""
    out = subprocess.run([_build_cpp(mrs, "sharded_tick_test")], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    for tag in ("communicator", "sharded_ticks_equal_local_ticks", "crash_and_destroy", "loopback_export_sets"):
        assert f"ok {tag}" in out.stdout, out.stdout


def _build_example(mrs):
    from mrs_multirotor_simulator_amd import swarm
    exe = os.path.join(ROOT, "tests", "cpp", "standalone_swarm")
    libdir = os.path.dirname(swarm.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-DMRS_NO_EIGEN", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "standalone_swarm.cpp"), "-o", exe, "-L", libdir, "-lmrs_swarm", "-lpthread",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_standalone_example_compiles(mrs):
    assert os.path.exists(_build_example(mrs))


@pytest.mark.gpu
def test_standalone_example_runs(mrs):
    """examples/standalone_swarm.cpp on the sample parameter file: config -> swarm -> paced loop; the first UAV climbs towards its goal."""
    out = subprocess.run([_build_example(mrs), "1.2", os.path.join(ROOT, "tests", "golden", "sample_config.yaml")], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("t_sim")]
    assert len(lines) == 2 and "3 UAVs" in out.stdout
    z = float(lines[-1].split("(")[1].split(")")[0].split(",")[2])
    assert z > 0.6, out.stdout  # spawned at 0.5 m, goal 5 m higher
    # the pipelined publisher: one payload per tick, the last one stamped with the loop's sim time
    ticks = int(lines[-1].split(" ticks ")[1].split(",")[0])
    published = int(lines[-1].split("published")[1].split()[0])
    assert published == ticks and abs(float(lines[-1].split("last stamped")[1].split()[0]) - float(lines[-1].split()[1])) < 1e-9, lines[-1]
