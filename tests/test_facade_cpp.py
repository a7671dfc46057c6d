"""The C++ facade (include/mrs_multirotor_simulator/uav_system/*.hpp) used like the reference's own header: it must
compile with g++ against libmrs_swarm.so (CPU check) and reproduce the oracle when run on the GPU."""
import os
import subprocess

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "facade_test")


def build_exe(mrs, name="facade_test"):
    from mrs_multirotor_simulator_amd import swarm
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    exe = os.path.join(ROOT, "tests", "cpp", name)
    libdir = os.path.dirname(swarm.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-DMRS_NO_EIGEN", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
           "-L", libdir, "-lmrs_swarm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_facade_compiles_against_the_c_abi(mrs):
    assert os.path.exists(build_exe(mrs))
    assert os.path.exists(build_exe(mrs, "facade_loop_test"))


@pytest.mark.gpu
def test_unchanged_per_uav_loop_over_pooled_objects(mrs, oracle):
    """tests/cpp/facade_loop_test.cpp: the reference's loop `for (i) uavs_[i]->makeStep(dt)` (src/multirotor_simulator.cpp:211-213)
    with getState() after each call (src/uav_system_ros.cpp:270-282) over 400 stand-alone UavSystem objects — oracle parity of the
    final states, at most two kernel launches per tick (the round's step launch + the state pack), the disturbances undone slot by
    slot; the time per call is printed for INTEGRATION.md §2."""
    exe = build_exe(mrs, "facade_loop_test")
    out = subprocess.run([exe], capture_output=True, text=True, check=True, timeout=600).stdout
    rows = [ln.split() for ln in out.splitlines()]
    stats = {r[0]: r[1:] for r in rows if r[0] not in ("STATE", "LATE0", "LATE3")}
    st = {k: int(v) for k, v in zip(stats["STATS"][0::2], stats["STATS"][1::2])}
    timed = {k: int(v) for k, v in zip(stats["TIMED"][0::2], stats["TIMED"][1::2])}
    n, ticks, dt = 400, 300, 0.001
    O = oracle
    p = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    o = O.OracleSwarm(n)
    pos = np.array([[4.0 * (i // 20), 4.0 * (i % 20), 0.0] for i in range(n)])
    o.construct(0, n, p, pos, 0.01 * np.arange(n))
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(o, nm)(0, n)
    o.set_input(0, n, O.ACTUATOR_CMD, np.zeros((n, 4)))
    o.step_n(0.01, 2)
    cmd = np.array([[pos[i, 0] + 1.0, pos[i, 1] - 2.0, 3.0 + 0.01 * i, 0.001 * i] for i in range(n)])
    o.set_input(0, n, O.POSITION_CMD, cmd)
    for tick in range(ticks):
        if tick == 120:
            cmd[17] = [0.0, 0.0, 9.0, 1.0]
            o.set_input(17, 1, O.POSITION_CMD, cmd[17:18])
            o.apply_force(300, 1, [[1.0, -2.0, 0.5]])
        if tick == 150:  # UAV 40 steps 2 dt, everybody else dt
            o.set_hold(40, 1, True)
            o.step(dt)
            o.set_hold(40, 1, False)
            o.set_hold(0, 40, True)
            o.set_hold(41, n - 41, True)
            o.step(2 * dt)
            o.set_hold(0, n, False)
        else:
            o.step(dt)
        if tick == 200:
            for dst, src in ((5, 6), (7, 8)):
                s = o.get_state(src, 1)
                o.set_state(dst, 1, s["x"], s["v"], s["R"], s["omega"], s["motor_rpm"])
                o.set_pid(dst, 1, o.get_pid(src, 1))
                cmd[dst] = cmd[src]
                o.set_input(dst, 1, O.POSITION_CMD, cmd[dst:dst + 1])
    so = o.get_state()
    seen = 0
    for r in rows:
        if r[0] != "STATE":
            continue
        i, v = int(r[1]), np.array(r[2:], dtype=float)
        got = dict(x=v[0:3], v=v[3:6], R=v[6:15].reshape(3, 3), omega=v[15:18], motor_rpm=v[18:22])
        for k in ("x", "v", "R", "omega"):
            helpers.assert_close(got[k], so[k][i], helpers.RTOL_NORTH_STAR, f"UAV {i}: {k}")
        helpers.assert_close(got["motor_rpm"], so["motor_rpm"][i, :4], helpers.RTOL_NORTH_STAR, f"UAV {i}: rpm")
        seen += 1
    assert seen == 9
    # UavSystem() constructed after 300 rounds (one in a reused slot, one in a slot no object ever held): the reference's zero state,
    # and the same first steps as a fresh oracle object
    late = O.OracleSwarm(1)
    s0 = late.get_state(0, 1)
    late.set_input(0, 1, O.ACTUATOR_CMD, [[0.55, 0.56, 0.57, 0.58]])
    late.step_n(dt, 3)
    s3 = late.get_state(0, 1)
    slots = [int(v) for v in stats["LATESLOTS"]]
    assert slots[0] < n <= slots[1], slots  # (reused, never used)
    n_late = 0
    for r in rows:
        if r[0] not in ("LATE0", "LATE3"):
            continue
        ref, v = (s0 if r[0] == "LATE0" else s3), np.array(r[2:], dtype=float)
        got = dict(x=v[0:3], v=v[3:6], R=v[6:15].reshape(3, 3), omega=v[15:18], motor_rpm=v[18:22])
        for k in ("x", "v", "R", "omega"):
            helpers.assert_close(got[k], ref[k][0], helpers.RTOL_FAST, f"late object {r[1]} {r[0]}: {k}")
        helpers.assert_close(got["motor_rpm"], ref["motor_rpm"][0, :4], helpers.RTOL_FAST, f"late object {r[1]} {r[0]}: rpm")
        n_late += 1
    assert n_late == 4
    # launches: every timed tick is ONE round (step launch + state pack = two kernels) but for the disturbances
    assert timed["rounds"] >= timed["ticks"] - 3 and timed["single_steps"] <= 8, (timed, st)
    # (the first tick after construction is the pool's observation round: every object steps on its own and reads the device once)
    assert st["rollbacks"] <= 8 and st["single_steps"] <= n + 8 and st["state_misses"] <= n + 16 and st["state_hits"] >= 0.98 * n * (ticks - 1), st
    print("facade loop:", " ".join(stats["LATENCY_US_PER_CALL"]), "us per makeStep + getState call,", " ".join(stats["TICK_US"]), "us per 400-UAV tick;", st)
    # the same objects stepped one by one (what every facade call cost before the pool): a launch + a synchronisation per call
    single = subprocess.run([exe, "single"], capture_output=True, text=True, check=True, timeout=600).stdout
    lat = [ln.split()[1] for ln in single.splitlines() if ln.startswith("LATENCY_US_PER_CALL")][0]
    print("facade loop, one launch per call:", lat, "us per makeStep + getState call")
    assert float(lat) > 3.0 * float(stats["LATENCY_US_PER_CALL"][0])


@pytest.mark.gpu
def test_facade_matches_oracle(mrs, oracle):
    exe = build_exe(mrs)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    rows = {ln.split()[0]: ln.split()[1:] for ln in out.splitlines()}

    def unpack(tag, n_motors=4):
        v = np.array(rows[tag], dtype=float)
        return dict(x=v[0:3], v=v[3:6], R=v[6:15].reshape(3, 3), omega=v[15:18], rpm=v[18:18 + n_motors], imu=v[18 + n_motors:])

    O = oracle
    p = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    s = O.OracleSwarm(1)
    s.construct(0, 1, p, [[10, 15, 0]], [3.14])
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(s, nm)(0, 1)
    s.set_input(0, 1, O.ACTUATOR_CMD, [[0.0] * 4])
    s.step_n(0.01, 2)

    def check(tag, sw, i=0, rtol=helpers.RTOL_FAST):
        g, st = unpack(tag), sw.get_state(i, 1)
        for k, ref in (("x", st["x"][0]), ("v", st["v"][0]), ("R", st["R"][0]), ("omega", st["omega"][0]),
                       ("rpm", st["motor_rpm"][0, :4]), ("imu", sw.get_imu(i, 1)[0])):
            helpers.assert_close(g[k], ref, rtol, f"{tag}:{k}")

    check("WARMUP", s)
    s.set_input(0, 1, O.POSITION_CMD, [[12, 13, 5, 1.0]])
    s.step_n(0.001, 1000)
    check("STEP1000", s, rtol=helpers.RTOL_NORTH_STAR)
    s.set_feedforward(0, 1, O.FF_VELOCITY_HDG, [[0.5, -0.25, 0.1, 0]])
    s.apply_force(0, 1, [[1.0, 2.0, -0.5]])
    s.step_n(0.001, 100)
    check("STEP1100", s, rtol=helpers.RTOL_NORTH_STAR)
    assert rows["ALLOC"][:2] == ["4", "4"] and abs(float(rows["ALLOC"][2]) + np.sqrt(0.5)) < 1e-12 and float(rows["ALLOC"][3]) == 1.0
    assert rows["CRASHED"][0] == "1" and float(rows["CRASHED"][2]) == 2.0
    assert rows["POSE"][0] == "1"

    n = 400
    sw = O.OracleSwarm(n)
    pos = np.array([[4.0 * (i // 20), 4.0 * (i % 20), 0.0] for i in range(n)])
    sw.construct(0, n, p, pos, np.zeros(n))
    cmd = np.array([[pos[i, 0] + 1.0, pos[i, 1] - 2.0, 3.0 + 0.01 * i, 0.001 * i] for i in range(n)])
    sw.set_input(0, n, O.POSITION_CMD, cmd)
    for _ in range(200):
        sw.step(0.001)
        sw.handle_collisions(True, False, 100.0)
    sw.timeout_input(0, 200)
    sw.set_mass(100, 50, 2.4)
    sw.step_n(0.001, 50)
    o = sw.get_outputs()
    got = np.array(rows["OUT7"], dtype=float)
    helpers.assert_close(got, [o["orientation"][7, 3], o["velocity_body"][7, 0], o["range"][7], o["position"][120, 2],
                               o["linear_acceleration"][120, 2]], 1e-11, "packed outputs")
    check("SWARM7", sw, 7, rtol=helpers.RTOL_LITERAL)
    check("SWARM399", sw, 399, rtol=helpers.RTOL_LITERAL)
