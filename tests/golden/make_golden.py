#!/usr/bin/env python3
"""Generates the committed fixtures of tests/golden/ (run in the build container, where /root/reference is mounted).

* nanoflann_radius_sets.npz — OUTPUT OF THE REFERENCE'S OWN CODE: include/nanoflann.hpp +
  KDTreeVectorOfVectorsAdaptor.h compiled unmodified (oracle/Makefile -> oracle/_ref) and queried exactly like
  src/multirotor_simulator.cpp:309-328 (3 dims, leaf_max_size 10, RadiusResultSet(3.0)) on seeded point clouds.
  Inputs (points) and outputs (CSR neighbour lists with squared distances) are stored.
* pid_reference_vectors.npz — OUTPUT OF THE REFERENCE'S OWN CODE: PIDController of
  include/mrs_multirotor_simulator/uav_system/controllers/pid.hpp (the one controller header that needs neither Eigen nor Boost),
  compiled unmodified into oracle/_ref/libref_pid.so and driven with seeded (error, dt) sequences; parameters, stimulus and the
  outputs of update() are stored.
* oracle_trajectories.npz — regression vectors of the CPU oracle (NOT reference outputs: the reference's dynamics cannot be
  built here) for BASELINE config 1 and a 64-UAV mixed-mode swarm; they freeze the oracle so that a later
  edit of oracle/uav_oracle.c cannot silently move the parity target.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from oracle import oracle_swarm as O  # noqa: E402


def nanoflann_sets():
    assert O.ref_lib() is not None, "oracle/_ref must be built from /root/reference"
    rng = np.random.default_rng(2025)
    clouds = {
        "dense": rng.uniform(0, 10, (600, 3)),
        "sparse": rng.uniform(-100, 100, (800, 3)),
        "grid12": np.stack(np.meshgrid(np.arange(12) * 1.2, np.arange(12) * 1.2, [0.0, 1.2], indexing="ij"), -1).reshape(-1, 3),
        "tmux400": np.stack(np.meshgrid(np.arange(20) * 4.0, np.arange(20) * 4.0, [0.0], indexing="ij"), -1).reshape(-1, 3),
    }
    out = {}
    for name, pts in clouds.items():
        off, idx, d2 = O.ref_radius_neighbours(pts, 3.0, 10)
        # canonical order inside a query (the kd-tree's traversal order is an implementation detail)
        for i in range(len(pts)):
            o = np.argsort(idx[off[i]:off[i + 1]], kind="stable")
            idx[off[i]:off[i + 1]] = idx[off[i]:off[i + 1]][o]
            d2[off[i]:off[i + 1]] = d2[off[i]:off[i + 1]][o]
        out[f"{name}_points"], out[f"{name}_offsets"], out[f"{name}_indices"], out[f"{name}_d2"] = pts, off, idx, d2
    np.savez_compressed(os.path.join(HERE, "nanoflann_radius_sets.npz"), **out)


def oracle_trajectories():
    out = {}
    p = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    s = O.OracleSwarm(1)
    s.construct(0, 1, p, [[10, 15, 0]], [3.14])
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(s, nm)(0, 1)
    s.set_input(0, 1, O.ACTUATOR_CMD, [[0.0] * 4])
    s.step_n(0.01, 2)
    s.set_input(0, 1, O.POSITION_CMD, [[12, 13, 5, 1.0]])
    rows = []
    for k in range(20):
        s.step_n(0.001, 1000)
        st = s.get_state()
        rows.append(np.concatenate([st["x"][0], st["v"][0], st["R"][0].ravel(), st["omega"][0], st["motor_rpm"][0, :4], s.get_imu()[0]]))
    out["config1_every_1000_steps"] = np.array(rows)

    rng = np.random.default_rng(64)
    n = 64
    s = O.OracleSwarm(n)
    names = ["x500", "f550", "naki", "t650"]
    modes, payloads = [], []
    st0 = helpers.random_state(rng, n, 8, tilted=True)
    for i in range(n):
        af = names[i % 4]
        s.construct(i, 1, helpers.oracle_params(af), [st0["x"][i]], [0.0])
        for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
            getattr(s, nm)(i, 1)
    s.set_state(0, n, st0["x"], st0["v"], st0["R"], st0["omega"], st0["motor_rpm"])
    for i in range(n):
        mode = [O.POSITION_CMD, O.VELOCITY_HDG_CMD, O.ACCELERATION_HDG_RATE_CMD, O.ATTITUDE_RATE_CMD, O.ACTUATOR_CMD][i % 5]
        if mode == O.ACTUATOR_CMD:
            pl = np.concatenate([rng.uniform(0.4, 0.6, 8)])
        elif mode == O.POSITION_CMD:
            pl = np.concatenate([st0["x"][i] + rng.uniform(-3, 3, 3), rng.uniform(-3, 3, 1)])
        elif mode == O.ATTITUDE_RATE_CMD:
            pl = np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(0.4, 0.6, 1)])
        else:
            pl = np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-1, 1, 1)])
        pl = np.pad(pl, (0, 8 - len(pl)))
        s.set_input(i, 1, mode, pl[None, :])
        modes.append(mode)
        payloads.append(pl)
    s.step_n(0.001, 200)
    st = s.get_state()
    out.update(mixed_x0=st0["x"], mixed_v0=st0["v"], mixed_R0=st0["R"], mixed_w0=st0["omega"], mixed_rpm0=st0["motor_rpm"],
               mixed_modes=np.array(modes), mixed_payloads=np.array(payloads), mixed_x=st["x"], mixed_v=st["v"], mixed_R=st["R"],
               mixed_w=st["omega"], mixed_rpm=st["motor_rpm"], mixed_imu=s.get_imu(), mixed_pid=s.get_pid())
    np.savez_compressed(os.path.join(HERE, "oracle_trajectories.npz"), **out)


def all_modes_payload(rng, mode, x, n_motors):
    """one setInput payload of `mode` (layouts of mrs_swarm.h), padded to 10 doubles"""
    import helpers
    if mode == O.ACTUATOR_CMD:
        pl = rng.uniform(0.35, 0.65, n_motors)
    elif mode == O.CONTROL_GROUP_CMD:
        pl = np.concatenate([rng.uniform(-0.15, 0.15, 3), rng.uniform(0.4, 0.6, 1)])
    elif mode == O.ATTITUDE_RATE_CMD:
        pl = np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(0.4, 0.6, 1)])
    elif mode == O.ATTITUDE_CMD:
        pl = np.concatenate([helpers.tilted_rotations(rng, 1).ravel(), rng.uniform(0.4, 0.6, 1)])
    elif mode == O.TILT_HDG_RATE_CMD:
        pl = np.concatenate([rng.normal(0, 0.2, 3) + [0, 0, 1], rng.uniform(-1, 1, 1), rng.uniform(0.4, 0.6, 1)])
    elif mode in (O.ACCELERATION_HDG_RATE_CMD, O.ACCELERATION_HDG_CMD):
        pl = np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-1, 1, 1)])
    elif mode in (O.VELOCITY_HDG_RATE_CMD, O.VELOCITY_HDG_CMD):
        pl = np.concatenate([rng.uniform(-3, 3, 3), rng.uniform(-1, 1, 1)])
    else:
        pl = np.concatenate([x + rng.uniform(-3, 3, 3), rng.uniform(-3, 3, 1)])
    return np.pad(pl, (0, 10 - len(pl)))


def all_modes():
    """every input mode (ACTUATOR_CMD ... POSITION_CMD) x airframes x500 / f550 / naki (4, 6, 8 motors),
    150 steps of 1 ms from tilted random states with the ground plane on; one feed-forward slot on a third of the UAVs."""
    import helpers
    rng = np.random.default_rng(310)
    names = ["x500", "f550", "naki"]
    modes = list(range(O.ACTUATOR_CMD, O.POSITION_CMD + 1))
    n = len(names) * len(modes)
    s = O.OracleSwarm(n)
    st0 = helpers.random_state(rng, n, 8, box=20.0, zlo=1.0, zhi=20.0, tilted=True)
    frames, mode_of, payloads, ff_kind, ff_payload = [], [], [], [], []
    for i in range(n):
        af, mode = names[i % 3], modes[i // 3]
        po = helpers.oracle_params(af, ground_enabled=True, ground_z=0.0)
        s.construct(i, 1, po, [st0["x"][i]], [0.0])
        for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
            getattr(s, nm)(i, 1)
        st0["motor_rpm"][i, po.n_motors:] = 0.0
        frames.append(af); mode_of.append(mode)
    s.set_state(0, n, st0["x"], st0["v"], st0["R"], st0["omega"], st0["motor_rpm"])
    for i in range(n):
        pl = all_modes_payload(rng, mode_of[i], st0["x"][i], s.get_params(i).n_motors)
        s.set_input(i, 1, mode_of[i], pl[None, :])
        payloads.append(pl)
        kind = int(rng.integers(0, 4)) if i % 3 == 0 else -1
        ffp = np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.2, 0.2, 1)])
        if kind >= 0:
            s.set_feedforward(i, 1, kind, ffp[None, :])
        ff_kind.append(kind); ff_payload.append(ffp)
    s.step_n(0.001, 150)
    st = s.get_state()
    np.savez_compressed(os.path.join(HERE, "all_modes_trajectories.npz"), frames=np.array(frames), modes=np.array(mode_of), payloads=np.array(payloads),
                        ff_kind=np.array(ff_kind), ff_payload=np.array(ff_payload), x0=st0["x"], v0=st0["v"], R0=st0["R"], w0=st0["omega"],
                        rpm0=st0["motor_rpm"], x=st["x"], v=st["v"], R=st["R"], w=st["omega"], rpm=st["motor_rpm"], imu=s.get_imu(), pid=s.get_pid())


def pid_sequences(seed=77, n_seq=96, n_steps=160):
    """Stimulus of the PID pinning vectors: per sequence one parameter set (the four gain sets of the cascade and random ones;
    saturation and anti-windup positive, zero and negative) and a run of (error, dt) pairs that walks through the derivative
    kick, both saturation sides (with exact hits of the bounds), the anti-windup threshold, sign changes, dt changes, a NaN and an
    infinite error, and a setSaturation / reset in the middle."""
    rng = np.random.default_rng(seed)
    cascade = [(2.0, 0.15, 0.2, 6.0, 1.0), (2.0, 0.05, 0.01, 4.0, 1.0), (6.0, 0.05, 0.01, 10.0, 0.1), (6.0, 0.05, 0.01, 1.0, 0.1),
               (4.0 * 0.0329, 0.04 * 0.0329, 0.0, -1.0, 1.0)]
    params = np.zeros((n_seq, 5))
    err = np.zeros((n_seq, n_steps))
    dt = np.zeros((n_seq, n_steps))
    event = np.zeros((n_seq, n_steps))  # 0 none, 1 reset before this update, 2 setSaturation(new_sat) before this update
    new_sat = np.zeros((n_seq, n_steps))
    for q in range(n_seq):
        if q < 3 * len(cascade):
            params[q] = cascade[q % len(cascade)]
        else:
            params[q] = [rng.uniform(0, 8), rng.uniform(0, 0.3), rng.uniform(0, 0.5), rng.choice([-1.0, 0.0, 0.5, 2.0, 6.0]),
                         rng.choice([-1.0, 0.0, 0.1, 1.0, 3.0])]
        scale = 10.0 ** rng.uniform(-3, 1.5)
        e = np.cumsum(rng.normal(0, 0.2, n_steps)) * scale + rng.normal(0, 0.05, n_steps) * scale
        e[rng.integers(0, n_steps, 6)] *= -1.0
        kp, sat = params[q, 0], params[q, 3]
        if kp > 0 and sat > 0:  # exact hits of the bounds through the proportional term alone would need kd = ki = 0; near hits do
            e[rng.integers(1, n_steps)] = sat / kp
            e[rng.integers(1, n_steps)] = -sat / kp
        err[q] = e
        dt[q] = rng.choice([0.001, 0.01, 0.004, 1.0 / 3.0 * 0.01], n_steps) if q % 3 == 0 else rng.choice([0.001, 0.01])
        if q % 8 == 5:
            err[q, n_steps // 2] = np.nan
        if q % 8 == 6:
            err[q, n_steps // 2] = np.inf
        if q % 4 == 1:
            event[q, n_steps // 3] = 1
        if q % 4 == 2:
            event[q, 2 * n_steps // 3] = 2
            new_sat[q, 2 * n_steps // 3] = rng.choice([0.5, 3.0, -1.0])
    return params, err, dt, event, new_sat


def pid_reference():
    """Golden vectors from the REFERENCE's own PIDController (controllers/pid.hpp compiled where it lies into
    oracle/_ref/libref_pid.so): the outputs of update() for the sequences above.  tests/test_pid_ref.py holds the oracle to them."""
    L = O.ref_pid_lib()
    assert L is not None, "oracle/_ref/libref_pid.so must be built from /root/reference"
    params, err, dt, event, new_sat = pid_sequences()
    out = np.zeros_like(err)
    for q in range(len(params)):
        h = L.ref_pid_create()
        L.ref_pid_set_params(h, *params[q])
        for k in range(err.shape[1]):
            if event[q, k] == 1:
                L.ref_pid_reset(h)
            elif event[q, k] == 2:
                L.ref_pid_set_saturation(h, new_sat[q, k])
            out[q, k] = L.ref_pid_update(h, err[q, k], dt[q, k])
        L.ref_pid_destroy(h)
    np.savez_compressed(os.path.join(HERE, "pid_reference_vectors.npz"), params=params, err=err, dt=dt, event=event, new_sat=new_sat, out=out)
    print("pid_reference_vectors.npz:", out.shape, "outputs,", int(np.isnan(out).sum()), "NaN")


if __name__ == "__main__":
    nanoflann_sets()
    oracle_trajectories()
    all_modes()
    pid_reference()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
