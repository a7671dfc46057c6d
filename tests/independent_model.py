"""A SECOND, independent CPU restatement of the hot path, for cross-checking oracle/uav_oracle.c — test infrastructure only.

Why it exists: the dynamics/cascade part of the oracle is "parity unpinned" (the reference ships no tests or golden vectors and
cannot be built here — DESIGN.md §2).  This file restates the same reference functions a second time, written straight from the
reference headers in a different style — one object per UAV, whole-matrix numpy expressions, LAPACK for Cholesky / inverses where
the reference calls Eigen's `LLT`, `inverse()` — so that a slip of the pen in either restatement (a transposed index, a wrong
sign, a swapped operand, a forgotten branch) shows up as a disagreement.  Leaf arithmetic differs in the last bits (LAPACK vs the
hand-rolled Eigen orders of the oracle), hence the comparison tolerance of 1e-9 in tests/test_independent_restatement.py.

Citations: MM = include/mrs_multirotor_simulator/uav_system/multirotor_model.hpp, US = .../uav_system.hpp,
CTL = .../controllers/, ODE = .../ode/boost/numeric/odeint.  Pure-Python loops: small cases only.
"""
import math

import numpy as np

(INPUT_UNKNOWN, ACTUATOR_CMD, CONTROL_GROUP_CMD, ATTITUDE_RATE_CMD, ATTITUDE_CMD, TILT_HDG_RATE_CMD, ACCELERATION_HDG_RATE_CMD,
 ACCELERATION_HDG_CMD, VELOCITY_HDG_RATE_CMD, VELOCITY_HDG_CMD, POSITION_CMD) = range(11)  # US:19-32
FF_VELOCITY_HDG_RATE, FF_VELOCITY_HDG, FF_ACCELERATION_HDG_RATE, FF_ACCELERATION_HDG = range(4)


def _normalized(v):
    """Eigen normalized()/normalize(): the vector itself when its squared norm is not positive."""
    z = float(v @ v)
    return v / math.sqrt(z) if z > 0 else v.copy()


def _re_orthonormalised(R):
    """R * L^-1 with L L^T = R^T R (MM:249-252, MM:314-316).  A failed factorisation leaves NaN, as Eigen's LLT does."""
    try:
        L = np.linalg.cholesky(R.T @ R)
        return R @ np.linalg.inv(L)
    except np.linalg.LinAlgError:
        return np.full((3, 3), np.nan)


class Pid:  # CTL/pid.hpp
    def __init__(self, kp, kd, ki, saturation, antiwindup):
        self.kp, self.kd, self.ki, self.saturation, self.antiwindup = kp, kd, ki, saturation, antiwindup
        self.last_error = 0.0
        self.integral = 0.0

    def update(self, error, dt):  # :67-96
        difference = (error - self.last_error) / dt
        self.last_error = error
        total = self.kp * error + self.kd * difference + self.ki * self.integral
        if self.saturation > 0:
            if total >= self.saturation:
                total = self.saturation
            elif total <= -self.saturation:
                total = -self.saturation
        if self.antiwindup > 0 and abs(total) < self.antiwindup:
            self.integral += error * dt
        return total


class Uav:
    """One UavSystem (US) with its MultirotorModel (MM) and controllers (CTL).  `p` is a dict of model parameters with
    J (3x3) and allocation_matrix (4 x n_motors, already scaled)."""

    def __init__(self, p, spawn_pos=None, spawn_heading=0.0):
        self.p = dict(p)
        n = self.p["n_motors"]
        # MM:141-158 initializeState
        self.x = np.zeros(3)
        self.v = np.zeros(3)
        self.v_prev = np.zeros(3)
        self.R = np.eye(3)
        self.omega = np.zeros(3)
        self.motor_rpm = np.zeros(n)
        self.input = np.zeros(n)
        self.external_force = np.zeros(3)
        self.external_moment = np.zeros(3)
        self.imu = np.zeros(3)
        self.initial_pos = np.zeros(3)
        self.takeoff_patch_enabled = bool(self.p["takeoff_patch_enabled"])  # mutated by step(), MM:275
        if spawn_pos is not None:  # MM:439-446: AngleAxis(-heading, z)
            self.initial_pos = np.array(spawn_pos, dtype=float)
            self.x = self.initial_pos.copy()
            c, s = math.cos(-spawn_heading), math.sin(-spawn_heading)
            self.R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        self.crashed = False
        self.active_input = INPUT_UNKNOWN
        self.cmd = {}
        self.ff = {}
        self.actuators = np.zeros(n)
        self.gains = dict(rate=(4.0, 0.04, 0.0), attitude=(6.0, 0.05, 0.01, 10.0, 1.0), velocity=(2.0, 0.05, 0.01, 4.0),
                          position=(2.0, 0.15, 0.2, 6.0), desaturation=True)  # header defaults of CTL/*.hpp
        self.initialize_controllers()

    # ---- US:159-169 + the controllers' constructors ----
    def initialize_controllers(self):
        p, g = self.p, self.gains
        A = np.asarray(p["allocation_matrix"], dtype=float)  # 4 x n
        inv = A.T @ np.linalg.inv(A @ A.T)  # CTL/mixer.hpp:76
        for i in range(p["n_motors"]):  # :82-99
            inv[i, 0:2] = _normalized(inv[i, 0:2])
            inv[i, 2] = 1.0 if inv[i, 2] > 1e-2 else (-1.0 if inv[i, 2] < -1e-2 else 0.0)
            inv[i, 3] = 1.0
        self.allocation_inv = inv
        J = np.asarray(p["J"], dtype=float).reshape(3, 3)
        kp, kd, ki = g["rate"]
        self.pid_rate = [Pid(kp * J[i, i], kd * J[i, i], ki * J[i, i], -1, 1.0) for i in range(3)]  # CTL/rate_controller.hpp:55-64
        kp, kd, ki, rp, yaw = g["attitude"]
        self.pid_att = [Pid(kp, kd, ki, rp, 0.1), Pid(kp, kd, ki, rp, 0.1), Pid(kp, kd, ki, yaw, 0.1)]  # attitude_controller.hpp:160-171
        kp, kd, ki, sat = g["velocity"]
        self.pid_vel = [Pid(kp, kd, ki, sat, 1.0) for _ in range(3)]
        kp, kd, ki, sat = g["position"]
        self.pid_pos = [Pid(kp, kd, ki, sat, 1.0) for _ in range(3)]

    def set_params(self, p):  # US:404-409: the gains fall back to the header defaults
        self.p = dict(p)
        self.gains = dict(rate=(4.0, 0.04, 0.0), attitude=(6.0, 0.05, 0.01, 10.0, 1.0), velocity=(2.0, 0.05, 0.01, 4.0),
                          position=(2.0, 0.15, 0.2, 6.0), desaturation=True)
        self.initialize_controllers()

    # ---- US:175-272 ----
    def set_input(self, mode, payload=None):
        self.active_input = mode
        if payload is not None:
            self.cmd[mode] = np.array(payload, dtype=float)

    def set_feedforward(self, kind, payload):
        self.ff[kind] = np.array(payload, dtype=float)

    def crash(self):
        self.crashed = True

    def apply_force(self, f):
        self.external_force = np.array(f, dtype=float)

    # ---- CTL/mixer.hpp:107-144 ----
    def mixer(self, cg):
        cg = np.array(cg, dtype=float)
        throttle = cg[3]
        motors = self.allocation_inv @ cg
        if self.gains["desaturation"]:
            lo = motors.min()
            if lo < 0.0:
                motors = motors + abs(lo)
            hi = motors.max()
            if hi > 1.0:
                if throttle > 1e-2:
                    for i in range(3):
                        cg[i] = cg[i] / (motors.mean() / throttle)
                    motors = self.allocation_inv @ cg
                else:
                    motors = motors / hi
        return motors

    # ---- CTL/attitude_controller.hpp ----
    def attitude_error_rates(self, Rd, dt):
        E = 0.5 * (Rd.T @ self.R - self.R.T @ Rd)
        e = [(E[1, 2] - E[2, 1]) / 2.0, (E[2, 0] - E[0, 2]) / 2.0, (E[0, 1] - E[1, 0]) / 2.0]
        return [self.pid_att[i].update(e[i], dt) for i in range(3)]

    def heading_rate_of_body_rate(self, w):  # :177-206
        R = self.R
        W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R_d = R @ W
        rx, ry = R[0, 0], R[1, 0]
        denom = rx * rx + ry * ry
        ax = ay = 0.0
        if abs(denom) > 1e-5:
            ax, ay = -ry / denom, rx / denom
        return ax * R_d[0, 0] + ay * R_d[1, 0]

    def yaw_rate_intrinsic(self, heading_rate):  # :212-251
        if abs(heading_rate) < 1e-3:
            return 0.0
        R = self.R
        heading_vector = np.array([R[0, 0], R[1, 0], 0.0])
        orbital_velocity = np.cross([0.0, 0.0, heading_rate], heading_vector)
        b_orb = _normalized(np.cross([0.0, 0.0, 1.0], heading_vector))
        projected = np.outer(b_orb, b_orb) @ R[:, 1]
        pn = np.linalg.norm(projected)
        if abs(pn) < 1e-5:
            return 0.0
        d = float(orbital_velocity @ projected)
        out = ((0 < d) - (d < 0)) * (np.linalg.norm(orbital_velocity) / pn)
        return float(out) if math.isfinite(out) else 0.0

    # ---- CTL/acceleration_controller.hpp ----
    def throttle_of_force(self, fd):  # :91-94, :116-119
        p = self.p
        thrust_force = float(fd @ self.R[:, 2])
        with np.errstate(invalid="ignore"):
            return float((np.sqrt(thrust_force / (p["kf"] * p["n_motors"])) - p["min_rpm"]) / (p["max_rpm"] - p["min_rpm"]))

    def acceleration_to_attitude(self, acc, heading):  # :44-97
        p = self.p
        fd = (acc + np.array([0.0, 0.0, p["g"]])) * p["mass"]
        z = _normalized(fd)
        bxd = np.array([math.cos(heading), math.sin(heading), 0.0])
        proj = np.eye(3) - np.outer(z, z)
        A = proj[:, 0:2]
        B = np.array([[1.0, 0.0], [0.0, 1.0], [0.0, 0.0]])
        BtA = B.T @ A
        with np.errstate(all="ignore"):
            try:
                pinv = np.linalg.inv(BtA.T @ BtA) @ BtA.T
            except np.linalg.LinAlgError:
                pinv = np.full((2, 2), np.nan)
            xb = _normalized(A @ pinv @ B.T @ bxd)
            yb = _normalized(np.cross(z, xb))
        Rd = np.column_stack([xb, yb, z])
        return Rd, self.throttle_of_force(fd)

    # ---- US:304-380 ----
    def make_step(self, dt):
        n = self.p["n_motors"]
        mode = self.active_input
        if self.crashed or mode == INPUT_UNKNOWN:
            self.actuators = np.zeros(n)
        else:
            c, ff = self.cmd, self.ff
            if mode == POSITION_CMD:
                ref = c[POSITION_CMD]
                err = ref[0:3] - self.x
                vel = np.array([self.pid_pos[i].update(err[i], dt) for i in range(3)])
                if FF_VELOCITY_HDG in ff:
                    vel = vel + ff[FF_VELOCITY_HDG][0:3]
                elif FF_VELOCITY_HDG_RATE in ff:
                    vel = vel + ff[FF_VELOCITY_HDG_RATE][0:3]
                c[VELOCITY_HDG_CMD] = np.concatenate([vel, [ref[3]]])
                mode = VELOCITY_HDG_CMD
            if mode == VELOCITY_HDG_CMD:
                ref = c[VELOCITY_HDG_CMD]
                err = ref[0:3] - self.v
                acc = np.array([self.pid_vel[i].update(err[i], dt) for i in range(3)])
                if FF_ACCELERATION_HDG in ff:
                    acc = acc + ff[FF_ACCELERATION_HDG][0:3]
                elif FF_ACCELERATION_HDG_RATE in ff:
                    acc = acc + ff[FF_ACCELERATION_HDG_RATE][0:3]
                c[ACCELERATION_HDG_CMD] = np.concatenate([acc, [ref[3]]])
                mode = ACCELERATION_HDG_CMD
            elif mode == VELOCITY_HDG_RATE_CMD:
                ref = c[VELOCITY_HDG_RATE_CMD]
                err = ref[0:3] - self.v
                acc = np.array([self.pid_vel[i].update(err[i], dt) for i in range(3)])
                rate = ref[3]
                if FF_ACCELERATION_HDG_RATE in ff:
                    acc = acc + ff[FF_ACCELERATION_HDG_RATE][0:3]
                    rate = rate + ff[FF_ACCELERATION_HDG_RATE][3]
                elif FF_ACCELERATION_HDG in ff:
                    acc = acc + ff[FF_ACCELERATION_HDG][0:3]
                c[ACCELERATION_HDG_RATE_CMD] = np.concatenate([acc, [rate]])
                mode = ACCELERATION_HDG_RATE_CMD
            if mode == ACCELERATION_HDG_CMD:
                ref = c[ACCELERATION_HDG_CMD]
                Rd, throttle = self.acceleration_to_attitude(ref[0:3], ref[3])
                c[ATTITUDE_CMD] = np.concatenate([Rd.reshape(9), [throttle]])
                mode = ATTITUDE_CMD
            elif mode == ACCELERATION_HDG_RATE_CMD:  # CTL/acceleration_controller.hpp:103-122
                ref = c[ACCELERATION_HDG_RATE_CMD]
                fd = (ref[0:3] + np.array([0.0, 0.0, self.p["g"]])) * self.p["mass"]
                c[TILT_HDG_RATE_CMD] = np.concatenate([_normalized(fd), [ref[3], self.throttle_of_force(fd)]])
                mode = TILT_HDG_RATE_CMD
            if mode == ATTITUDE_CMD:  # CTL/attitude_controller.hpp:79-100
                ref = c[ATTITUDE_CMD]
                rates = self.attitude_error_rates(ref[0:9].reshape(3, 3), dt)
                c[ATTITUDE_RATE_CMD] = np.array(rates + [ref[9]])
                mode = ATTITUDE_RATE_CMD
            elif mode == TILT_HDG_RATE_CMD:  # :106-145
                ref = c[TILT_HDG_RATE_CMD]
                z = _normalized(ref[0:3])
                y = _normalized(np.cross(z, self.R[:, 0]))
                x = _normalized(np.cross(y, z))
                rates = self.attitude_error_rates(np.column_stack([x, y, z]), dt)
                parasitic = self.heading_rate_of_body_rate(rates)
                rates[2] += self.yaw_rate_intrinsic(ref[3] - parasitic)
                c[ATTITUDE_RATE_CMD] = np.array(rates + [ref[4]])
                mode = ATTITUDE_RATE_CMD
            if mode == ATTITUDE_RATE_CMD:  # CTL/rate_controller.hpp:67-81
                ref = c[ATTITUDE_RATE_CMD]
                err = ref[0:3] - self.omega
                c[CONTROL_GROUP_CMD] = np.array([self.pid_rate[i].update(err[i], dt) for i in range(3)] + [ref[3]])
                mode = CONTROL_GROUP_CMD
            if mode == CONTROL_GROUP_CMD:
                self.actuators = self.mixer(c[CONTROL_GROUP_CMD])
                mode = ACTUATOR_CMD
            elif mode == ACTUATOR_CMD:
                self.actuators = np.array(c[ACTUATOR_CMD][:n], dtype=float)
        # MM:392-410 setInput
        val = np.where(np.isfinite(self.actuators), self.actuators, 0.0)
        val = np.clip(val, 0.0, 1.0)
        self.input = self.p["min_rpm"] + (self.p["max_rpm"] - self.p["min_rpm"]) * val
        self.model_step(dt)

    # ---- MM:301-366 ----
    def rhs(self, y):
        p = self.p
        v = y[3:6]
        Rraw = np.column_stack([y[6:9], y[9:12], y[12:15]])
        w = y[15:18]
        with np.errstate(all="ignore"):
            R = _re_orthonormalised(Rraw)
            W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            torque_thrust = np.asarray(p["allocation_matrix"], dtype=float) @ (self.motor_rpm ** 2)  # member rpm: constant over the stages
            speed = np.linalg.norm(v)
            resistance = p["air_resistance_coeff"] * math.pi * p["arm_length"] * p["arm_length"] * speed * speed
            vhat = v / speed if speed != 0 else v
            J = np.asarray(p["J"], dtype=float).reshape(3, 3)
            v_dot = (-np.array([0.0, 0.0, p["g"]]) + torque_thrust[3] * R[:, 2] / p["mass"] + self.external_force / p["mass"]
                     - resistance * vhat / p["mass"])
            R_dot = R @ W
            w_dot = np.linalg.inv(J) @ (torque_thrust[0:3] - np.cross(w, J @ w) + self.external_moment)
        d = np.concatenate([v, v_dot, R_dot[:, 0], R_dot[:, 1], R_dot[:, 2], w_dot])
        d[np.isnan(d)] = 0.0  # :361-365
        return d

    # ---- MM:220-286 ----
    def model_step(self, dt):
        p = self.p
        y0 = np.concatenate([self.x, self.v, self.R[:, 0], self.R[:, 1], self.R[:, 2], self.omega])
        # classical RK4, one step (ODE/stepper/runge_kutta4.hpp:43-95)
        k1 = self.rhs(y0)
        k2 = self.rhs(y0 + 0.5 * dt * k1)
        k3 = self.rhs(y0 + 0.5 * dt * k2)
        k4 = self.rhs(y0 + dt * k3)
        y = y0 + dt / 6.0 * k1 + dt / 3.0 * k2 + dt / 3.0 * k3 + dt / 6.0 * k4
        if np.isnan(y).any():  # :228-233
            y = y0
        self.x, self.v, self.omega = y[0:3].copy(), y[3:6].copy(), y[15:18].copy()
        self.R = np.column_stack([y[6:9], y[9:12], y[12:15]])
        c = math.exp(-dt / p["motor_time_constant"])
        self.motor_rpm = c * self.motor_rpm + (1.0 - c) * self.input
        with np.errstate(all="ignore"):
            self.R = _re_orthonormalised(self.R)
        if p["ground_enabled"] and self.x[2] < p["ground_z"] and self.v[2] < 0:
            self.x[2] = p["ground_z"]
            self.v = np.zeros(3)
            self.omega = np.zeros(3)
        if self.takeoff_patch_enabled:
            hover_rpm = math.sqrt((p["mass"] * p["g"]) / (p["n_motors"] * p["kf"]))
            if self.input.mean() <= 0.90 * hover_rpm:
                if self.x[2] < self.initial_pos[2] and self.v[2] < 0:
                    self.x[2] = self.initial_pos[2]
                    self.v = np.zeros(3)
                    self.omega = np.zeros(3)
            else:
                self.takeoff_patch_enabled = False
        self.imu = self.R.T @ ((self.v - self.v_prev) / dt + np.array([0.0, 0.0, p["g"]]))
        self.v_prev = self.v.copy()


def params_from_struct(ps):
    """dict of model parameters from an oracle/product ModelParams ctypes struct (allocation rows are MAX_MOTORS wide)."""
    n = int(ps.n_motors)
    width = len(ps.allocation_matrix) // 4
    A = np.array(ps.allocation_matrix, dtype=float).reshape(4, width)[:, :n]
    d = {k: float(getattr(ps, k)) for k in ("g", "mass", "kf", "km", "prop_radius", "arm_length", "body_height", "motor_time_constant",
                                           "max_rpm", "min_rpm", "air_resistance_coeff", "ground_z")}
    d.update(n_motors=n, ground_enabled=bool(ps.ground_enabled), takeoff_patch_enabled=bool(ps.takeoff_patch_enabled),
             J=np.array(ps.J, dtype=float).reshape(3, 3), allocation_matrix=A)
    return d
