"""Synthetic swarm states for benchmarks and tests (BASELINE configs 3-5): pure numpy, no dependency on the product library or on
the test oracle."""
import numpy as np


def random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - z * w)
    R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w)
    R[:, 2, 1] = 2 * (y * z + x * w)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def tilted_rotations(rng, n, max_tilt=0.5):
    """rotations with a bounded tilt from vertical and a free heading (keeps the cascade in its working range)."""
    axis = rng.normal(size=(n, 3))
    axis[:, 2] = 0
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = rng.uniform(0, max_tilt, n)
    K = np.zeros((n, 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -axis[:, 2], axis[:, 1]
    K[:, 1, 0], K[:, 1, 2] = axis[:, 2], -axis[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -axis[:, 1], axis[:, 0]
    Rt = np.eye(3)[None] + np.sin(ang)[:, None, None] * K + (1 - np.cos(ang))[:, None, None] * (K @ K)
    h = rng.uniform(-np.pi, np.pi, n)
    Rz = np.zeros((n, 3, 3))
    Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = np.cos(h), -np.sin(h), np.sin(h), np.cos(h), 1
    return Rt @ Rz


def random_state(rng, n, n_motors, box=500.0, zlo=5.0, zhi=100.0, tilted=False):
    """BASELINE config 3 generator: x~U(-box,box)^2 x U(zlo,zhi), v~N(0,1), R random, omega~N(0,.5), rpm~U(3000,5000)."""
    x = np.stack([rng.uniform(-box, box, n), rng.uniform(-box, box, n), rng.uniform(zlo, zhi, n)], axis=1)
    v = rng.normal(0, 1, (n, 3))
    R = tilted_rotations(rng, n) if tilted else random_rotations(rng, n)
    omega = rng.normal(0, 0.5, (n, 3))
    rpm = np.zeros((n, 8))
    rpm[:, :n_motors] = rng.uniform(3000, 5000, (n, n_motors))
    return dict(x=x, v=v, R=R, omega=omega, motor_rpm=rpm)
