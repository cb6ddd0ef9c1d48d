"""Inputs of golden vectors G13 / G14 that are formulas rather than stored arrays (shared by make_golden.py and the GPU tests):
initial guess and support mask of the (64, 64, 64) reconstruct_fullfield run, initial guess of the (64, 64, 64)
reconstruct_ptychography run."""
import numpy as np

SHAPE = (64, 64, 64)


def initial_guess(shape=SHAPE):
    y, x, z = np.mgrid[:shape[0], :shape[1], :shape[2]].astype(np.float64)
    w = np.sin(0.37 * y + 0.91 * x + 1.7 * z) * np.cos(0.53 * x - 0.29 * z + 0.11 * y)
    d = (8.7e-7 * (1 + 0.2 * w)).astype(np.float32)
    b = (5.1e-8 * (1 - 0.2 * w)).astype(np.float32)
    return d.astype(np.float64), b.astype(np.float64)          # exactly what a float32 volume holds


def mask():
    m = np.ones(SHAPE, dtype=np.float32)
    m[:3] = 0
    return m
