"""Deterministic parameter values shared by the fixture generator and the tests (seeded numpy PCG64, so the
multi-megabyte hash tables never need to be stored in a fixture).  Magnitudes are larger than a fresh tcnn
initialisation on purpose: the resulting fields have structure (densities spanning ~0.1 .. 10)."""
import numpy as np


def grid_params(n, seed, std=0.1):
    return (np.random.default_rng(1000 + seed).standard_normal(n) * std).astype(np.float32)


def mlp_params(shapes, seed, gain=1.5):
    rng = np.random.default_rng(5000 + seed)
    return np.concatenate([(rng.uniform(-1.0, 1.0, a * b) * gain * np.sqrt(6.0 / (a + b))).astype(np.float32) for a, b in shapes])


def plane_params(shape, seed, time_plane):
    rng = np.random.default_rng(9000 + seed)
    if time_plane:
        return (1.0 + 0.2 * rng.standard_normal(shape)).astype(np.float32)
    return rng.uniform(0.1, 0.5, shape).astype(np.float32)
