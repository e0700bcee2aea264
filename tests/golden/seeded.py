"""Seeded tensors shared by the fixture generator (make_golden.py, which loads them into the REAL reference modules) and
the tests (which load the same values into the product modules / the oracle): weights of the PatchTST-size layer cases
are regenerated from a seed instead of being stored (12 MB per encoder layer)."""
import numpy as np


def rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def state_like(shapes, seed):
    """{name: array}: 2-D+ tensors ~ N(0, 1/fan_in), LayerNorm weights 1 + 0.1 N(0,1), biases 0.1 N(0,1)"""
    out = {}
    for i, (name, shp) in enumerate(sorted(shapes.items())):
        shp = tuple(shp)
        if len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            out[name] = rand(shp, seed + 7 * i, 1.0 / np.sqrt(fan_in))
        elif "norm" in name and name.endswith("weight"):
            out[name] = 1.0 + rand(shp, seed + 7 * i, 0.1)
        else:
            out[name] = rand(shp, seed + 7 * i, 0.1)
    return out


def probes(arr, seed, n=4):
    """n seeded random projections + the norm of a tensor: a size-independent fingerprint of a gradient"""
    a = np.asarray(arr, np.float64).reshape(-1)
    rng = np.random.default_rng(seed)
    return np.array([float(np.dot(a, rng.standard_normal(a.shape[0]))) for _ in range(n)] + [float(np.linalg.norm(a))])
