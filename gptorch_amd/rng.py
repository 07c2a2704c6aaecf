"""
Deterministic, counter-based synthetic-data generator (splitmix64 -> Box-Muller).

Used by bench.py, the golden-fixture generator and the parity tests so that the
GPU box regenerates the *same* X, Y from (seed, N, D) without shipping arrays.
Implements the workload of SURVEY.md section 8(d): X ~ N(0,1)^{N x D},
y = sin(sum_d x_d) + 0.1*eps.

Pure numpy (uint64 arithmetic wraps modulo 2^64 by construction).
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n outputs of the splitmix64 stream started at `seed` (counter form)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """Uniform doubles in (0, 1): 53 high bits, centred so 0 never occurs."""
    z = splitmix64(seed, n, offset)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, shape) -> np.ndarray:
    """Standard normals, element i from uniforms (2i, 2i+1) (cosine branch)."""
    n = int(np.prod(shape))
    u = uniform(seed, 2 * n)
    u1, u2 = u[0::2], u[1::2]
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.reshape(shape)


def make_regression(n: int, d: int, dy: int = 1, seed: int = 0):
    """X ~ N(0,1)^{n x d} (seed), Y[:, j] = sin(sum_d x_d + j) + 0.1 eps (seed+1)."""
    x = normal(seed, (n, d))
    eps = normal(seed + 1, (n, dy))
    s = x.sum(axis=1, keepdims=True) + np.arange(dy, dtype=np.float64)[None, :]
    y = np.sin(s) + 0.1 * eps
    return x, y


def checksum(a: np.ndarray) -> str:
    """Order-sensitive 64-bit checksum of the raw bytes (hex), for fixtures."""
    v = np.ascontiguousarray(a).view(np.uint64).ravel()
    with np.errstate(over="ignore"):
        w = splitmix64(0x1234, v.size)
        return format(int(np.bitwise_xor.reduce(v * (w | np.uint64(1)) + w)), "016x")
