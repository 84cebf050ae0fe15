"""Deterministic tensor fill shared by the golden generator and the tests.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Every tensor is a pure function of (name, shape, seed): a splitmix64 counter
hash -> uniform double in [0,1) -> affine map chosen by the *kind* of the
state_dict entry.  No dependence on torch / numpy RNG streams, so the same
weights can be poured into the reference modules (in the build container),
into the oracle and into the HIP-backed modules (on the GPU box) without
shipping 31 MB checkpoints.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def u01(name, n, seed=0):
    """n uniform doubles in [0,1), a pure function of (name, seed, index)."""
    key = (zlib.crc32(name.encode()) & 0xFFFFFFFF) * 0x100000001B3 + int(seed) * 0xD1B54A32D192ED03
    key = np.uint64(key & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = _splitmix64(_splitmix64(idx + key) ^ key)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(name, shape, lo, hi, seed=0):
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * u01(name, n, seed)).reshape(shape).astype(np.float32)


def randint(name, shape, high, seed=0):
    n = int(np.prod(shape)) if len(shape) else 1
    return np.minimum((u01(name, n, seed) * high).astype(np.int64), high - 1).reshape(shape)


def fill_entry(name, shape, seed=0):
    """Value for one state_dict entry, chosen by the entry's role.

    conv / convT / linear weights : U(-b, b), b = sqrt(3 / fan_in)  (unit-gain)
    biases                        : U(-0.1, 0.1)
    BN weight                     : U(0.5, 1.5)
    BN running_mean               : U(-0.2, 0.2)
    BN running_var                : U(0.5, 1.5)
    num_batches_tracked           : 0
    """
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return uniform(name, shape, -0.2, 0.2, seed)
    if leaf == "running_var":
        return uniform(name, shape, 0.5, 1.5, seed)
    if leaf in ("bias", "in_proj_bias"):
        return uniform(name, shape, -0.1, 0.1, seed)
    if len(shape) == 1:  # BN affine weight
        return uniform(name, shape, 0.5, 1.5, seed)
    if len(shape) == 4:
        if ".up." in name or name.startswith("up."):  # ConvTranspose2d (Cin, Cout, 2, 2)
            fan_in = shape[0]
        else:  # Conv2d (Cout, Cin, kh, kw)
            fan_in = shape[1] * shape[2] * shape[3]
    else:  # linear-like (out, in)
        fan_in = shape[-1]
    b = float(np.sqrt(3.0 / fan_in))
    return uniform(name, shape, -b, b, seed)


def fill_state_dict(sd, seed=0, prefix=""):
    """In-place pour of deterministic values into a torch state_dict-like mapping."""
    import torch

    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(torch.from_numpy(fill_entry(prefix + k, tuple(v.shape), seed)).to(v.dtype))
    return sd
