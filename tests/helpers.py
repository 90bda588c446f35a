"""Shared helpers for the parity tests: build oracle objects and the matching engine
objects through the C ABI, convert between oracle Polys and `[L][N]` uint64 arrays."""
import os
import random
import subprocess

import numpy as np

from fhe_oracle import bfv as obfv
from fhe_oracle import coracle
from fhe_oracle.rns import ScalingFactor
from fhe_oracle.rq import Context as OCtx, Poly, Scaler as OScaler, POWER_BASIS, NTT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "libfhe_emu.so")
HIP_LIB = os.path.join(ROOT, "fhe.rs_amd", "libfhe_hip.so")


def build_emu():
    # tools/sanitize_emu.sh points this at an -fsanitize build of the same sources
    if os.environ.get("FHE_EMU_LIB"):
        return os.environ["FHE_EMU_LIB"]
    srcs = [os.path.join(ROOT, "fhe.rs_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "fhe.rs_amd", "csrc"))]
    srcs += [os.path.join(ROOT, "tests", "emu", "emu_rt.hpp"), os.path.join(ROOT, "include", "fhe_hip.h")]
    def stale():
        return (not os.path.exists(EMU_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(EMU_LIB) for s in srcs)
    if stale():
        # one builder at a time (pytest-xdist workers of a fresh checkout all arrive here at once); build.sh renames
        # the finished library into place, so a worker that did not build never sees a partial file
        import fcntl
        os.makedirs(os.path.dirname(EMU_LIB), exist_ok=True)
        with open(os.path.join(os.path.dirname(EMU_LIB), ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if stale():
                subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build.sh")], stdout=subprocess.DEVNULL)
    return EMU_LIB


def load_engine(kind):
    """kind: 'emu' (host emulation of the kernel sources, CPU CI) or 'hip' (the product)."""
    import fhe_rs_amd
    from fhe_rs_amd import _lib
    want = build_emu() if kind == "emu" else HIP_LIB
    if _lib.loaded_path() != want:
        if kind == "emu":
            _lib._load_for_tests(want)
        else:
            _lib._lib = None
            _lib.lib()
    return fhe_rs_amd


def arr(poly_or_rows):
    rows = poly_or_rows.coefficients if isinstance(poly_or_rows, Poly) else poly_or_rows
    return np.array(rows, dtype=np.uint64)


def rows(a):
    return [[int(x) for x in r] for r in np.asarray(a)]


def rand_poly(ctx, rep, rng):
    return obfv.random_poly(ctx, rep, rng)


def oracle_tables(octx):
    return dict(omegas=[op.omegas for op in octx.ops], omegas_shoup=[op.omegas_shoup for op in octx.ops],
                zetas_inv=[op.zetas_inv for op in octx.ops], zetas_inv_shoup=[op.zetas_inv_shoup for op in octx.ops],
                size_inv=[op.size_inv for op in octx.ops], size_inv_shoup=[op.size_inv_shoup for op in octx.ops])


def ksk_arrays(oksk):
    """Oracle KeySwitchingKey -> (c0, c0_shoup, c1, c1_shoup) arrays [ndigits][Lk][N]."""
    return (np.array([p.coefficients for p in oksk.c0], dtype=np.uint64),
            np.array([p.coefficients_shoup for p in oksk.c0], dtype=np.uint64),
            np.array([p.coefficients for p in oksk.c1], dtype=np.uint64),
            np.array([p.coefficients_shoup for p in oksk.c1], dtype=np.uint64))


def ct_arr(ct):
    return np.array([p.coefficients for p in ct.c], dtype=np.uint64)


class Xfer:
    """Moves arrays to the place the engine call should see them: numpy (host-pointer API, dev = False), torch CUDA
    tensors (device-pointer `_dev` API, dev = True) or `DeviceArray`s allocated through the C ABI itself (dev = "abi":
    the `_dev` API as a host without PyTorch uses it; the caller makes a `Stream` current)."""

    def __init__(self, dev):
        self.dev = dev
        if dev == "abi":
            import fhe_rs_amd
            self.DeviceArray = fhe_rs_amd.DeviceArray
        elif dev:
            import torch
            self.torch = torch

    def to(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
        if not self.dev:
            return a
        if self.dev == "abi":
            return self.DeviceArray.from_numpy(a)
        return self.torch.from_numpy(a.view(np.int64)).cuda()

    def to_bytes(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.uint8))
        if not self.dev:
            return a
        if self.dev == "abi":
            return self.DeviceArray.from_numpy(a)
        return self.torch.from_numpy(a.copy()).cuda()

    def back_bytes(self, x):
        if not self.dev:
            return np.asarray(x)
        return x.download() if self.dev == "abi" else x.cpu().numpy()

    def back(self, x):
        if not self.dev:
            return np.asarray(x)
        if self.dev == "abi":
            return x.download()
        return x.cpu().numpy().view(np.uint64)
