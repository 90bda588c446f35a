"""fhe.rs_amd -- MI355X-native engine for the fhe.rs BFV hot path (NTT, RNS scaler, key
switch, ct x ct + relinearise, rotation, modulus switch) behind the C ABI in include/fhe_hip.h.

The directory is literally called `fhe.rs_amd`, which Python cannot import by statement;
`import fhe_rs_amd` (repo root shim) loads it under that importable name.
"""
from . import _lib  # noqa: F401
from .api import *  # noqa: F401,F403
from .api import (Context, Scaler, Switcher, KeySwitchingKey, RelinearizationKey, GaloisKey, EvaluationKey, RGSWCiphertext,
                  BfvParameters, Multiplicator, FheError, Stream, DeviceArray)  # noqa: F401
