"""Host-side mirror of the reference types that sit on the hot path, as thin ctypes wrappers
over the C ABI (include/fhe_hip.h).  Names, argument meaning and error behaviour follow fhe.rs:

    Context, Scaler, Switcher                  fhe_math::rq            (crates/fhe-math/src/rq/)
    KeySwitchingKey, RelinearizationKey,
    GaloisKey, EvaluationKey, Multiplicator,
    BfvParameters                              fhe::bfv                (crates/fhe/src/bfv/)

Every compute method accepts either
  * a numpy uint64 array  -> host-pointer entry point (synchronous, returns a new array),
  * a torch CUDA tensor   -> `_dev` entry point on torch's current stream (device pointers;
    torch is used purely as the device allocator / stream provider), or
  * a `DeviceArray`       -> `_dev` entry point on the current `Stream`: buffers and streams that come from the C
    ABI itself (fhe_buf_alloc / fhe_stream_create), i.e. what a host without PyTorch -- the Rust shim, a C program --
    uses; results are DeviceArrays that stay resident until `.download()`.
Polynomials are `[..., L, N]` row-major exactly like `rq::Poly`; leading dims are the batch.
Nothing in this module computes on the CPU: without the HIP library it cannot be imported.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, FheError  # noqa: F401  (re-exported)

try:  # torch is only the device allocator / stream provider
    import torch
except Exception:  # pragma: no cover
    torch = None


import threading

_tls = threading.local()


class Stream:
    """A HIP stream handed out by the C ABI (fhe_stream_create).  `with Stream(dev) as s:` makes it the stream every
    `_dev` call of this thread runs on (instead of torch's current stream)."""

    def __init__(self, device=0, _foreign=None):
        if _foreign is not None:     # a hipStream_t the host made itself: used, never destroyed here
            self._h, self.device, self._owned = C.c_void_p(_foreign), device, False
            return
        h = C.c_void_p()
        check(_lib.lib().fhe_stream_create(device, C.byref(h)))
        self._h, self.device, self._owned = h, device, True

    @classmethod
    def foreign(cls, handle, device=0):
        """Wraps a HIP stream that the host created (and will destroy) itself -- what a torch / hip-rs host passes."""
        return cls(device, _foreign=int(handle))

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        check(_lib.lib().fhe_stream_sync(self._h))

    def destroy(self):
        if self._h is not None and self._owned:
            check(_lib.lib().fhe_stream_destroy(self._h))
        self._h = None

    def __enter__(self):
        self._prev = getattr(_tls, "stream", None)
        _tls.stream = self
        return self

    def __exit__(self, *exc):
        _tls.stream = self._prev
        return False

    def __del__(self):
        if getattr(self, "_h", None) is not None and getattr(self, "_owned", False) and _lib._lib is not None:
            _lib._lib.fhe_stream_destroy(self._h)


class DeviceArray:
    """A device buffer owned through the C ABI (fhe_buf_alloc, or fhe_buf_alloc_async inside `with Stream`), with a shape: the device-resident shadow of a
    `[..., L, N]` coefficient array (`rq::Poly.coefficients`) for hosts that do not bring their own HIP allocator."""

    def __init__(self, shape, device=0, itemsize=8, _ptr=None, _base=None):
        self.shape = tuple(int(d) for d in shape)
        self.device, self.itemsize = device, itemsize
        self._base = _base
        self._astream = None
        self._other_stream_use = False    # (set by _note_use: some call used this array on another stream)
        if _ptr is None:
            h = C.c_void_p()
            st = getattr(_tls, "stream", None)
            if st is not None and st.device == device and st.handle is not None:
                # inside `with Stream(...)`: stream-ordered allocation (no device synchronisation when results are
                # allocated and dropped per call); the array remembers its stream for the matching free
                check(_lib.lib().fhe_buf_alloc_async(device, max(self.nbytes, 1), st.handle, C.byref(h)))
                self._astream = st
            else:
                check(_lib.lib().fhe_buf_alloc(device, max(self.nbytes, 1), C.byref(h)))
            self._p = h.value
        else:
            self._p = _ptr

    @classmethod
    def from_numpy(cls, a, device=0):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, device, a.dtype.itemsize)
        check(_lib.lib().fhe_buf_upload(C.c_void_p(d._p), a.ctypes.data_as(C.c_void_p), a.nbytes, _stream()))
        return d

    @property
    def nbytes(self):
        n = self.itemsize
        for d in self.shape:
            n *= d
        return n

    # the few members the wrappers below use on torch tensors
    is_cuda = True

    def is_contiguous(self):
        return True

    def contiguous(self):
        return self

    def element_size(self):
        return self.itemsize

    def data_ptr(self):
        return self._p

    def numel(self):
        return self.nbytes // self.itemsize

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, i):
        """View of the i-th slice along the first dimension (no copy; keeps the parent alive)."""
        if not isinstance(i, int) or not self.shape:
            raise TypeError("DeviceArray supports integer indexing of the first dimension only")
        if i < 0:
            i += self.shape[0]
        if not 0 <= i < self.shape[0]:
            raise IndexError(i)
        sub = self.nbytes // self.shape[0]
        return DeviceArray(self.shape[1:], self.device, self.itemsize, _ptr=self._p + i * sub, _base=self if self._base is None else self._base)

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        v = DeviceArray(shape, self.device, self.itemsize, _ptr=self._p, _base=self if self._base is None else self._base)
        if v.nbytes != self.nbytes:
            raise ValueError("reshape changes the size")
        return v

    def download(self):
        """Waits for the current stream and returns the contents as a numpy array (uint64 / uint8)."""
        out = np.empty(self.shape, dtype=np.uint64 if self.itemsize == 8 else np.uint8)
        check(_lib.lib().fhe_buf_download(out.ctypes.data_as(C.c_void_p), C.c_void_p(self._p), self.nbytes, _stream()))
        return out

    def _note_use(self):
        """Called for every engine call that takes this array: remembers whether any of them ran on a stream other
        than the allocating one (an array used inside another `with Stream` block and dropped after
        returning to its own stream must not be freed in the allocating stream's order)."""
        root = self if self._base is None else self._base   # (not `or`: __len__ makes an empty base falsy)
        if root._astream is not None and getattr(_tls, "stream", None) is not root._astream:
            root._other_stream_use = True

    def _release(self, L):
        st = getattr(self, "_astream", None)
        # stream-ordered free only when EVERY use of the array was enqueued on the allocating stream and that stream is
        # still the current one.  Otherwise the synchronous free, which waits for the device (ADVICE r03 / r04:
        # hipFreeAsync on the allocating stream could hand the block back while another stream still read it).
        if (st is not None and st.handle is not None and getattr(_tls, "stream", None) is st
                and not getattr(self, "_other_stream_use", False)):
            return L.fhe_buf_free_async(C.c_void_p(self._p), st.handle)
        return L.fhe_buf_free(C.c_void_p(self._p))                        # (hipFree: waits for the device)

    def free(self):
        if self._base is None and self._p is not None:
            check(self._release(_lib.lib()))
        self._p = None

    def __del__(self):
        if getattr(self, "_base", None) is None and getattr(self, "_p", None) is not None and _lib._lib is not None:
            self._release(_lib._lib)


def _is_dev(x):
    return isinstance(x, DeviceArray) or (torch is not None and isinstance(x, torch.Tensor))


def _empty_dev(ref, shape, itemsize=8, zero=False):
    """An uninitialised (or zeroed) device array of `shape` next to `ref` (same allocator, same device)."""
    if isinstance(ref, DeviceArray):
        d = DeviceArray(shape, ref.device, itemsize)
        if zero:
            check(_lib.lib().fhe_buf_zero_async(C.c_void_p(d._p), d.nbytes, _stream()))
        return d
    dt = torch.int64 if itemsize == 8 else torch.uint8
    return (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dt, device=ref.device)


def _np(x):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64))
    return a


def _ptr(a):
    return a.ctypes.data_as(_lib.u64p)


def _dptr(t):
    if not t.is_cuda or not t.is_contiguous() or t.element_size() != 8:
        raise ValueError("device buffers must be contiguous 8-byte CUDA tensors")
    if isinstance(t, DeviceArray):
        t._note_use()
    return C.c_void_p(t.data_ptr())


def _dptr8(t):
    if not t.is_cuda or not t.is_contiguous() or t.element_size() != 1:
        raise ValueError("wire buffers must be contiguous uint8 CUDA tensors")
    if isinstance(t, DeviceArray):
        t._note_use()
    return C.c_void_p(t.data_ptr())


def _stream():
    s = getattr(_tls, "stream", None)
    if s is not None:
        return s.handle
    if torch is not None and torch.cuda.is_available():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None   # the device's null stream


def _limbs(x: int):
    out = []
    while True:
        out.append(x & 0xFFFFFFFFFFFFFFFF)
        x >>= 64
        if x == 0:
            break
    return np.array(out, dtype=np.uint64)


class Context:
    """rq::Context (crates/fhe-math/src/rq/context.rs:9-92)."""

    def __init__(self, moduli, degree, device=0, tables=None, _handle=None, _owner=None):
        self._owner = _owner
        if _handle is not None:
            self._h = _handle
        else:
            m = _np(moduli)
            h = C.c_void_p()
            if tables is None:
                args = [None] * 6
                keep = []
            else:
                keep = [_np(tables[k]) for k in ("omegas", "omegas_shoup", "zetas_inv", "zetas_inv_shoup",
                                                 "size_inv", "size_inv_shoup")]
                args = [_ptr(a) for a in keep]
            check(_lib.lib().fhe_ctx_create(device, degree, len(m), _ptr(m), *args, C.byref(h)))
            self._h = h
        L = _lib.lib()
        self.degree = L.fhe_ctx_degree(self._h)
        self.nmoduli = L.fhe_ctx_nmoduli(self._h)
        self.device = L.fhe_ctx_device(self._h)
        mm = np.zeros(self.nmoduli, dtype=np.uint64)
        check(L.fhe_ctx_moduli(self._h, _ptr(mm)))
        self.moduli = [int(x) for x in mm]

    def __del__(self):
        if self._owner is None and getattr(self, "_h", None) is not None and _lib._lib is not None:
            _lib._lib.fhe_ctx_destroy(self._h)

    def at_level(self, level):
        """Context::context_at_level."""
        h = C.c_void_p()
        check(_lib.lib().fhe_ctx_at_level(self._h, level, C.byref(h)))
        return Context(None, None, _handle=h, _owner=self._owner or self)

    def niterations_to(self, other):
        n = C.c_size_t()
        check(_lib.lib().fhe_ctx_niterations_to(self._h, other._h, C.byref(n)))
        return n.value

    def table(self, which):
        shape = (self.nmoduli, self.degree) if which < 4 else ((self.nmoduli,) if which < 6 else (max(self.nmoduli - 1, 0),))
        out = np.zeros(max(int(np.prod(shape)), 1), dtype=np.uint64)
        check(_lib.lib().fhe_ctx_get_table(self._h, which, _ptr(out)))
        return out[: int(np.prod(shape))].reshape(shape)

    # -- helpers ------------------------------------------------------------------
    def _batch(self, x, rows=None):
        rows = self.nmoduli if rows is None else rows
        shp = tuple(x.shape)
        if len(shp) < 2 or shp[-1] != self.degree or shp[-2] != rows:
            raise ValueError(f"expected [..., {rows}, {self.degree}], got {shp}")
        b = 1
        for d in shp[:-2]:
            b *= d
        return b

    def _unary_inplace(self, host_fn, dev_fn, x):
        L = _lib.lib()
        if _is_dev(x):
            check(getattr(L, dev_fn)(self._h, _dptr(x), self._batch(x), _stream()))
            return x
        a = _np(x).copy()
        check(getattr(L, host_fn)(self._h, _ptr(a), self._batch(a)))
        return a

    def _binary_inplace(self, host_fn, dev_fn, x, y):
        L = _lib.lib()
        if _is_dev(x):
            check(getattr(L, dev_fn)(self._h, _dptr(x), _dptr(y), self._batch(x), _stream()))
            return x
        a, b = _np(x).copy(), _np(y)
        if a.shape != b.shape:
            raise ValueError("shape mismatch")
        check(getattr(L, host_fn)(self._h, _ptr(a), _ptr(b), self._batch(a)))
        return a

    # -- NttOperator / Poly representation changes (rq/mod.rs:335-354) ---------------
    def ntt_forward(self, polys):
        return self._unary_inplace("fhe_ntt_forward", "fhe_ntt_forward_dev", polys)

    def ntt_backward(self, polys):
        return self._unary_inplace("fhe_ntt_backward", "fhe_ntt_backward_dev", polys)

    # -- Rq wire format (rq/convert.rs:17-99, zq/mod.rs:783-793) ------------------------------
    @property
    def serialized_size(self):
        return int(_lib.lib().fhe_poly_serialized_size(self._h))

    def serialize(self, polys, from_ntt=False):
        """[..., L, N] -> uint8 [..., serialized_size]: the `coefficients` payload of the Rq message."""
        L = _lib.lib()
        b = self._batch(polys)
        oshape = tuple(polys.shape[:-2]) + (self.serialized_size,)
        if _is_dev(polys):
            out = _empty_dev(polys, oshape, 1)
            check(L.fhe_poly_serialize_dev(self._h, _dptr(polys), _dptr8(out), b, 1 if from_ntt else 0, _stream()))
            return out
        x = _np(polys)
        out = np.zeros(oshape, dtype=np.uint8)
        check(L.fhe_poly_serialize(self._h, _ptr(x), out.ctypes.data_as(_lib.u8p), b, 1 if from_ntt else 0))
        return out

    def deserialize(self, data, to_ntt=False):
        """uint8 [..., serialized_size] -> [..., L, N]."""
        L = _lib.lib()
        if data.shape[-1] != self.serialized_size:
            raise FheError(-1, "InvalidCoefficientCount")
        oshape = tuple(data.shape[:-1]) + (self.nmoduli, self.degree)
        b = 1
        for d in data.shape[:-1]:
            b *= int(d)
        if _is_dev(data):
            out = _empty_dev(data, oshape)
            check(L.fhe_poly_deserialize_dev(self._h, _dptr8(data.contiguous()), _dptr(out), b, 1 if to_ntt else 0, _stream()))
            return out
        x = np.ascontiguousarray(np.asarray(data, dtype=np.uint8))
        out = np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_poly_deserialize(self._h, x.ctypes.data_as(_lib.u8p), _ptr(out), b, 1 if to_ntt else 0))
        return out

    # -- rq/ops.rs ----------------------------------------------------------------------
    def add(self, a, b):
        return self._binary_inplace("fhe_poly_add", "fhe_poly_add_dev", a, b)

    def sub(self, a, b):
        return self._binary_inplace("fhe_poly_sub", "fhe_poly_sub_dev", a, b)

    def mul(self, a, b):
        return self._binary_inplace("fhe_poly_mul", "fhe_poly_mul_dev", a, b)

    def neg(self, a):
        return self._unary_inplace("fhe_poly_neg", "fhe_poly_neg_dev", a)

    def mul_shoup(self, a, b, b_shoup):
        L = _lib.lib()
        if _is_dev(a):
            check(L.fhe_poly_mul_shoup_dev(self._h, _dptr(a), _dptr(b), _dptr(b_shoup), self._batch(a), _stream()))
            return a
        x, y, ys = _np(a).copy(), _np(b), _np(b_shoup)
        check(L.fhe_poly_mul_shoup(self._h, _ptr(x), _ptr(y), _ptr(ys), self._batch(x)))
        return x

    def shoup(self, a):
        """Poly::compute_coefficients_shoup (host, setup-time)."""
        x = _np(a)
        out = np.zeros_like(x)
        check(_lib.lib().fhe_poly_shoup(self._h, _ptr(x), _ptr(out), self._batch(x)))
        return out

    def substitute(self, exponent, polys, ntt=True):
        """Poly::substitute with SubstitutionExponent::new(ctx, exponent)."""
        L = _lib.lib()
        if _is_dev(polys):
            out = _empty_dev(polys, polys.shape)
            check(L.fhe_poly_substitute_dev(self._h, exponent, _dptr(polys), _dptr(out), self._batch(polys),
                                            1 if ntt else 0, _stream()))
            return out
        x = _np(polys)
        out = np.zeros_like(x)
        check(L.fhe_poly_substitute(self._h, exponent, _ptr(x), _ptr(out), self._batch(x), 1 if ntt else 0))
        return out

    def switch_down(self, polys):
        """Poly::<PowerBasis>::switch_down: [..., L, N] -> [..., L-1, N]."""
        L = _lib.lib()
        b = self._batch(polys)
        oshape = tuple(polys.shape[:-2]) + (self.nmoduli - 1, self.degree)
        if _is_dev(polys):
            out = _empty_dev(polys, oshape)
            check(L.fhe_poly_switch_down_dev(self._h, _dptr(polys), _dptr(out), b, _stream()))
            return out
        x = _np(polys)
        out = np.zeros(oshape if self.nmoduli > 1 else (1,), dtype=np.uint64)
        check(L.fhe_poly_switch_down(self._h, _ptr(x), _ptr(out), b))
        return out

    def ciphertext_switch_down(self, ct):
        """Ciphertext::switch_down: [..., parts, L, N] Ntt -> [..., parts, L-1, N] Ntt."""
        L = _lib.lib()
        parts = ct.shape[-3]
        b = self._batch(ct) // parts
        oshape = tuple(ct.shape[:-2]) + (self.nmoduli - 1, self.degree)
        if _is_dev(ct):
            out = _empty_dev(ct, oshape)
            check(L.fhe_bfv_switch_down_dev(self._h, parts, _dptr(ct), _dptr(out), b, _stream()))
            return out
        x = _np(ct)
        out = np.zeros(oshape if self.nmoduli > 1 else (1,), dtype=np.uint64)
        check(L.fhe_bfv_switch_down(self._h, parts, _ptr(x), _ptr(out), b))
        return out

    def switch_down_to(self, polys, to_ctx):
        """Poly::<PowerBasis>::switch_down_to (rq/mod.rs:498-507): [..., L, N] -> [..., to.L, N] in one call."""
        L = _lib.lib()
        b = self._batch(polys)
        oshape = tuple(polys.shape[:-2]) + (to_ctx.nmoduli, self.degree)
        if _is_dev(polys):
            out = _empty_dev(polys, oshape)
            check(L.fhe_poly_switch_down_to_dev(self._h, to_ctx._h, _dptr(polys), _dptr(out), b, _stream()))
            return out
        x = _np(polys)
        out = np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_poly_switch_down_to(self._h, to_ctx._h, _ptr(x), _ptr(out), b))
        return out

    def ciphertext_switch_to_level(self, ct, levels):
        """Ciphertext::switch_to_level (bfv/ciphertext.rs:164-183), `levels` = target_level - level:
        [..., parts, L, N] Ntt -> [..., parts, L-levels, N] Ntt with one inverse / forward transform pair."""
        L = _lib.lib()
        parts = ct.shape[-3]
        b = self._batch(ct) // parts
        oshape = tuple(ct.shape[:-2]) + (max(self.nmoduli - levels, 0), self.degree)
        if _is_dev(ct):
            out = _empty_dev(ct, oshape if self.nmoduli > levels else (1,))
            check(L.fhe_bfv_switch_to_level_dev(self._h, levels, parts, _dptr(ct), _dptr(out), b, _stream()))
            return out
        x = _np(ct)
        out = np.zeros(oshape if self.nmoduli > levels else (1,), dtype=np.uint64)
        check(L.fhe_bfv_switch_to_level(self._h, levels, parts, _ptr(x), _ptr(out), b))
        return out

    def dot_product_scalar(self, cts, pts):
        """bfv::dot_product_scalar (crates/fhe/src/bfv/ops/dot_product.rs:54-180):
        cts [..., count, parts, L, N] and pts [..., count, L, N] (`Plaintext::poly_ntt`), or either
        without the leading batch dims (then shared by the batch) -> [..., parts, L, N].
        With parts == 1 this is fhe_math::rq::dot_product (rq/ops.rs:449-570)."""
        L = _lib.lib()
        cb, pb = tuple(cts.shape[:-4]), tuple(pts.shape[:-3])
        count, parts = cts.shape[-4], cts.shape[-3]
        if pts.shape[-3] != count:
            raise FheError(-1, "DotProductLengthMismatch")
        bshape = cb if len(cb) >= len(pb) else pb
        batch = 1
        for d in bshape:
            batch *= d
        cts_shared = 1 if (cb != bshape) else 0
        pts_shared = 1 if (pb != bshape) else 0
        oshape = bshape + (parts, self.nmoduli, self.degree)
        if _is_dev(cts):
            out = _empty_dev(cts, oshape)
            check(L.fhe_bfv_dot_product_scalar_dev(self._h, parts, count, _dptr(cts), cts_shared, _dptr(pts), pts_shared,
                                                   _dptr(out), batch, _stream()))
            return out
        x, y = _np(cts), _np(pts)
        out = np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_bfv_dot_product_scalar(self._h, parts, count, _ptr(x), cts_shared, _ptr(y), pts_shared, _ptr(out), batch))
        return out

    def mul_plain(self, ct, pt):
        """`&Ciphertext * &Plaintext` (ops/mod.rs:229-257): ct [..., parts, L, N] times pt [..., L, N]
        (or [L, N] shared by the batch)."""
        L = _lib.lib()
        parts = ct.shape[-3]
        batch = self._batch(ct) // parts
        shared = 1 if len(pt.shape) == 2 and len(ct.shape) > 3 else 0
        if _is_dev(ct):
            out = _empty_dev(ct, ct.shape)
            check(L.fhe_bfv_mul_plain_dev(self._h, parts, _dptr(ct), _dptr(pt), shared, _dptr(out), batch, _stream()))
            return out
        x, y = _np(ct), _np(pt)
        out = np.zeros_like(x)
        check(L.fhe_bfv_mul_plain(self._h, parts, _ptr(x), _ptr(y), shared, _ptr(out), batch))
        return out

    def random_from_seed(self, seeds):
        """Poly::<Ntt>::random_from_seed (rq/mod.rs:276-292) per 32-byte seed: seeds [batch, 32] uint8 ->
        [batch, L, N]; numpy in -> numpy out, torch CUDA uint8 tensor in -> CUDA tensor out.

        PARITY UNPINNED (as include/fhe_hip.h says for fhe_poly_from_seed): SHA-256 and the ChaCha block function are
        known-answer tested, but the generator's word layout and the `Uniform` rejection sampler restate
        rand_chacha 0.10 / rand 0.10 -- crates the reference does not vendor -- and no vector from a real fhe.rs run
        exists here.  The test-side oracle was written from the same reading, so agreement with it does not pin the
        stream either.  A host that must interoperate with seeded ciphertexts made by fhe.rs expands c1 itself and
        uploads it (or first checks one (seed, modulus) pair against its Rust build)."""
        L = _lib.lib()
        if _is_dev(seeds):
            b = int(seeds.numel() // 32)
            out = _empty_dev(seeds, (b, self.nmoduli, self.degree))
            check(L.fhe_poly_from_seed_dev(self._h, _dptr8(seeds), _dptr(out), b, _stream()))
            return out
        sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8)).reshape(-1, 32)
        out = np.zeros((sd.shape[0], self.nmoduli, self.degree), dtype=np.uint64)
        check(L.fhe_poly_from_seed(self._h, sd.ctypes.data_as(_lib.u8p), _ptr(out), sd.shape[0]))
        return out

    def synth_uniform(self, seed, ct0, part0, nparts, batch):
        """Device-side synthetic residues [batch, nparts, L, N] (bench / parity inputs)."""
        shape = (batch, nparts, self.nmoduli, self.degree)
        if getattr(_tls, "stream", None) is not None or torch is None or not torch.cuda.is_available():
            out = DeviceArray(shape, max(self.device, 0))   # a Stream is current: stay on the ABI's own allocator
        else:
            out = torch.empty(shape, dtype=torch.int64, device=f"cuda:{self.device}")
        check(_lib.lib().fhe_synth_uniform_dev(self._h, seed, ct0, part0, nparts, _dptr(out), batch, _stream()))
        return out


class Scaler:
    """rq::scaler::Scaler (crates/fhe-math/src/rq/scaler.rs:18-127); the scaling factor is
    ScalingFactor::new(numerator, denominator) given as Python ints."""

    def __init__(self, from_ctx, to_ctx, numerator=1, denominator=1, _handle=None, _owner=None):
        self.from_ctx, self.to_ctx = from_ctx, to_ctx
        self._owner = _owner
        if _handle is not None:
            self._h = _handle
        else:
            n, d = _limbs(numerator), _limbs(denominator)
            h = C.c_void_p()
            check(_lib.lib().fhe_scaler_create(from_ctx._h, to_ctx._h, _ptr(n), len(n), _ptr(d), len(d), C.byref(h)))
            self._h = h
        self.number_common_moduli = _lib.lib().fhe_scaler_number_common_moduli(self._h)

    @classmethod
    def from_constants(cls, from_ctx, to_ctx, number_common_moduli, is_one, k):
        """All RnsScaler fields supplied by the caller (what a Rust host already holds)."""
        keep = {n: _np(k[n]) for n in ("gamma", "gamma_shoup", "omega", "omega_shoup", "theta_omega_lo",
                                        "theta_omega_hi", "theta_garner_lo", "theta_garner_hi")}
        sign = np.ascontiguousarray(np.asarray(k["theta_omega_sign"], dtype=np.uint8))
        h = C.c_void_p()
        check(_lib.lib().fhe_scaler_create_from_constants(
            from_ctx._h, to_ctx._h, number_common_moduli, 1 if is_one else 0,
            _ptr(keep["gamma"]), _ptr(keep["gamma_shoup"]), _ptr(keep["omega"]), _ptr(keep["omega_shoup"]),
            int(k["theta_gamma_lo"]), int(k["theta_gamma_hi"]), 1 if k["theta_gamma_sign"] else 0,
            _ptr(keep["theta_omega_lo"]), _ptr(keep["theta_omega_hi"]), sign.ctypes.data_as(_lib.u8p),
            _ptr(keep["theta_garner_lo"]), _ptr(keep["theta_garner_hi"]), int(k["theta_garner_shift"]), C.byref(h)))
        return cls(from_ctx, to_ctx, _handle=h)

    def __del__(self):
        if self._owner is None and getattr(self, "_h", None) is not None and _lib._lib is not None:
            _lib._lib.fhe_scaler_destroy(self._h)

    def constants(self, which):
        nf, nt = self.from_ctx.nmoduli, self.to_ctx.nmoduli
        size = {0: nt, 1: nt, 2: nt * nf, 3: nt * nf, 4: nf, 5: nf, 6: nf, 7: nf, 8: nf, 9: 5}[which]
        out = np.zeros(size, dtype=np.uint64)
        check(_lib.lib().fhe_scaler_get_constants(self._h, which, _ptr(out)))
        return out

    def scale(self, polys, ntt):
        """Scaler::scale == Poly::scale(&scaler): [..., Lfrom, N] -> [..., Lto, N]."""
        L = _lib.lib()
        b = self.from_ctx._batch(polys)
        oshape = tuple(polys.shape[:-2]) + (self.to_ctx.nmoduli, self.to_ctx.degree)
        if _is_dev(polys):
            out = _empty_dev(polys, oshape, zero=True)
            check(L.fhe_poly_scale_dev(self._h, _dptr(polys), _dptr(out), b, 1 if ntt else 0, _stream()))
            return out
        x = _np(polys)
        out = np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_poly_scale(self._h, _ptr(x), _ptr(out), b, 1 if ntt else 0))
        return out


class Switcher(Scaler):
    """rq::switcher::Switcher (crates/fhe-math/src/rq/switcher.rs:11-26)."""

    def __init__(self, from_ctx, to_ctx):
        h = C.c_void_p()
        check(_lib.lib().fhe_switcher_create(from_ctx._h, to_ctx._h, C.byref(h)))
        super().__init__(from_ctx, to_ctx, _handle=h)

    def switch(self, polys, ntt=False):
        return self.scale(polys, ntt)


class KeySwitchingKey:
    """bfv::KeySwitchingKey (crates/fhe/src/bfv/keys/key_switching_key.rs:22-46): c0, c1 are
    `[ndigits, Lk, N]` NttShoup polynomials over ctx_ksk; Shoup twins are optional."""

    def __init__(self, ctx_ciphertext, ctx_ksk, c0, c1, c0_shoup=None, c1_shoup=None, log_base=0):
        self.ctx_ciphertext, self.ctx_ksk, self.log_base = ctx_ciphertext, ctx_ksk, log_base
        L = _lib.lib()
        h = C.c_void_p()
        if _is_dev(c0):
            nd = c0.shape[0]
            check(L.fhe_ksk_create_dev(ctx_ciphertext._h, ctx_ksk._h, nd, _dptr(c0), _dptr(c1), log_base, _stream(),
                                       C.byref(h)))
        else:
            a0, a1 = _np(c0), _np(c1)
            nd = a0.shape[0]
            s0 = _np(c0_shoup) if c0_shoup is not None else None
            s1 = _np(c1_shoup) if c1_shoup is not None else None
            check(L.fhe_ksk_create(ctx_ciphertext._h, ctx_ksk._h, nd, _ptr(a0), _ptr(s0) if s0 is not None else None,
                                   _ptr(a1), _ptr(s1) if s1 is not None else None, log_base, C.byref(h)))
        self.ndigits = nd
        self._h = h
        forced = getattr(_tls, "ks_forced_mode", (0, 0))   # (tests and A/B tools: keys made inside `forced_mode`)
        if forced != (0, 0):
            self.set_mode(*forced)

    @classmethod
    def forced_mode(cls, mode, w_budget=0):
        """Context manager for TEST code: keys created inside it BY THIS THREAD take `mode` (the parity suites run their
        key-switch cases once per evaluation strategy this way).  Thread-local, restored on exit; the library itself
        has no process-wide switch, and product code sets the mode per key (`set_mode`)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = getattr(_tls, "ks_forced_mode", (0, 0))
            _tls.ks_forced_mode = (int(mode), int(w_budget))
            try:
                yield
            finally:
                _tls.ks_forced_mode = old
        return cm()

    def __del__(self):
        if getattr(self, "_h", None) is not None and _lib._lib is not None:
            _lib._lib.fhe_ksk_destroy(self._h)

    AUTO, FUSED, UNFUSED, UNFUSED_SUB, FUSED_SUB = 0, 1, 2, 3, 4

    def set_mode(self, mode, w_budget=0):
        """How every key switch through this handle is evaluated (fhe_ksk_set_mode): AUTO, FUSED, UNFUSED (batched
        digit transforms + streaming lazy MAC), UNFUSED_SUB, FUSED_SUB (N >= 32768 on 8192-point sub-blocks);
        `w_budget` bytes of transformed rows per launch pair."""
        check(_lib.lib().fhe_ksk_set_mode(self._h, int(mode), int(w_budget)))
        return self

    def mode(self):
        m, w = C.c_int(), C.c_size_t()
        check(_lib.lib().fhe_ksk_get_mode(self._h, C.byref(m), C.byref(w)))
        return dict(mode=m.value, w_budget=w.value)

    def key_switch(self, p):
        """KeySwitchingKey::key_switch: p [..., L, N] PowerBasis -> (c0, c1) [..., Lk, N] Ntt."""
        L = _lib.lib()
        b = self.ctx_ciphertext._batch(p)
        oshape = tuple(p.shape[:-2]) + (self.ctx_ksk.nmoduli, self.ctx_ksk.degree)
        if _is_dev(p):
            o0, o1 = _empty_dev(p, oshape), _empty_dev(p, oshape)
            check(L.fhe_key_switch_dev(self._h, _dptr(p), _dptr(o0), _dptr(o1), b, _stream()))
            return o0, o1
        x = _np(p)
        o0, o1 = np.zeros(oshape, dtype=np.uint64), np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_key_switch(self._h, _ptr(x), _ptr(o0), _ptr(o1), b))
        return o0, o1


class RelinearizationKey:
    """bfv::RelinearizationKey (crates/fhe/src/bfv/keys/relinearization_key.rs:24-110)."""

    def __init__(self, ksk: KeySwitchingKey):
        if ksk.ctx_ksk.nmoduli == 1:
            raise FheError(-17, "KeySwitchingNotSupported")
        self.ksk = ksk

    def relinearizes(self, ct3):
        """[..., 3, L, N] Ntt -> [..., 2, L, N] Ntt."""
        L = _lib.lib()
        ctx = self.ksk.ctx_ciphertext
        if ct3.shape[-3] != 3:
            raise FheError(-13, "InvalidPolynomialCount: relinearization expects 3 parts")
        b = ctx._batch(ct3) // 3
        oshape = tuple(ct3.shape[:-3]) + (2, ctx.nmoduli, ctx.degree)
        if _is_dev(ct3):
            out = _empty_dev(ct3, oshape)
            check(L.fhe_bfv_relinearize_dev(self.ksk._h, _dptr(ct3), _dptr(out), b, _stream()))
            return out
        x = _np(ct3)
        out = np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_bfv_relinearize(self.ksk._h, _ptr(x), _ptr(out), b))
        return out


class GaloisKey:
    """bfv::GaloisKey (crates/fhe/src/bfv/keys/galois_key.rs:19-123)."""

    def __init__(self, ksk: KeySwitchingKey, exponent: int):
        n = ksk.ctx_ciphertext.degree
        if (exponent % (2 * n)) & 1 == 0:
            raise FheError(-10, "InvalidSubstitutionExponent")
        self.ksk, self.exponent = ksk, exponent % (2 * n)

    def relinearize(self, ct):
        """[..., 2, L, N] Ntt -> same shape."""
        L = _lib.lib()
        ctx = self.ksk.ctx_ciphertext
        if ct.shape[-3] != 2:
            raise FheError(-13, "InvalidPolynomialCount: rotation expects 2 parts")
        b = ctx._batch(ct) // 2
        if _is_dev(ct):
            out = _empty_dev(ct, ct.shape)
            check(L.fhe_bfv_galois_dev(self.ksk._h, self.exponent, _dptr(ct), _dptr(out), b, _stream()))
            return out
        x = _np(ct)
        out = np.zeros_like(x)
        check(L.fhe_bfv_galois(self.ksk._h, self.exponent, _ptr(x), _ptr(out), b))
        return out


class EvaluationKey:
    """The rotation part of bfv::EvaluationKey (crates/fhe/src/bfv/keys/evaluation_key.rs:
    rotates_rows :110-131, rotates_columns_by :145-170, exponent map :278-286)."""

    def __init__(self, degree, galois_keys):
        self.degree = degree
        self.gk = {g.exponent: g for g in galois_keys}

    def rotates_rows(self, ct):
        e = 2 * self.degree - 1
        if e not in self.gk:
            raise FheError(-11, "EvaluationKeyError::Unsupported(RowRotation)")
        return self.gk[e].relinearize(ct)

    def computes_inner_sum(self, ct):
        """EvaluationKey::computes_inner_sum (evaluation_key.rs:56-100)."""
        n = self.degree
        seq, i = [], 1
        while i < n // 2:
            seq.append(pow(3, i, 2 * n))
            i *= 2
        seq.append(2 * n - 1)
        if any(e not in self.gk for e in seq):
            raise FheError(-11, "EvaluationKeyError::Unsupported(InnerSum)")
        L = _lib.lib()
        handles = (C.c_void_p * len(seq))(*[self.gk[e].ksk._h for e in seq])
        exps = (C.c_size_t * len(seq))(*seq)
        ctx = self.gk[seq[0]].ksk.ctx_ciphertext
        b = ctx._batch(ct) // 2
        if _is_dev(ct):
            out = _empty_dev(ct, ct.shape)
            check(L.fhe_bfv_inner_sum_dev(handles, exps, len(seq), _dptr(ct), _dptr(out), b, _stream()))
            return out
        x = _np(ct)
        out = np.zeros_like(x)
        check(L.fhe_bfv_inner_sum(handles, exps, len(seq), _ptr(x), _ptr(out), b))
        return out

    def supports_expansion(self, level):
        """evaluation_key.rs:174-186: the Galois keys of (N >> l) + 1, l < level, are present."""
        return level <= self.degree.bit_length() - 1 and all((self.degree >> l) + 1 in self.gk for l in range(level))

    def expands(self, ct, size):
        """EvaluationKey::expands (evaluation_key.rs:192-256): ct [2, L, N] (or [batch, 2, L, N])
        -> [size, 2, L, N] (or [size, batch, 2, L, N])."""
        if size == 0 or size > self.degree:
            raise FheError(-20, "InvalidExpansionSize")
        level = (size - 1).bit_length()
        if not self.supports_expansion(level):
            raise FheError(-21, "EvaluationKeyError::Unsupported(Expansion)")
        L = _lib.lib()
        if level == 0:   # the reference returns vec![ct.clone()] without looking at any key
            return (ct.clone() if _is_dev(ct) else _np(ct).copy())[None]
        handles = (C.c_void_p * level)(*[self.gk[(self.degree >> l) + 1].ksk._h for l in range(level)])
        ctx = self.gk[self.degree + 1].ksk.ctx_ciphertext
        b = ctx._batch(ct) // 2
        oshape = (size,) + tuple(ct.shape)
        if _is_dev(ct):
            out = _empty_dev(ct, oshape)
            check(L.fhe_bfv_expand_dev(handles, level, _dptr(ct), _dptr(out), size, b, _stream()))
            return out
        x = _np(ct)
        out = np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_bfv_expand(handles, level, _ptr(x), _ptr(out), size, b))
        return out

    def rotates_columns_by(self, ct, i):
        if not (1 <= i < self.degree // 2):
            raise FheError(-1, "InvalidRotationStep")
        e = pow(3, i, 2 * self.degree)
        if e not in self.gk:
            raise FheError(-11, "EvaluationKeyError::Unsupported(ColumnRotation)")
        return self.gk[e].relinearize(ct)


class RGSWCiphertext:
    """bfv::RGSWCiphertext{ksk0, ksk1} (crates/fhe/src/bfv/rgsw_ciphertext.rs:19-156)."""

    def __init__(self, ksk0: KeySwitchingKey, ksk1: KeySwitchingKey):
        self.ksk0, self.ksk1 = ksk0, ksk1

    def external_product(self, ct):
        """`&Ciphertext * &RGSWCiphertext`: ct, result [..., 2, L, N] Ntt."""
        L = _lib.lib()
        ctx = self.ksk0.ctx_ciphertext
        if ct.shape[-3] != 2:
            raise FheError(-13, "Ciphertext must have two parts")
        b = ctx._batch(ct) // 2
        if _is_dev(ct):
            out = _empty_dev(ct, ct.shape)
            check(L.fhe_bfv_rgsw_mul_dev(self.ksk0._h, self.ksk1._h, _dptr(ct), _dptr(out), b, _stream()))
            return out
        x = _np(ct)
        out = np.zeros_like(x)
        check(L.fhe_bfv_rgsw_mul(self.ksk0._h, self.ksk1._h, _ptr(x), _ptr(out), b))
        return out


class BfvParameters:
    """The device-table part of bfv::BfvParameters (crates/fhe/src/bfv/parameters.rs:560-738):
    per-level contexts, the extended multiplication basis and per-level mul parameters."""

    def __init__(self, degree, plaintext_modulus, moduli=None, moduli_sizes=None, device=0, tables_fn=None):
        """tables_fn(modulus, degree) -> dict(omegas, omegas_shoup, zetas_inv, zetas_inv_shoup, size_inv,
        size_inv_shoup): the host's NttOperator tables for every modulus the parameter set builds a context over
        (fhe_params_create_with_tables); None: the engine's own psi (fhe_params_create)."""
        L = _lib.lib()
        if moduli_sizes:
            sizes = (C.c_size_t * len(moduli_sizes))(*moduli_sizes)
            out = np.zeros(len(moduli_sizes), dtype=np.uint64)
            check(L.fhe_generate_moduli(sizes, len(moduli_sizes), degree, _ptr(out)))
            moduli = [int(x) for x in out]
        m = _np(moduli)
        h = C.c_void_p()
        if tables_fn is None:
            check(L.fhe_params_create(device, degree, len(m), _ptr(m), plaintext_modulus, C.byref(h)))
        else:
            def _cb(_user, modulus, deg, om, oms, zi, zis, si, sis):
                try:
                    t = tables_fn(int(modulus), int(deg))
                    arrs = [_np(t[key]).reshape(-1) for key in ("omegas", "omegas_shoup", "zetas_inv", "zetas_inv_shoup")]
                    if any(a.size != deg for a in arrs):
                        return 1   # a short table would be read out of bounds: creation fails (NttOperatorUnavailable)
                    for dst, a in zip((om, oms, zi, zis), arrs):
                        C.memmove(dst, a.ctypes.data, 8 * deg)
                    si[0], sis[0] = int(t["size_inv"]), int(t["size_inv_shoup"])
                    return 0
                except Exception:
                    return 1
            cb = _lib.NTT_TABLES_FN(_cb)
            check(L.fhe_params_create_with_tables(device, degree, len(m), _ptr(m), plaintext_modulus, cb, None,
                                                  C.byref(h)))
            # (the C side calls `cb` only while fhe_params_create_with_tables runs and caches every table per modulus:
            # nothing has to keep the callback alive afterwards)
        self._h = h
        self.degree, self.plaintext, self.moduli, self.device = degree, plaintext_modulus, [int(x) for x in m], device
        self.max_level = L.fhe_params_max_level(h)

    def __del__(self):
        if getattr(self, "_h", None) is not None and _lib._lib is not None:
            _lib._lib.fhe_params_destroy(self._h)

    def _get(self, fn, level):
        h = C.c_void_p()
        check(getattr(_lib.lib(), fn)(self._h, level, C.byref(h)))
        return h

    def context_at_level(self, level):
        return Context(None, None, _handle=self._get("fhe_params_ctx", level), _owner=self)

    def mul_context_at_level(self, level):
        return Context(None, None, _handle=self._get("fhe_params_mul_ctx", level), _owner=self)

    def extender(self, level):
        return Scaler(self.context_at_level(level), self.mul_context_at_level(level),
                      _handle=self._get("fhe_params_extender", level), _owner=self)

    def down_scaler(self, level):
        return Scaler(self.mul_context_at_level(level), self.context_at_level(level),
                      _handle=self._get("fhe_params_down_scaler", level), _owner=self)

    def plaintext_context(self):
        """parameters.rs:578-595: the shortest prefix of the moduli with >= bits(t) + 60 bits."""
        if getattr(self, "_plain_ctx", None) is None:
            acc, count = 0, 0
            for m in self.moduli:
                acc += int(m).bit_length()
                count += 1
                if acc >= int(self.plaintext).bit_length() + 60:
                    break
            self._plain_ctx = Context(self.moduli[:max(1, count)], self.degree, device=self.device)
        return self._plain_ctx

    def plain_scaler(self, level):
        """CipherPlainContext::scaler (parameters.rs:636-643): ciphertext level -> plaintext context, t / q."""
        cache = self.__dict__.setdefault("_plain_scalers", {})
        if level not in cache:
            ctx = self.context_at_level(level)
            q = 1
            for m in ctx.moduli:
                q *= int(m)
            cache[level] = Scaler(ctx, self.plaintext_context(), int(self.plaintext), q)
        return cache[level]

    def decrypt(self, s_ntt, ct, level=0):
        """SecretKey::try_decrypt (secret_key.rs:198-247, small plaintext modulus): s_ntt [L, N] is the
        secret key over the ciphertext context in Ntt form; ct [..., nparts, L, N] -> [..., N] in [0, t)."""
        L = _lib.lib()
        sc = self.plain_scaler(level)
        nparts = int(ct.shape[-3])
        b = sc.from_ctx._batch(ct) // nparts
        oshape = tuple(ct.shape[:-3]) + (self.degree,)
        if _is_dev(ct):
            out = _empty_dev(ct, oshape)
            check(L.fhe_bfv_decrypt_dev(sc._h, int(self.plaintext), _dptr(s_ntt), _dptr(ct), nparts, _dptr(out), b,
                                        _stream()))
            return out
        x, sk = _np(ct), _np(s_ntt)
        out = np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_bfv_decrypt(sc._h, int(self.plaintext), _ptr(sk), _ptr(x), nparts, _ptr(out), b))
        return out


class Multiplicator:
    """bfv::Multiplicator (crates/fhe/src/bfv/ops/mul.rs:21-243)."""

    def __init__(self, extender_lhs, extender_rhs, down_scaler, rk=None, mod_switch=False, _handle=None, _keep=()):
        self._keep = (extender_lhs, extender_rhs, down_scaler, rk) + tuple(_keep)
        if _handle is None:
            h = C.c_void_p()
            check(_lib.lib().fhe_mul_create(extender_lhs._h, extender_rhs._h, down_scaler._h,
                                            rk.ksk._h if rk is not None else None, 1 if mod_switch else 0, C.byref(h)))
            _handle = h
        self._h = _handle
        p, r = C.c_size_t(), C.c_size_t()
        check(_lib.lib().fhe_mul_out_shape(self._h, C.byref(p), C.byref(r)))
        self.out_parts, self.out_rows = p.value, r.value

    @classmethod
    def default(cls, params: BfvParameters, rk: RelinearizationKey = None, level=0, mod_switch=False):
        """Multiplicator::default(rk) (+ enable_mod_switching); rk=None gives `&ct * &ct`."""
        h = C.c_void_p()
        check(_lib.lib().fhe_mul_create_default(params._h, level, rk.ksk._h if rk is not None else None,
                                                1 if mod_switch else 0, C.byref(h)))
        m = cls(None, None, None, rk, mod_switch, _handle=h, _keep=(params,))
        m.base_ctx = params.context_at_level(level)
        return m

    def __del__(self):
        if getattr(self, "_h", None) is not None and _lib._lib is not None:
            _lib._lib.fhe_mul_destroy(self._h)

    def basis(self):
        """Moduli of the multiplication context (Multiplicator::mul_ctx)."""
        n = C.c_size_t()
        check(_lib.lib().fhe_mul_basis(self._h, C.byref(n), None))
        out = np.zeros(n.value, dtype=np.uint64)
        check(_lib.lib().fhe_mul_basis(self._h, C.byref(n), _ptr(out)))
        return [int(x) for x in out]

    def set_chunk(self, chunk):
        """Ciphertext pairs per pipeline pass (0 = default); an option of this handle (fhe_mul_set_chunk)."""
        check(_lib.lib().fhe_mul_set_chunk(self._h, chunk))
        return self

    def set_streams(self, n):
        """2 (default): chunks alternate between the caller's stream and an internal one; 1: caller's stream only."""
        check(_lib.lib().fhe_mul_set_streams(self._h, n))
        return self

    def options(self):
        c, s = C.c_size_t(), C.c_size_t()
        check(_lib.lib().fhe_mul_get_options(self._h, C.byref(c), C.byref(s)))
        return dict(chunk=c.value, streams=s.value)

    def multiply(self, lhs, rhs):
        """Multiplicator::multiply: lhs, rhs [..., 2, L, N] Ntt -> [..., parts, rows, N] Ntt."""
        L = _lib.lib()
        if tuple(lhs.shape) != tuple(rhs.shape) or lhs.shape[-3] != 2:
            raise FheError(-13, "MultiplicationPolynomialCount: expected 2-part ciphertexts of equal shape")
        b = 1
        for d in lhs.shape[:-3]:
            b *= d
        oshape = tuple(lhs.shape[:-3]) + (self.out_parts, self.out_rows, lhs.shape[-1])
        if _is_dev(lhs):
            out = _empty_dev(lhs, oshape)
            check(L.fhe_bfv_mul_dev(self._h, _dptr(lhs), _dptr(rhs), _dptr(out), b, _stream()))
            return out
        x, y = _np(lhs), _np(rhs)
        out = np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_bfv_mul(self._h, _ptr(x), _ptr(y), _ptr(out), b))
        return out

    def tensor(self, lhs, rhs):
        """`&ct * &ct` (ops/mod.rs:259-358) for any number of parts: lhs [..., la, L, N], rhs [..., lb, L, N]
        -> [..., la + lb - 1, L, N]; no relinearisation."""
        L = _lib.lib()
        if tuple(lhs.shape[:-3]) != tuple(rhs.shape[:-3]) or tuple(lhs.shape[-2:]) != tuple(rhs.shape[-2:]):
            raise FheError(-11, "ParameterMismatch: operands of different batch shape or level")
        la, lb = int(lhs.shape[-3]), int(rhs.shape[-3])
        b = 1
        for d in lhs.shape[:-3]:
            b *= d
        oshape = tuple(lhs.shape[:-3]) + (la + lb - 1,) + tuple(lhs.shape[-2:])
        if _is_dev(lhs):
            out = _empty_dev(lhs, oshape)
            check(L.fhe_bfv_tensor_dev(self._h, la, lb, _dptr(lhs), _dptr(rhs), _dptr(out), b, _stream()))
            return out
        x, y = _np(lhs), _np(rhs)
        out = np.zeros(oshape, dtype=np.uint64)
        check(L.fhe_bfv_tensor(self._h, la, lb, _ptr(x), _ptr(y), _ptr(out), b))
        return out


# ---- zq::primes (host) ---------------------------------------------------------------------
def generate_prime(num_bits, modulo, upper_bound):
    p = _lib.lib().fhe_generate_prime(num_bits, modulo, upper_bound)
    return p or None


def supports_opt(p):
    return bool(_lib.lib().fhe_supports_opt(p))


def is_prime(p):
    return bool(_lib.lib().fhe_is_prime(p))


def generate_moduli(sizes, degree):
    arr = (C.c_size_t * len(sizes))(*sizes)
    out = np.zeros(len(sizes), dtype=np.uint64)
    check(_lib.lib().fhe_generate_moduli(arr, len(sizes), degree, _ptr(out)))
    return [int(x) for x in out]


def device_count():
    return _lib.lib().fhe_device_count()


# ---- execution options ------------------------------------------------------------------------------
def set_f64(on=True):
    """fhe_engine_set_f64: rows whose moduli are all below 2^50 take the FP64-FMA kernels (default) or the integer ones."""
    _lib.lib().fhe_engine_set_f64(1 if on else 0)


def get_f64():
    return bool(_lib.lib().fhe_engine_get_f64())


# ---- profiling ---------------------------------------------------------------------------------
def prof_enable(on=True):
    _lib.lib().fhe_prof_enable(1 if on else 0)


def prof_reset():
    _lib.lib().fhe_prof_reset()


def prof_report():
    """{launch label: (launches, total_ms)} from the HIP events the launches carried (entries of one label summed)."""
    out = {}
    for label, _sym, n, ms in prof_entries():
        a = out.get(label, (0, 0.0))
        out[label] = (a[0] + n, a[1] + ms)
    return out


def prof_entries():
    """[(launch label, kernel symbol, launches, total_ms)]: one entry per kernel instantiation, named as rocprofv3 names
    it (fhe_prof_get + fhe_prof_get_symbol)."""
    L = _lib.lib()
    out = []
    for i in range(L.fhe_prof_count()):
        name, sym = C.create_string_buffer(64), C.create_string_buffer(512)
        n, ms = C.c_uint64(), C.c_double()
        check(L.fhe_prof_get(i, name, 64, C.byref(n), C.byref(ms)))
        check(L.fhe_prof_get_symbol(i, sym, 512))
        out.append((name.value.decode(), sym.value.decode(), n.value, ms.value))
    return out


def workspace_trim():
    """Frees the engine's idle scratch buffers; returns the bytes released."""
    return int(_lib.lib().fhe_workspace_trim())


WORKSPACE_DEFAULT = (1 << 64) - 1


def workspace_set_limit(per_stream_bytes=0, total_bytes=WORKSPACE_DEFAULT):
    """Bounds on the scratch the engine retains between calls (fhe_workspace_set_limit; 0 = none; the default total
    is a quarter of the device's memory)."""
    check(_lib.lib().fhe_workspace_set_limit(int(per_stream_bytes), int(total_bytes)))


def workspace_stats():
    h, u, b, o, a = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    check(_lib.lib().fhe_workspace_stats(C.byref(h), C.byref(u), C.byref(b), C.byref(o), C.byref(a)))
    return dict(held_bytes=h.value, in_use_bytes=u.value, blocks=b.value, owners=o.value, internal_streams=a.value)


def workspace_pool_stats(device=0):
    """What the driver says the library's two private pools on `device` hold (fhe_workspace_pool_stats)."""
    v = [C.c_size_t() for _ in range(4)]
    check(_lib.lib().fhe_workspace_pool_stats(int(device), *[C.byref(x) for x in v]))
    return dict(scratch_reserved_bytes=v[0].value, scratch_used_bytes=v[1].value,
                buffers_reserved_bytes=v[2].value, buffers_used_bytes=v[3].value)


UBENCH_KINDS = {"mad_u64_u32": 0, "mul_lo_u32": 1, "mul_hi_u32": 2, "shoup_lazy": 3, "fwd_butterfly": 4,
                "fwd_butterfly_narrow": 5, "inv_butterfly": 6, "shoup_mac": 7, "tensor_mul": 8, "tensor_mac2": 9,
                # round 6: the FP64-FMA forms for moduli below 2^50 (csrc/zq_f64.hpp)
                "f64_fma": 10, "f64_rndne": 11, "f64_mulmod": 12, "f64_fwd_butterfly": 13, "f64_inv_butterfly": 14,
                "f64_mac": 15}


def ubench_int(kind, min_seconds=0.05, device=0):
    """Register-resident integer-issue microbenchmark (fhe_ubench_int): lane-operations per second chip-wide."""
    v = C.c_double()
    check(_lib.lib().fhe_ubench_int(device, UBENCH_KINDS[kind] if isinstance(kind, str) else int(kind),
                                    float(min_seconds), C.byref(v)))
    return v.value


def ubench_scaler(scaler, min_seconds=0.05):
    """RnsScaler::scale's no-HBM ceiling for this scaler's kernel instance (fhe_ubench_scaler): columns per second."""
    v = C.c_double()
    check(_lib.lib().fhe_ubench_scaler(scaler._h, float(min_seconds), C.byref(v)))
    return v.value


def ubench_copy(nbytes, min_seconds=0.05, device=0):
    """The box's streaming rate (fhe_ubench_copy): read + write bytes per second of a 16-byte-per-lane copy."""
    v = C.c_double()
    check(_lib.lib().fhe_ubench_copy(int(device), int(nbytes), float(min_seconds), C.byref(v)))
    return v.value


def device_mem_info(device=0):
    f, t = C.c_size_t(), C.c_size_t()
    check(_lib.lib().fhe_device_mem_info(device, C.byref(f), C.byref(t)))
    return f.value, t.value
