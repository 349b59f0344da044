"""Deferred evaluation of the reference's camera loop.

The reference calls the projector once per (camera, scale) and sums in Python (``vfa/model/vfanet.py:64-82``):

    for cam in range(N):
        f8, f16, f32 = self.vfa8(lat8, calib, grid), self.vfa16(lat16, calib, grid), self.vfa32(lat32, calib, grid)
        ortho += f8 + f16 + f32

With the one-line import swap of INTEGRATION.md that loop runs on this build's ``VFA`` modules.  Executed call by call it is
HOST-bound: 21 calls x ~275 us of Python / launch overhead = 5.8 ms per MultiviewC frame, ten times the batched frame, and every call
launches the persistent kernel for a twenty-first of its work.  So in inference (no gradient wanted) ``VFA.forward`` returns a
``DeferredOrtho``: a ``torch.Tensor`` subclass without storage (a wrapper: shape, dtype and device are real) that holds a record of
(module, feature map, calibration), adds to other records of the same grid (``f8 + f16 + f32``, ``ortho += ...``, ``0 + ...``) and is
replaced by the real ``(1, C, L, W)`` tensor the first time anything else is asked of it -- any ``torch`` function (``self.fuse(ortho)``
is ``F.conv2d``), any tensor method, indexing, arithmetic with a real tensor (all of them arrive at ``__torch_function__``).  By then the record holds the whole frame, and ONE batched launch (``fused_frame`` / ``pipe_frame``:
the path ``aggregate_views`` takes) computes it: the loop costs the batched frame plus bookkeeping.

Sums are re-associated exactly as ``aggregate_views`` re-associates them (inside the post-GEMM tolerance; pre-GEMM tensors are
bitwise either way).  A feature map, calibration or grid modified IN PLACE between the call and the use would be read in its new
state: the record keeps the tensors' version counters and raises if one moved (inference tensors -- ``torch.inference_mode()`` -- have no
counter and cannot be modified in place outside the mode: nothing to check).  ``VFA_AMD_LAZY=0`` switches the deferral off (every call
computes at once).

Caveats, by construction:
* ``a += b`` on two records returns a NEW record (a wrapper has no storage to update): another name bound to ``a`` before the ``+=``
  keeps the old sum, where a real tensor would have seen the update.  The reference's loop rebinds ``ortho`` and holds no alias.
* The frame is computed on the stream that is current at FIRST USE, not on the one the feature maps were produced on; a caller that
  moves between streams between the call and the use orders them itself (as it would for any tensor produced on another stream).
* Every Python-level use (functions, methods, ``data_ptr()``, ``numpy()``, ``torch.save``, ``copy.deepcopy``, ``__dlpack__`` /
  ``torch.from_dlpack``) goes through ``__torch_function__`` and sees the real tensor; operators reached below the Python API arrive at
  ``__torch_dispatch__`` and do too.  The legacy capsule function ``torch.utils.dlpack.to_dlpack`` reads the storage of its argument
  in C++ without either hook: this module wraps it (below) so that a record is computed first.  A C++ extension that takes the
  ``at::Tensor`` of a record directly (pybind11, not ``torch.ops``) would still find no storage: call ``vfa_amd.materialize`` first.
"""
import os

import torch

LAZY = os.environ.get("VFA_AMD_LAZY", "1") == "1"


def version_of(t):
    """The in-place version counter of a tensor; None for inference tensors (they have none, and raise when asked)."""
    return None if t.is_inference() else t._version


def _is_zero(x):
    return isinstance(x, (int, float)) and not isinstance(x, bool) and x == 0


def _real(x):
    if isinstance(x, DeferredOrtho):
        return x.materialize()
    if isinstance(x, (list, tuple)):
        return type(x)(_real(v) for v in x)
    if isinstance(x, dict):
        return {k: _real(v) for k, v in x.items()}
    return x


class DeferredOrtho(torch.Tensor):
    """Sum of not-yet-computed ``VFA.forward`` results on one grid: a wrapper tensor (no storage) whose every use goes through
    ``__torch_function__``."""

    @staticmethod
    def __new__(cls, terms, grid, crange, shape, device):
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=torch.float32, device=device, requires_grad=False)
        r._terms, r._grid, r._crange, r._value = terms, grid, crange, None
        r._grid_version = version_of(grid)
        return r

    def __init__(self, *a, **k):
        pass

    def materialize(self):
        if self._value is None:
            from . import vfa_op
            if version_of(self._grid) != self._grid_version:
                raise RuntimeError("the grid passed to VFA.forward was modified in place before the (deferred) result was used")
            self._value = vfa_op._materialize(self._terms, self._grid, self._crange)
            self._terms = None
        return self._value

    def _merge(self, other):
        """self + other as a record, or None when the sum must be computed."""
        if _is_zero(other):
            return self
        if (isinstance(other, DeferredOrtho) and other._value is None and self._value is None and other._grid is self._grid
                and other._crange == self._crange and other.shape == self.shape):
            with torch._C.DisableTorchFunctionSubclass():
                return DeferredOrtho(self._terms + other._terms, self._grid, self._crange, tuple(self.shape), self.device)
        return None

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _METADATA:  # shape, dtype, device, ...: the wrapper knows them
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if func in _ADDS and len(args) == 2 and not kwargs:  # sums of records stay records
            a, b = args
            merged = a._merge(b) if isinstance(a, DeferredOrtho) else (b._merge(a) if isinstance(b, DeferredOrtho) else None)
            if merged is not None:
                return merged
        with torch._C.DisableTorchFunctionSubclass():
            return func(*_real(args), **_real(kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # (wrapper subclasses must define it; every use is intercepted one level up, in __torch_function__ -- what still arrives here
        # came in below the Python API: compute and run on the real tensors)
        return func(*_real(args), **_real(kwargs or {}))

    def __repr__(self):
        if self._value is not None:
            return repr(self._value)
        return f"DeferredOrtho({len(self._terms)} VFA.forward results, shape {tuple(self.shape)})"


_T = torch.Tensor
_METADATA = {_T.shape.__get__, _T.dtype.__get__, _T.device.__get__, _T.requires_grad.__get__, _T.ndim.__get__, _T.is_cuda.__get__,
             _T.layout.__get__, _T.grad_fn.__get__, _T.is_leaf.__get__, _T.names.__get__, _T.size, _T.dim, _T.ndimension, _T.__len__,
             _T.is_floating_point, _T.is_complex, _T.numel, _T.nelement}
_ADDS = {_T.add, _T.__add__, _T.__radd__, _T.__iadd__, _T.add_, torch.add}


def _guard_legacy_dlpack():
    """``torch.utils.dlpack.to_dlpack`` (``torch._C._to_dlpack``) exports the storage of its argument without passing through
    ``__torch_function__`` / ``__torch_dispatch__``: on a record it would hand out (and on this torch: crash on) null storage."""
    import torch.utils.dlpack as _dl
    inner = _dl.to_dlpack
    if getattr(inner, "_vfa_amd_guard", False):
        return

    def to_dlpack(tensor, *args, **kwargs):
        return inner(materialize(tensor), *args, **kwargs)

    to_dlpack.__doc__ = inner.__doc__
    to_dlpack._vfa_amd_guard = True
    _dl.to_dlpack = to_dlpack
    if getattr(torch, "to_dlpack", None) is inner:
        torch.to_dlpack = to_dlpack


def materialize(x):
    """The tensor behind a ``DeferredOrtho`` (computing it if need be); anything else is returned as it is."""
    return x.materialize() if isinstance(x, DeferredOrtho) else x


_guard_legacy_dlpack()
