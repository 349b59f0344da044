"""Deferred evaluation of the reference's camera loop.

The reference calls the projector once per (camera, scale) and sums in Python (``vfa/model/vfanet.py:64-82``):

    for cam in range(N):
        f8, f16, f32 = self.vfa8(lat8, calib, grid), self.vfa16(lat16, calib, grid), self.vfa32(lat32, calib, grid)
        ortho += f8 + f16 + f32

With the one-line import swap of INTEGRATION.md that loop runs on this build's ``VFA`` modules.  Executed call by call it is
HOST-bound: 21 calls x ~275 us of Python / launch overhead = 5.8 ms per MultiviewC frame, ten times the batched frame, and every call
launches the persistent kernel for a twenty-first of its work.  So in inference (no gradient wanted) ``VFA.forward`` returns a
``DeferredOrtho`` instead of a tensor: a record of (module, feature map, calibration) that knows its shape, adds to other records of the
same grid (``f8 + f16 + f32``, ``ortho += ...``, ``0 + ...``) and turns into the real ``(1, C, L, W)`` tensor the first time anything
else is asked of it -- any ``torch`` function (``self.fuse(ortho)`` is ``F.conv2d``), any tensor method or attribute, indexing,
arithmetic with a real tensor.  By then the record holds the whole frame, and ONE batched launch (``fused_frame`` / ``pipe_frame``:
the path ``aggregate_views`` takes) computes it: the loop costs the batched frame plus bookkeeping.

Sums are re-associated exactly as ``aggregate_views`` re-associates them (inside the post-GEMM tolerance; pre-GEMM tensors are
bitwise either way).  A feature map modified IN PLACE between the call and the use would be read in its new state: the record
keeps the tensor's version counter and raises if it moved.  ``VFA_AMD_LAZY=0`` switches the deferral off (every call computes at once).
"""
import os

import torch

LAZY = os.environ.get("VFA_AMD_LAZY", "1") == "1"


def _is_zero(x):
    return isinstance(x, (int, float)) and not isinstance(x, bool) and x == 0


class DeferredOrtho:
    """Sum of not-yet-computed ``VFA.forward`` results on one grid.  Not a ``torch.Tensor`` subclass: it takes part in torch's
    ``__torch_function__`` protocol (any torch function that receives it gets the materialised tensor) and forwards everything
    else to that tensor."""

    __slots__ = ("_terms", "_grid", "_crange", "_value", "_shape", "_device")

    def __init__(self, terms, grid, crange, shape, device):
        self._terms, self._grid, self._crange, self._value, self._shape, self._device = terms, grid, crange, None, shape, device

    # ------------------------------------------------------------------ cheap facts that need no computation
    @property
    def shape(self):
        return torch.Size(self._shape)

    @property
    def dtype(self):
        return torch.float32

    @property
    def device(self):
        return self._device

    @property
    def requires_grad(self):
        return False

    @property
    def is_cuda(self):
        return True

    def size(self, dim=None):
        return torch.Size(self._shape) if dim is None else self._shape[dim]

    def dim(self):
        return len(self._shape)

    # ------------------------------------------------------------------ sums of pending results stay pending
    def _same_frame(self, other):
        return (isinstance(other, DeferredOrtho) and other._value is None and self._value is None and other._grid is self._grid
                and other._crange == self._crange and other._shape == self._shape)

    def __add__(self, other):
        if _is_zero(other):
            return self
        if self._same_frame(other):
            return DeferredOrtho(self._terms + other._terms, self._grid, self._crange, self._shape, self._device)
        return self.materialize() + (other.materialize() if isinstance(other, DeferredOrtho) else other)

    __radd__ = __add__
    __iadd__ = __add__

    # ------------------------------------------------------------------ everything else wants the tensor
    def materialize(self):
        if self._value is None:
            from . import vfa_op
            self._value = vfa_op._materialize(self._terms, self._grid, self._crange)
            self._terms = None
        return self._value

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def real(x):
            if isinstance(x, DeferredOrtho):
                return x.materialize()
            if isinstance(x, (list, tuple)):
                return type(x)(real(v) for v in x)
            return x
        return func(*real(args), **{k: real(v) for k, v in (kwargs or {}).items()})

    def __getattr__(self, name):  # (only reached for names not defined above)
        return getattr(self.materialize(), name)

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __len__(self):
        return self._shape[0]

    def __iter__(self):
        return iter(self.materialize())

    def __repr__(self):
        return repr(self.materialize()) if self._value is not None else f"DeferredOrtho({len(self._terms)} VFA.forward results, shape {tuple(self._shape)})"

    def __neg__(self):
        return -self.materialize()

    def __sub__(self, other):
        return self.materialize() - materialize(other)

    def __rsub__(self, other):
        return materialize(other) - self.materialize()

    def __mul__(self, other):
        return self.materialize() * materialize(other)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self.materialize() / materialize(other)

    def __rtruediv__(self, other):
        return materialize(other) / self.materialize()

    def __matmul__(self, other):
        return self.materialize() @ materialize(other)

    def __eq__(self, other):
        return self.materialize() == materialize(other)

    def __ne__(self, other):
        return self.materialize() != materialize(other)

    def __lt__(self, other):
        return self.materialize() < materialize(other)

    def __gt__(self, other):
        return self.materialize() > materialize(other)

    def __le__(self, other):
        return self.materialize() <= materialize(other)

    def __ge__(self, other):
        return self.materialize() >= materialize(other)

    __hash__ = object.__hash__


def materialize(x):
    """The tensor behind a ``DeferredOrtho`` (computing it if need be); anything else is returned as it is."""
    return x.materialize() if isinstance(x, DeferredOrtho) else x
