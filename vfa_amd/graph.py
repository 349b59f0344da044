"""hipGraph capture of the aggregate for launch-bound (small-grid) deployments.

One frame of the path is 13 kernel launches plus PyTorch dispatch; on small grids (a few thousand cells) their
launch latency, not the GPU, bounds the frame rate.  The C-ABI entry points are stream-ordered, allocate nothing and
take no host round trip, so the whole camera loop can be captured once into a hipGraph (``torch.cuda.CUDAGraph`` is
hipGraph on ROCm) and replayed per frame on static input buffers.  Inference only (no autograd inside a capture).
"""
import torch

from . import _lib
from .aggregate import aggregate_views
from .vfa_op import owned_capture_states


class GraphedAggregate:
    """Replays ``aggregate_views`` for fixed shapes.

    >>> g = GraphedAggregate(vfa8, vfa16, vfa32, lat8, lat16, lat32, calibs, grid)   # example tensors fix the shapes
    >>> ortho = g(lat8_new, lat16_new, lat32_new, calibs_new)                         # copies in, replays, returns view
    The returned tensor is the graph's static output buffer: consume or clone it before the next call.
    """

    def __init__(self, vfa8, vfa16, vfa32, lat8, lat16, lat32, calibs, grid, crange=(-1, 0.95), warmup=2):
        self.mods = (vfa8, vfa16, vfa32)
        self.static_in = [t.detach().clone() for t in (lat8, lat16, lat32, calibs)]
        self.grid = grid.detach().clone()
        self.crange = crange
        side = torch.cuda.Stream(device=grid.device)
        side.wait_stream(_lib.current_stream(grid.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):  # allocator / library warm-up outside the capture
                self._run()
        _lib.current_stream(grid.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # (the workspaces the captured frame replays into live and die with this object, not with the process)
        with owned_capture_states() as self._states, torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = self._run()

    def _run(self):
        lat8, lat16, lat32, calibs = self.static_in
        return aggregate_views(*self.mods, lat8, lat16, lat32, calibs, self.grid, self.crange)

    def __call__(self, lat8, lat16, lat32, calibs=None):
        for dst, src in zip(self.static_in, (lat8, lat16, lat32, calibs)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.static_out
