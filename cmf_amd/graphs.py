"""HIP-graph capture and replay of a whole log-density evaluation.

The tabular / 2-D configurations (C1, C2) are launch-bound: ~150-250 kernels of a few microseconds each, issued from
Python.  Capturing one ``elbo`` call into a HIP graph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm) removes the host
cost: a replay is one ``hipGraphLaunch``.  Every kernel of the path launches on the current stream, allocates nothing
itself (buffers come from PyTorch's allocator, which serves captures from a private pool) and never synchronises, so the
whole call is capturable; only the Cholesky retry *report* (a device-to-host read) is deferred: the retry kernels are in
the graph, their flags stay on the device in ``head.last_gram.fail``.
"""
import torch

__all__ = ["ElboGraph"]


def _find_heads(module):
    return [m for m in module.modules() if type(m).__name__ in ("NonSquareHeadDensity", "ManifoldFlowHeadDensity")]


class ElboGraph:
    """``g = ElboGraph(density, example_x, **elbo_kwargs); out = g(x)`` with ``x`` of the example's shape.

    ``out`` is the dict ``density.elbo`` returns; its tensors are static buffers owned by the graph (clone them to keep
    values across replays).  The callable must be re-created when the batch shape, the kwargs, the train / eval mode or
    the parameter *storage* changes (in-place parameter updates are picked up: weights are re-packed inside the graph
    only if their version changed at capture time, so call ``refresh()`` after an optimiser step)."""

    def __init__(self, density, example_x, warmup=2, **elbo_kwargs):
        if not example_x.is_cuda:
            raise RuntimeError("ElboGraph needs a GPU tensor (there is no CPU fallback)")
        self.density, self.kwargs = density, dict(elbo_kwargs)
        self._heads = _find_heads(density)
        self._x = example_x.detach().clone()
        self._capture(warmup)

    def _capture(self, warmup):
        saved = [(h, h.check_cholesky) for h in self._heads]
        for h, _ in saved:
            h.check_cholesky = "lazy"                    # no device-to-host read inside the captured region
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(warmup):                  # packs weights, sets kernel attributes, warms the allocator
                    self.density.elbo(self._x.clone(), **self.kwargs)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: another thread's HIP calls (the RCCL watchdog of a process group) do not invalidate the capture
            with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self._in = self._x.clone()               # dequantisation mutates its input: work on a graph-owned copy
                self.out = self.density.elbo(self._in, **self.kwargs)
        finally:
            for h, mode in saved:
                h.check_cholesky = mode

    def refresh(self):
        """Re-capture (e.g. after the optimiser updated the parameters in place)."""
        self._capture(warmup=1)

    def __call__(self, x):
        if x.shape != self._x.shape:
            raise ValueError(f"ElboGraph was captured for shape {tuple(self._x.shape)}, got {tuple(x.shape)}")
        self._x.copy_(x)
        self.graph.replay()
        return self.out

    def cholesky_attempts(self):
        """Read the retry flags of the last replay (synchronises): 1 = first factorisation succeeded."""
        worst = 1
        for h in self._heads:
            if h.last_gram is not None:
                fail = h.last_gram.fail.tolist()
                n = 1
                while n <= h.MAX_ATTEMPTS and fail[n - 1]:
                    n += 1
                worst = max(worst, n)
        return worst
