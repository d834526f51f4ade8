"""
HIP-graph capture of an evaluation (launch-bound shapes: few series or short chains).

Every entry point of the C ABI only enqueues kernels on the stream it is given and allocates nothing, and the Python layer
allocates outputs and workspaces through torch's caching allocator, so a whole ``log_likelihood()`` / ``cholesky`` /
``solve`` call can be captured once with ``torch.cuda.CUDAGraph`` (hipGraph on ROCm) and replayed: the 6-13 launches of a
call then cost one graph launch instead of one dispatch (+ Python) each.  Inputs are read from the tensors the captured call
used: update them IN PLACE (``tensor.copy_(new)``) between replays.
"""
from typing import Any, Callable

import torch


class CapturedCall:
    """``CapturedCall(fn)`` runs ``fn`` a few times, captures one more run into a graph; calling the object replays it and
    returns the (static) output tensors of the captured run."""

    def __init__(self, fn: Callable[[], Any], warmup: int = 2):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.output = fn()

    def __call__(self):
        self.graph.replay()
        return self.output


def capture(fn: Callable[[], Any], warmup: int = 2) -> CapturedCall:
    return CapturedCall(fn, warmup)
