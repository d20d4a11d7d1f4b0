"""hipGraph capture of a mapping / tracking iteration (convenience around torch.cuda.CUDAGraph, no kernel of its own).

A fused mapping iteration (``mapping_loss`` + ``nice_slam_amd.backward``) is seven launches of 5-100 us (window kernel -- which also
zero-fills the iteration's gradient buffer --, sample placement, decoder passes, compositor, dX, dW, finalize), a tracking iteration
nine of 5-30 us (+ tracking loss, compositor backward, pose gradient; no dW / finalize): launched eagerly from Python the host sets the pace
(~10 us per launch), replayed from a graph the GPU runs them back to back.  Everything this package launches is capturable: no
host synchronisation, no data-dependent shapes, workspaces / packed decoder buffers / gradient blobs / the in-kernel pixel
draw's state at stable addresses.  ``CapturedStep`` wraps the usual torch recipe (warm-up on a side stream, capture, replay):

    step = nice_slam_amd.graphs.CapturedStep(one_iteration)     # one_iteration(): no arguments, static shapes,
    for _ in range(n): step()                                   # reads its inputs from tensors it closes over

What the closure must respect is torch's, not ours: no ``.item()`` / boolean-mask indexing inside (use
``nice_slam_amd.aabb_keep`` or the fused losses instead of the compaction of Mapper.py:471-481), optimisers with
``capturable=True`` (``MaskedGridAdam(capturable=True)`` keeps its step counts on the device), and fresh inputs are written INTO
the closed-over tensors (``t.copy_(new)``) before each replay.  The side-stream warm-up is not optional when a leaf tensor
(a pose) receives its gradient through autograd's AccumulateGrad: a loss tensor of an EAGER iteration that is still alive keeps
that node, the node remembers the eager stream, and the captured backward then drags that stream into the capture (on ROCm 7.2
``hipStreamEndCapture`` dies on it instead of reporting unjoined work; measured, tests/perf/capture_segv_probe.py).

Multi-GPU (``parallel.ShardedMapping``): the iteration has ONE collective.  ``CapturedStep`` captures it with the kernels (RCCL
collectives are capturable); ``SegmentedStep`` is the middle path for when that capture is refused or the backend cannot be captured
(gloo): the kernels before the collective and the kernels behind it are two graphs, the collective an ordinary eager call between
the two replays (``ShardedMapping.split_exchange``) -- one graph boundary instead of ~10 eager launches.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


def default_capture_error_mode() -> str:
    """"global" (torch's default: any thread's unsafe call during the capture is an error) unless a torch.distributed process group
    is alive: its watchdog THREAD polls the events of earlier eager collectives, a query that "global" mode turns into an error
    which aborts the process (seen with one rank over RCCL) -- then "thread_local": only the capturing thread's calls are checked.
    Note what that gives up: an allocation or a synchronisation from ANOTHER thread of the process (a tracker or data-loader thread)
    during the capture is no longer reported.  Pass ``capture_error_mode="global"`` explicitly to keep the check."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return "thread_local"
    return "global"


def _warm_up(fn, n, dev):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):                          # warm-up off the capturing stream (allocator, lazy inits)
        for _ in range(max(1, n)):
            fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)                            # (also: every collective enqueued so far has completed on the device)


class CapturedStep:
    def __init__(self, fn: Callable[[], object], warmup: int = 2, device=None, generators=(), capture_error_mode: Optional[str] = None):
        """``generators``: torch.Generator objects ``fn`` draws from besides the device's default one (e.g.
        ``ShardedMapping.generator(device)``): they are registered with the graph so that every replay draws afresh.
        ``capture_error_mode``: None = ``default_capture_error_mode()``."""
        self.fn = fn
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _warm_up(fn, warmup, dev)
        self.graph = torch.cuda.CUDAGraph()
        for g in generators:
            self.graph.register_generator_state(g)
        self.capture_error_mode = capture_error_mode or default_capture_error_mode()
        with torch.cuda.graph(self.graph, capture_error_mode=self.capture_error_mode):
            self.result = fn()                              # static output tensors of the captured iteration

    def __call__(self):
        self.graph.replay()
        return self.result


class SegmentedStep:
    """``first()`` -> state, then ``between(state)`` EAGER, then ``second(state)``: two graphs that share a memory pool (what
    ``first`` allocates stays valid for ``between`` and ``second``) around one call that stays out of any graph.  For a
    ``ShardedMapping(split_exchange=True)`` iteration:

        seg = SegmentedStep(first=lambda: (backward(sharder.mapping_loss(...)), sharder.deferred())[1],
                            between=sharder.reduce_deferred, second=sharder.scatter_deferred)
        for _ in range(n): seg()

    The capture itself executes ``between`` once (for real) so that ``second`` is recorded against reduced buffers."""

    def __init__(self, first: Callable[[], object], between: Callable[[object], None], second: Callable[[object], object],
                 warmup: int = 2, device=None, generators=(), capture_error_mode: Optional[str] = None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.between = between

        def whole():
            st = first()
            between(st)
            return second(st)
        _warm_up(whole, warmup, dev)
        self.capture_error_mode = capture_error_mode or default_capture_error_mode()
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        for g in generators:
            self.graph_a.register_generator_state(g)
        with torch.cuda.graph(self.graph_a, capture_error_mode=self.capture_error_mode):
            self.state = first()
        between(self.state)
        with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode=self.capture_error_mode):
            self.result = second(self.state)

    def __call__(self):
        self.graph_a.replay()
        self.between(self.state)
        self.graph_b.replay()
        return self.result
