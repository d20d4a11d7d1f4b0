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
"""
from __future__ import annotations

from typing import Callable

import torch


class CapturedStep:
    def __init__(self, fn: Callable[[], object], warmup: int = 2, device=None, generators=()):
        """``generators``: torch.Generator objects ``fn`` draws from besides the device's default one (e.g.
        ``ShardedMapping.generator(device)``): they are registered with the graph so that every replay draws afresh."""
        self.fn = fn
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                      # warm-up off the capturing stream (allocator, lazy inits)
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        for g in generators:
            self.graph.register_generator_state(g)
        # With a torch.distributed process group alive (ShardedMapping), its watchdog thread polls the events of earlier eager
        # collectives: under the default "global" capture mode that query is an error while this thread captures and the watchdog
        # aborts the process.  "thread_local" checks this thread's calls only; the short sleep lets the watchdog reap what is done.
        mode = "global"
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            import time
            time.sleep(0.5)
            mode = "thread_local"
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.result = fn()                              # static output tensors of the captured iteration

    def __call__(self):
        self.graph.replay()
        return self.result
