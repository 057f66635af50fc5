"""A device loop without the host in it: the enqueue-only sequence of one iteration is captured ONCE in a hipGraph and
replayed; what an iteration reports is one row of doubles that is copied into a device ring behind every replay and comes
to the host when somebody looks.

This is how the reference-shaped drivers run at the speed of the benchmark: `FieldTransformation.run` (fthmc/ft_hmc.py:272-346)
and `qed_helpers.ft_run` (ipynb/ft_hmc.py:437-487) are Python loops that synchronise with the device several times per
trajectory; here a trajectory costs the host one graph launch and one small copy.  (`train.GraphTrainer` is the same idea
for the training loop.)
"""
from __future__ import annotations

import contextlib
import gc
from typing import Callable, List, Optional

import numpy as np
import torch


@contextlib.contextmanager
def capture(graph: "torch.cuda.CUDAGraph", stream: "torch.cuda.Stream"):
    """`torch.cuda.graph(graph, stream=stream, capture_error_mode='thread_local')` with Python's cyclic garbage collector held
    off until the capture has ended.  An automatic collection that starts in the capturing thread may finalise an object left
    over from earlier work -- an older captured graph with its memory pool, a stream, an event -- whose destructor calls into the
    HIP runtime (hipGraphExecDestroy, hipFree, ...): not permitted while this thread captures, and an error inside a destructor
    ends the process (seen once: `Fatal Python error: Aborted`, `Garbage-collecting`, under the `enqueue` of a re-capture).
    torch.cuda.graph collects once on entry; what becomes garbage during the capture waits for its end.  thread_local: another
    thread (a process group's watchdog) may go on calling the runtime."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream, capture_error_mode='thread_local'):
            yield
    finally:
        if was_enabled:
            gc.enable()


class GraphLoop:
    """`enqueue()` must only enqueue device work on the current stream (no host synchronisation, no allocation that
    changes between iterations) and leave the iteration's results in `row` (a contiguous float64 device tensor).

    The first `step()` runs the sequence eagerly (allocator, workspaces), then captures it -- the capture does not execute,
    so n calls of `step()` are exactly n iterations, with the same random draws as n eager ones (torch's generator takes
    part in the capture).  `stream`: the stream the loop runs on (default: a new one); an owner that re-captures often hands
    in the same one every time."""

    def __init__(self, enqueue: Callable[[], None], row: torch.Tensor, chunk: int = 256, use_graph: bool = True,
                 stream: Optional[torch.cuda.Stream] = None):
        self.enqueue, self.row = enqueue, row
        self.dev = row.device
        self.chunk = max(1, int(chunk))
        self.ring = torch.empty(self.chunk, row.numel(), dtype=row.dtype, device=self.dev)
        self.flushed: List[np.ndarray] = []
        self.n = 0
        self.graph = None
        self.use_graph = bool(use_graph)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.dev)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))

    @property
    def captured(self) -> bool:
        return self.graph is not None

    def step(self):
        with torch.cuda.stream(self.stream):
            if self.use_graph and self.graph is None:
                self.enqueue()
                self.stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with capture(g, self.stream):
                    self.enqueue()
                self.graph = g
            elif self.use_graph:
                self.graph.replay()
            else:
                self.enqueue()
            self.ring[self.n % self.chunk].copy_(self.row.reshape(-1))
            self.n += 1
            if self.n % self.chunk == 0:
                self.flushed.append(self.ring.cpu().numpy())

    def last(self) -> np.ndarray:
        """the last iteration's row on the host (synchronises)"""
        self.stream.synchronize()
        return self.row.detach().reshape(-1).cpu().numpy()

    def rows(self) -> np.ndarray:
        """[iterations, row] of every iteration so far, on the host (synchronises)"""
        self.stream.synchronize()
        chunks = list(self.flushed)
        k = self.n % self.chunk
        if k:
            chunks.append(self.ring[:k].cpu().numpy())
        if not chunks:
            return np.zeros((0, self.row.numel()))
        return np.concatenate(chunks, axis=0)

    def reset_history(self):
        self.stream.synchronize()
        self.flushed, self.n = [], 0

    def join(self):
        """the caller's current stream waits for everything enqueued so far"""
        torch.cuda.current_stream(self.dev).wait_stream(self.stream)
