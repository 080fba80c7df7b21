"""A training step as a CHAIN of hipGraphs with the RCCL collectives launched eagerly between them.

One GPU replays the whole step as one hipGraph (bench.py).  With collectives in the step that either means capturing
RCCL inside the graph -- which nothing here can verify on more than one rank -- or launching ~60 kernels per step from
Python, ~1.2-1.6 ms of host work against ~0.8-1.1 ms of GPU work per rank: the ranks would be host-bound.  ``SegmentedGraph``
keeps RCCL out of stream capture and still removes the per-kernel host cost: while the step function runs ONCE under
``capture()``, every collective call site (distributed.start_collective) ends the hipGraph being captured, issues the
collective eagerly, records it, and begins the next hipGraph; ``wait()`` on its handle does the same.  ``replay()`` then
walks the recorded program:

    graph 0 -> start all-gather (RCCL stream) -> graph 1 (runs beside it) -> wait -> graph 2 -> ... -> all-reduce -> graph n

i.e. ~10 graph launches + the collectives per step.  All segments allocate from ONE graph memory pool and are replayed in
capture order, so a tensor produced in one segment is at the same address when a later segment or a collective reads it.
Capture mode is "relaxed": autograd runs the backward on its own device thread, so a segment begun on the caller's thread
may be ended from there.
"""
import gc
import warnings

import torch


class _Handle:
    def __init__(self, owner, idx, work):
        self.owner, self.idx, self.work = owner, idx, work

    def wait(self):
        self.owner._wait(self.idx, self.work)


class SegmentedGraph:
    def __init__(self):
        self.actions = []          # ('graph', CUDAGraph) | ('call', fn) | ('wait', index of the call)
        self.pool = None
        self._g = None
        self.recording = False

    # -- recording -----------------------------------------------------------------------------------------------
    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._g.capture_begin(pool=self.pool, capture_error_mode='relaxed')

    def _end(self):
        # a segment with no kernels in it (a wait() directly behind its collective, two waits in a row) is dropped:
        # torch reports it with a warning when the capture ends
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            self._g.capture_end()
        empty = any('Graph is empty' in str(w.message) for w in caught)
        for w in caught:
            if 'Graph is empty' not in str(w.message):
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if not empty:
            self.actions.append(('graph', self._g))
        self._g = None

    def collective(self, start_fn):
        """Called (through distributed.start_collective) where the step starts a collective: ``start_fn()`` issues it and
        returns an object with wait() (or None for a blocking call)."""
        self._end()
        work = start_fn()
        idx = len(self.actions)
        self.actions.append(('call', start_fn))
        self._begin()
        return _Handle(self, idx, work)

    def _wait(self, idx, work):
        self._end()
        if work is not None:
            work.wait()
        self.actions.append(('wait', idx))
        self._begin()

    def capture(self, fn, stream=None):
        """Run ``fn()`` once on a side stream, recording it as segments.  Returns fn's result (static tensors).
        ``stream``: the side stream the step was WARMED UP on.  autograd pins every AccumulateGrad node to the stream of
        the parameter's first use; a capture on another stream makes the engine synchronise the two, the warm-up stream
        joins the capture, and a segment that ends in the middle of the backward pass then ends with unjoined work."""
        from . import distributed as gdist
        if gdist.RECORDER is not None:
            raise RuntimeError('a SegmentedGraph capture is already running')
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        self.pool = torch.cuda.graph_pool_handle()
        side = stream if stream is not None else torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        gdist.RECORDER = self
        self.recording = True
        try:
            with torch.cuda.stream(side):
                self._begin()
                try:
                    out = fn()
                finally:
                    if self._g is not None:
                        self._end()
        finally:
            gdist.RECORDER = None
            self.recording = False
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return out

    # -- replay --------------------------------------------------------------------------------------------------
    def replay(self):
        works = {}
        for i, (kind, x) in enumerate(self.actions):
            if kind == 'graph':
                x.replay()
            elif kind == 'call':
                works[i] = x()
            else:
                w = works.pop(x)
                if w is not None:
                    w.wait()

    def describe(self):
        n_graph = sum(1 for k, _ in self.actions if k == 'graph')
        n_call = sum(1 for k, _ in self.actions if k == 'call')
        return f'{n_graph} hipGraph segments + {n_call} eager collectives'
