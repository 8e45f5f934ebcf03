"""Segmented hipGraph capture of a training step that may contain collectives.

A step is captured once and replayed.  A collective reached while capturing (dp.collective) either stays INSIDE the capture
(NERFSIG_CAPTURE_COLLECTIVES=1: RCCL's kernels become graph nodes) or ends the running segment: it is remembered as the eager call
that follows that segment in every replay, and the next segment begins in the same memory pool.  One rank: a single segment.
(trainer.GraphedWatermarkLoop carries its own copy of this logic, interleaved with its stream schedule; stage1.GraphedCleanLoop uses
this class.)"""
import gc

import torch

from . import dp


class SegmentedCapture:
    def __init__(self):
        self.segments, self.between = [], []
        self.stream = None

    def capture(self, fn):
        """Capture fn() (no arguments; its tensors are static) on this object's stream.  Returns fn's result (static tensors)."""
        self.segments, self.between = [torch.cuda.CUDAGraph()], []
        open_capture = [True]

        def boundary(coll, last=False):
            self.segments[-1].capture_end()
            self.between.append(coll)
            if last:
                open_capture[0] = False
                return
            g = torch.cuda.CUDAGraph()
            self.segments.append(g)
            g.capture_begin(pool=self.segments[0].pool(), capture_error_mode="thread_local")

        gc.collect()
        torch.cuda.synchronize()
        dp.drain_watchdog()       # (a no-op without an nccl group) nothing of the warm-up's collectives may be left with ProcessGroupNCCL's watchdog thread
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        prev = dp.set_boundary(boundary)
        try:
            with torch.cuda.stream(self.stream):
                self.segments[0].capture_begin(capture_error_mode="thread_local")
                try:
                    out = fn()
                except BaseException:
                    # fn() raised in the middle of a segment: the stream must not be left capturing (every later launch on it would fail with "operation not
                    # permitted when stream is capturing").  End the open segment, drop everything captured so far, and let the error travel.
                    if open_capture[0]:
                        open_capture[0] = False
                        try:
                            self.segments[-1].capture_end()
                        except Exception:      # noqa: BLE001 -- the capture may already be invalidated by the failing call; the original error is the one to report
                            pass
                    self.segments, self.between = [], []
                    raise
                if open_capture[0]:
                    self.segments[-1].capture_end()
        finally:
            dp.set_boundary(prev)
        torch.cuda.current_stream().wait_stream(self.stream)
        return out

    def replay(self):
        for i, g in enumerate(self.segments):
            g.replay()
            if i < len(self.between):
                self.between[i]()        # the collective between two segments (ordered behind the segment on the current stream)
