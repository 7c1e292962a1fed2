"""hipGraph capture of a launch-bound search step.

One fwd+bwd of the fusion hypernet is ~60 short kernels; eager Python issues them at
~2 ms/step while the GPU needs a fraction of that.  GraphedStep captures the whole step
(our HIP kernels are launched on torch's current stream, so they land in the capture like
any aten op) and replays it with one hipGraphLaunch.  Dropout stays fresh across replays:
the kernels add a DEVICE counter to their Philox offsets and the graph advances it.
"""
import torch

from . import cell as K


class GraphedStep:
    """fn() must read its inputs from static tensors, set ``.grad = None`` on everything it
    differentiates (so the backward writes instead of accumulating) and return tensors that
    stay referenced (they become static graph outputs)."""

    def __init__(self, fn, warmup=3):
        dev = torch.cuda.current_device()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.counter = torch.zeros(1, dtype=torch.int64, device=f'cuda:{dev}')
        saved = (K.DROP.offset, K.DROP.device_counter)
        K.DROP.offset, K.DROP.device_counter = 0, self.counter
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph):
                self.outputs = fn()
                self.span = K.DROP.offset
                if self.span > 0:
                    self.counter.add_(self.span)      # next replay draws new dropout masks
        finally:
            K.DROP.offset, K.DROP.device_counter = saved
        # keep later eager calls clear of the offsets the graph will use
        K.DROP.offset += 1 << 40

    def replay(self):
        self.graph.replay()
        return self.outputs
