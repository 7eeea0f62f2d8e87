"""HIP-graph replay of an inference forward with fixed shapes.

The reference runs its small configurations (Cora: one 2708-node graph; PROTEINS: batch_size = 1, finetune-rag.py:27) as
a dozen eager launches per forward, and so does this package's eager path: at that size the forward is bound by the
host's launch rate (~0.14 ms of Python + HIP launches for ~0.07 ms of GPU work on c3), not by any kernel.  Every entry of
the C ABI allocates nothing, never synchronises and reads nothing back (the bf16-filtered retrieval repairs overflowed
rows on the device), so the whole forward can be captured ONCE into a HIP graph and replayed:

    fwd = CapturedForward(lambda x: model(x, adj), features)      # capture (model.eval(), fixed shapes, bank built)
    out = fwd(new_features)                                       # copy into the static input + one graph launch

The result tensor is the graph's static output buffer (clone it to keep it across calls).  Anything the forward reads
besides the captured inputs -- the adjacency, the bank, the weights -- is read from the same device addresses at every
replay: update those tensors in place, and re-capture after a bank grows (add_resources reallocates).
"""
from __future__ import annotations

import torch


class CapturedForward:
    def __init__(self, fn, *example_inputs: torch.Tensor, warmup: int = 2):
        if not example_inputs or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise ValueError("CapturedForward: the example inputs must be ROCm device tensors")
        self._fn = fn
        self.static_inputs = [t.clone() for t in example_inputs]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.no_grad():
            with torch.cuda.stream(side):
                for _ in range(max(warmup, 1)):   # workspaces, LDS attributes, caches: everything lazy happens here
                    fn(*self.static_inputs)
            cur.wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_output = fn(*self.static_inputs)

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self.static_inputs):
            raise ValueError(f"CapturedForward: expected {len(self.static_inputs)} inputs, got {len(inputs)}")
        for dst, src in zip(self.static_inputs, inputs):
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise ValueError(f"CapturedForward: input of shape {tuple(src.shape)} / {src.dtype}, captured with "
                                 f"{tuple(dst.shape)} / {dst.dtype} (re-capture for a new shape)")
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_output
