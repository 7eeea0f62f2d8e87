"""The tiled kernel on graphs of the same size and degree whose columns are (a) uniformly random (c2-like) and (b) confined
to a window of w rows around the destination row: what the kernel can do when every gather hits its L2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K
from ragraph_amd.graph import CSRGraph
dev = torch.device("cuda:0")
n, D, deg = 100_000, 256, 11
g0 = torch.Generator(device=dev).manual_seed(1)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
H = torch.randn(n, D, device=dev)
for w in (0, 200, 2000, 20000):
    rows = torch.arange(n, device=dev).repeat_interleave(deg)
    if w == 0:
        cols = torch.randint(0, n, (n * deg,), device=dev, generator=g0)
    else:
        cols = (rows + torch.randint(-w, w + 1, (n * deg,), device=dev, generator=g0)).clamp(0, n - 1)
    g, _ = CSRGraph.from_coo(rows, cols, torch.rand(n * deg, device=dev), n, sort_cols=True)
    plan = g.tile_plan(D // 32)
    v2 = g.tiled_values(plan, g.val)
    a = t(lambda: K.spmm_csr_panels(g.rowptr, g.col, g.val, H, True, True))
    b = t(lambda: K.spmm_csr_tiled(plan, v2, H, n, True, True))
    print(f"window {w if w else 'random':>7}: panel kernel {a:7.1f} us, tiled kernel {b:7.1f} us (S {plan.S}, passes {plan.passes})")
