"""Randomised soak of the dense / sparse / row kernels against the CPU oracle (bit for bit; the softmax family to 1e-6): random shapes through every
dispatch of ragraph_linear_f32, ragraph_spmm_csr_f32, gather_reduce, segment_reduce, segment_softmax, topk_rows.
(The oracle is test infrastructure: this tool is a checker, like tests/.)   python tools/soak_ops.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import cref
from ragraph_amd import kernels as K

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ri = lambda lo, hi: int(rng.integers(lo, hi + 1))
t0, counts = time.time(), {}


def fail(what, **kw):
    print(f"MISMATCH {what}: {kw}", flush=True)
    sys.exit(1)


while time.time() - t0 < budget:
    op = ("linear", "spmm", "gather_reduce", "segment_reduce", "segment_softmax", "topk_rows")[ri(0, 5)]
    counts[op] = counts.get(op, 0) + 1
    if op == "linear":
        n = (ri(1, 64), ri(64, 3000), ri(3000, 20000))[ri(0, 2)]
        F = (ri(1, 40), ri(40, 400), ri(400, 3000), 64, 128, 256, 1433)[ri(0, 6)]
        D = (ri(1, 20), ri(20, 300), 64, 128, 256, 7)[ri(0, 5)]
        if n * F * D > 4e9:
            n = max(1, int(4e9 // (F * D)))
        X = rng.standard_normal((n, F), dtype=np.float32)
        if ri(0, 3) == 0:
            X *= (rng.random((n, F)) < 0.02)
        W = rng.standard_normal((D, F), dtype=np.float32)
        b = rng.standard_normal(D, dtype=np.float32) if ri(0, 1) else None
        act = (K.ACT_NONE, K.ACT_LEAKY)[ri(0, 1)]
        got = K.linear(T(X), T(W), None if b is None else T(b), act=act, alpha=0.2).cpu().numpy()
        if not np.array_equal(got, cref.linear(X, W, b, act, 0.2)):
            fail(op, n=n, F=F, D=D, bias=b is not None, act=act)
    elif op == "spmm":
        n = (ri(1, 64), ri(64, 5000), ri(5000, 60000))[ri(0, 2)]
        m = (ri(1, 64), ri(64, 5000), n)[ri(0, 2)]
        D = (4 * ri(1, 5), 4 * ri(5, 75), 32, 64, 128, 256)[ri(0, 5)]   # (the entry takes multiples of 4)
        deg = (ri(0, 3), ri(3, 40), ri(40, 300))[ri(0, 2)]
        counts_r = rng.poisson(deg, n).astype(np.int64)
        if ri(0, 4) == 0:
            counts_r[ri(0, n - 1)] = ri(4000, 9000)   # a hub row (blocked summation beyond 4096 entries)
        if counts_r.sum() * D > 3e8:
            counts_r = np.minimum(counts_r, 8)
        rowptr = np.concatenate([[0], np.cumsum(counts_r)]).astype(np.int64)
        nnz = int(rowptr[-1])
        col = rng.integers(0, m, nnz).astype(np.int32)
        val = rng.standard_normal(nnz).astype(np.float32)
        X = rng.standard_normal((m, D), dtype=np.float32)
        b = rng.standard_normal(D, dtype=np.float32) if ri(0, 1) else None
        act = (K.ACT_NONE, K.ACT_PRELU, K.ACT_RELU)[ri(0, 2)]
        yin = rng.standard_normal((n, D), dtype=np.float32) if ri(0, 2) == 0 else None
        got = K.spmm_csr(T(rowptr), T(col), T(val), T(X), bias=None if b is None else T(b), act=act, alpha=0.25,
                         beta=0.5 if yin is not None else 0.0, y_in=None if yin is None else T(yin),
                         long_rows=bool(counts_r.max() > K.ROW_BLOCK)).cpu().numpy()
        ref = cref.spmm_csr(rowptr, col, val, X, b, act, 0.25, 0.5 if yin is not None else 0.0, yin)
        if not np.array_equal(got, ref):
            fail(op, n=n, m=m, D=D, nnz=nnz, act=act, hub=int(counts_r.max()))
    elif op == "gather_reduce":
        N, Dv, C = ri(10, 50000), (ri(1, 300), 64, 128, 256)[ri(0, 3)], ri(1, 12)
        B, k = ri(1, 5000), ri(1, 32)
        V = rng.standard_normal((N, Dv), dtype=np.float32)
        Lb = rng.random((N, C), dtype=np.float32)
        idx = rng.integers(5, N + 5, (B, k)).astype(np.int64)
        gv, gl = K.gather_reduce(T(V), T(Lb), T(idx), idx_base=5)
        rv, rl = cref.gather_reduce(V, Lb, idx, idx_base=5)
        if not (np.array_equal(gv.cpu().numpy(), rv) and np.array_equal(gl.cpu().numpy(), rl)):
            fail(op, N=N, Dv=Dv, C=C, B=B, k=k)
    elif op == "segment_reduce":
        G, D = ri(1, 300), (ri(1, 300), 64, 128, 256)[ri(0, 3)]
        sizes = rng.integers(0, 200, G)
        seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        X = rng.standard_normal((int(seg[-1]), D), dtype=np.float32)
        mean = bool(ri(0, 1))
        got = K.segment_reduce(T(X), T(seg), mean_mode=mean).cpu().numpy()
        if not np.array_equal(got, cref.segment_reduce(X, seg, None, mean), equal_nan=True):
            fail(op, G=G, D=D, mean=mean)
    elif op == "segment_softmax":
        n = ri(1, 3000)
        sizes = rng.integers(0, 60, n)
        if ri(0, 3) == 0:
            sizes[ri(0, n - 1)] = ri(4000, 9000)
        rowptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        x = (3 * rng.standard_normal(int(rowptr[-1]))).astype(np.float32)
        got = K.segment_softmax(T(rowptr), T(x), long_rows=bool(sizes.max() > K.ROW_BLOCK)).cpu().numpy()
        if not np.allclose(got, cref.segment_softmax(rowptr, x), rtol=0, atol=1e-6):   # (expf: another libm -- the contract's 1e-6)
            fail(op, n=n, longest=int(sizes.max()))
    else:
        B, N = ri(1, 2000), ri(1, 30000)
        k = ri(1, min(32, N))
        S = rng.standard_normal((B, N), dtype=np.float32)
        if ri(0, 2) == 0:
            S = np.round(S, 1)   # ties
        gs, gi = K.topk_rows(T(S), k)
        rs, ri_ = cref.topk_rows(S, k)
        if not (np.array_equal(gi.cpu().numpy(), ri_) and np.array_equal(gs.cpu().numpy(), rs)):
            fail(op, B=B, N=N, k=k)
print(f"ops soak ok: {sum(counts.values())} cases in {time.time() - t0:.0f} s {counts}", flush=True)
