"""Randomised soak of the bf16-filtered exact top-k against the fp32 kernels (both on the GPU; the fp32 kernels are
the oracle-checked ones): random shapes, banks with duplicates / clusters / tiny norms, random idx_base.
  python tools/soak_filtered.py [seconds] [seed] [filtered|index|prior|spec]
"index": through KeyIndex.topk -- the product dispatch (fused small-bank kernel, direct / ring filter, fp32 kernels).
"prior": topk_cosine_filtered under a FORCED speculative first bound drawn around the batch's true k-th best scores (below all
of them, among them, above all of them).  "spec": KeyIndex over several calls with fresh queries, synchronised in between, so
that the index derives, uses and withdraws its own prior."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mode = sys.argv[3] if len(sys.argv) > 3 else "filtered"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(seed)
cpu = torch.Generator().manual_seed(seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=cpu))


t0, n, n_over_total, kinds = time.time(), 0, 0, {}
while time.time() - t0 < budget:
    D = (64, 128, 256)[ri(0, 2)]
    k = (1, 2, 3, 5, 10, 17, 32)[ri(0, 6)]
    N = (ri(8192, 40000), ri(40000, 200000), ri(200000, 600000), 65536, 16384, 8192, 32768, 32767, 65535, ri(32768, 70000))[ri(0, 9)]
    B = (ri(1, 40), ri(40, 300), ri(300, 3000), ri(3000, 20000), 256, 257, 512, 16384, 16385, 64, 65, 2047, 2048,
         ri(2048, 6000))[ri(0, 13)]
    if B * N > 3e9:
        B = max(1, int(3e9 // N))
    kind = ri(0, 5)
    keys = torch.randn(N, D, device=dev, generator=g)
    if kind == 1:    # clustered bank: 50 centres + noise (dense top of the score distribution)
        c = torch.randn(50, D, device=dev, generator=g)
        keys = c[torch.randint(0, 50, (N,), device=dev, generator=g)] + 0.35 * keys
    elif kind == 2:  # exact duplicates (ties at the k-th place)
        m = min(N // 3, 500)
        keys[N - m:] = keys[:m]
    elif kind == 3:  # a block of near-duplicates of one row
        m = min(N // 4, 1500)
        keys[N // 2:N // 2 + m] = keys[7] + 1e-3 * torch.randn(m, D, device=dev, generator=g)
    elif kind == 5:  # heavy-tailed rows (one dominant entry): a few, a run of them, or every 97th -- the int8 copy's HEAVY granules
        m = (1, 7, ri(2, 2000), N // 97)[ri(0, 3)]
        rows = torch.randint(0, N, (m,), device=dev, generator=g) if ri(0, 1) else (torch.arange(m, device=dev) * 97 + ri(0, 96)) % N
        keys[rows] = 0.03 * keys[rows]
        keys[rows, torch.randint(0, D, (rows.numel(),), device=dev, generator=g)] = 1.0
    kn = K.normalize_rows(keys)
    q = torch.randn(B, D, device=dev, generator=g)
    if kind == 5 and B > 4:   # some queries next to heavy rows: heavy keys among the winners
        q[3] = keys[rows[0]] + 0.05 * q[3]
    if kind == 1:
        q = kn[torch.randint(0, N, (B,), device=dev, generator=g)] + 0.2 * q
    if kind == 3 and B > 2:
        q[1] = keys[7]
    if kind == 4 and B > 3:
        q[2] = 0.0  # zero query: every key ties
    base = (0, 5, 1_000_000)[ri(0, 2)]
    if mode == "index":
        idx = K.KeyIndex(kn)
        s1, i1 = idx.topk(q, k, idx_base=base)
        if ri(0, 1):  # a second call on the same index (cached copies, the overflow feedback of the first)
            s1, i1 = idx.topk(q, k, idx_base=base)
        over = 0
    elif mode == "spec":
        idx = K.KeyIndex(kn)
        over = 0
        for c in range(ri(3, 6)):   # earlier calls with other queries of the same kind feed the statistics
            qq = torch.randn(B, D, device=dev, generator=g)
            if kind == 1:
                qq = kn[torch.randint(0, N, (B,), device=dev, generator=g)] + 0.2 * qq
            if ri(0, 3) == 0:
                qq = qq * torch.linspace(0.2, 1.0, D, device=dev)   # a batch from another distribution now and then
            sq, iq = idx.topk(qq, k, idx_base=base)
            torch.cuda.synchronize()
            s0q, i0q = K.topk_cosine(qq, kn, k, idx_base=base)
            if not (torch.equal(iq, i0q) and torch.equal(sq, s0q)):
                print(f"MISMATCH (spec, call {c}) B={B} N={N} D={D} k={k} kind={kind} prior={idx.search_index.last_prior}", flush=True)
                sys.exit(1)
        s1, i1 = idx.topk(q, k, idx_base=base)
        kinds["spec_used"] = kinds.get("spec_used", 0) + (1 if idx.search_index.last_prior is not None else 0)
    elif mode == "prior":
        if not K.filter_helps(B, N, D, k):
            continue
        s0, i0 = K.topk_cosine(q, kn, k, idx_base=base)
        kth = s0[:, k - 1]
        lo, hi = float(kth.min()), float(kth.max())
        prior = (lo - 0.02, lo - 0.2, 0.5 * (lo + hi), hi + 0.02, lo + 1e-4)[ri(0, 4)]
        if B > 300 and prior > lo:   # (every miss is an exact scan: keep the forced-miss cases small)
            prior = lo - 0.01
        K.set_filter_prior(prior)
        try:
            s1, i1, over = K.topk_cosine_filtered(q, kn, K.keys_to_bf16(kn), k, idx_base=base)
        finally:
            K.set_filter_prior(None)
    else:
        s1, i1, over = K.topk_cosine_filtered(q, kn, K.keys_to_bf16(kn), k, idx_base=base)
    s0, i0 = K.topk_cosine(q, kn, k, idx_base=base)
    ok = torch.equal(i0, i1) and torch.equal(s0, s1)
    n += 1
    n_over_total += int(over)
    kinds[kind] = kinds.get(kind, 0) + 1
    if not ok:
        bad = (i0 != i1).any(dim=1).nonzero().flatten()[:5].tolist()
        print(f"MISMATCH B={B} N={N} D={D} k={k} kind={kind} base={base} overflow={over} rows={bad}", flush=True)
        sys.exit(1)
print(f"soak ok: {n} shapes in {time.time() - t0:.0f} s (kinds {kinds}), {n_over_total} overflow rows recomputed", flush=True)
