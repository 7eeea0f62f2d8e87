"""Key-sharded exact top-k emulated on ONE GPU (one host thread + stream per shard, exchanges through a thread barrier as
ShardedToyGraphBase does them with RCCL; the scheme of tests/test_gpu_fullsize.py) against the unsharded call:
  python tools/shard_soak.py G B N D k [seed]      (one shape; exit code 1 on a mismatch)
  python tools/shard_soak.py soak SECONDS [seed]   (random shapes)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K
from ragraph_amd.sharded import shard_bounds

dev = torch.device("cuda:0")


def run_shape(G, B, N, D, k, seed, prior_mode=0):
    """prior_mode: 0 = bound pass; 1 = a speculative prior below every query's k-th best (every row proven); 2 = a prior at
    the 30 % quantile of the k-th best scores (the rows below it are NOT proven: the owner's verdict must name exactly them,
    and every proven row must carry the single-GPU bits)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=g))
    q = torch.randn(B, D, device=dev, generator=g)
    full_s, full_i = K.topk_cosine(q, kn, k)
    bounds = [shard_bounds(N, G, r) for r in range(G)]
    plan_n = max(hi - lo for lo, hi in bounds)
    shards = [kn[lo:hi].contiguous() for lo, hi in bounds]
    copies = [K.keys_to_bf16(s) for s in shards]
    prior = None
    if prior_mode and K.sharded_speculates(B, plan_n, D, k, G):
        kth = full_s[:, k - 1]
        live = kth[full_s[:, 0] != 0] if bool((full_s[:, 0] != 0).any()) else kth
        prior = float(live.min()) - 0.01 if prior_mode == 1 else float(torch.quantile(live.float(), 0.3))
    torch.cuda.synchronize()
    barrier = threading.Barrier(G)
    slots, out, errs = [None] * G, [None] * G, []
    m = min(k, 2 * (-(-k // G)))
    while G * m > 64 and m > -(-k // G):   # (ShardedToyGraphBase._exchange: theta_sharpen selects among at most 64 values)
        m -= 1

    def exchange_for(r):
        def exchange(phase, theta, scores):
            torch.cuda.current_stream().synchronize()
            slots[r] = scores[:, :m].clone()
            torch.cuda.current_stream().synchronize()
            barrier.wait()
            K.theta_sharpen(torch.stack(slots).contiguous(), theta, k)
            torch.cuda.current_stream().synchronize()
            barrier.wait()
        exchange.n_shards = G
        return exchange

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                K.set_filter_prior(prior)          # (thread-local: every shard's thread sets the group's prior)
                try:
                    s, i, over = K.topk_cosine_filtered(q, shards[r], copies[r], k, idx_base=bounds[r][0],
                                                        exchange=exchange_for(r), plan_n=plan_n)
                finally:
                    K.set_filter_prior(None)
                torch.cuda.current_stream().synchronize()
                out[r] = (s, i, int(over))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    ms, mi = K.topk_merge(torch.stack([o[0] for o in out]), torch.stack([o[1] for o in out]))
    if prior is None:
        bad = (mi != full_i).any(dim=1).nonzero().flatten()
        ok = bad.numel() == 0 and torch.equal(ms, full_s)
        return ok, bad[:6].tolist(), [o[2] for o in out]
    # under a prior: the owner's verdict (ragraph_verify_merged_prior_f32) names the rows that are not proven; every other
    # row must be the single-GPU row, and a row IS proven exactly when its true k-th best reaches the prior
    words = K.verify_merged_prior(ms, prior).cpu().tolist()
    zero = (full_s[:, 0] == 0) & (full_s[:, k - 1] == 0)
    proven = zero | (full_s[:, k - 1] >= prior)
    differs = (mi != full_i).any(dim=1) | (ms != full_s).any(dim=1)
    bad = (differs & proven).nonzero().flatten()
    ok = bad.numel() == 0 and int(words[0]) == int((~proven).sum())
    if prior_mode == 1:
        ok = ok and int(words[0]) == 0 and not bool(differs.any())
    return ok, bad[:6].tolist(), [o[2] for o in out] + [f"prior {prior:.4f} missed {int(words[0])} not-proven {int((~proven).sum())}"]


if sys.argv[1] == "soak":
    budget, seed = float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cpu = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=cpu))
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        G = (2, 3, 4, 8)[ri(0, 3)]
        D = (64, 128, 256)[ri(0, 2)]
        k = (1, 3, 5, 10, 32)[ri(0, 4)]
        N = (ri(20000, 100000), ri(100000, 400000), ri(400000, 1000000))[ri(0, 2)]
        B = (ri(1, 64), ri(65, 300), ri(300, 3000), ri(3000, 20000), 256, 257, 2048)[ri(0, 6)]
        if B * N > 2e9:
            B = max(1, int(2e9 // N))
        if not K.filter_helps(B, -(-N // G), D, k):
            continue
        try:
            ok, bad, over = run_shape(G, B, N, D, k, seed + n, prior_mode=ri(0, 2))
        except BaseException as e:  # noqa: BLE001
            print(f"ERROR G={G} B={B} N={N} D={D} k={k} seed={seed + n}: {e}", flush=True)
            sys.exit(1)
        n += 1
        if not ok:
            print(f"MISMATCH G={G} B={B} N={N} D={D} k={k} seed={seed + n - 1} rows={bad} overflow={over}", flush=True)
            sys.exit(1)
    print(f"shard soak ok: {n} shapes in {time.time() - t0:.0f} s", flush=True)
else:
    G, B, N, D, k = [int(x) for x in sys.argv[1:6]]
    seed = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    ok, bad, over = run_shape(G, B, N, D, k, seed)
    print(f"G={G} B={B} N={N} D={D} k={k}: {'ok' if ok else 'MISMATCH rows ' + str(bad)} overflow {over}")
    sys.exit(0 if ok else 1)
