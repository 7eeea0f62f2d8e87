"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/ragraph_hip.h declares; the
product path refuses to run without a device (no silent fallback); host-side bookkeeping (shard bounds, data
stand-ins, split planner contract) behaves."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "ragraph_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ragraph_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    from ragraph_amd import _native as N

    if not os.path.exists(N.SO_PATH):
        N.build()
    lib = ctypes.CDLL(N.SO_PATH)
    names = _header_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ragraph_hip.h but not exported"
    assert sorted(N.SIGNATURES) == names, "the ctypes table and the header must list the same entry points"
    assert N.lib().ragraph_abi_version() == 1


def test_pure_host_entry_points_work_without_gpu():
    from ragraph_amd import _native as N

    L = N.lib()
    ws = L.ragraph_topk_cosine_workspace_bytes(4096, 1_000_000, 256, 10)
    assert ws >= 4096 * 256 * 4
    # any row width (the reference takes every emb_size): widths without a fused kernel take score slabs
    assert L.ragraph_topk_cosine_workspace_bytes(10, 10, 100, 3) >= 10 * 100 * 4 + 10 * 10 * 4
    assert L.ragraph_topk_cosine_workspace_bytes(10, 10, 0, 3) == 0    # D < 1 -> 0
    # banks longer than one dense launch's columns: key chunks + per-chunk lists
    assert L.ragraph_topk_cosine_workspace_bytes(64, 9_000_000, 300, 10) >= 64 * 3_000_000 * 4 + 3 * 64 * 10 * 12
    # argument validation happens before any device work, so it is observable here
    rc = L.ragraph_topk_cosine_f32(None, 1, None, 1, 256, 1, 0, None, None, None, 0, None)
    assert rc == N.EINVAL and b"null" in L.ragraph_last_error()


def test_bank_copy_buffer_holds_the_int8_granule_table():
    """ragraph_keys_bf16_rows(N) does not know D, yet the buffer it sizes (rows of 2 D bytes) must hold, behind the int8 rows and
    their tail row, the granules' maxima (floats, padded to 16 bytes), one class bit per granule (whole words) and still the
    16 KB the last int8 stage of a D = 64 level may read past the rows (csrc/filter_common.h: FilterI8View)."""
    from ragraph_amd import _native as N

    L = N.lib()
    for n in (1, 2, 255, 256, 257, 511, 513, 4096, 65535, 65537, 1_000_000, 4_000_003, 100_000_000, 2_000_000_000):
        rows = L.ragraph_keys_bf16_rows(n)
        npad = -(-n // 256) * 256
        for D in (64, 128, 256):
            gk = 32768 // D
            granules = -(-npad // gk)
            table = -(-granules * 4 // 16) * 16 + -(-granules // 32) * 4 + 16
            used = (npad + 1 + npad // 2 + 1) * 2 * D + table
            assert rows * 2 * D >= used + 16384, (n, D, rows)


def test_filtered_topk_plan_is_well_formed():
    """Schedule of the bf16-filtered exact top-k (host arithmetic, include/ragraph_hip.h): for every shape the levels
    partition [0, N) in increasing multiples of 256 that start behind the exact sample, a level's expected candidates
    (1.3 k x its end / the previous end) stay inside the per-query list, and the workspace covers a slab level 0."""
    from ragraph_amd import _native as N

    L = N.lib()
    plan = (ctypes.c_int64 * 7)()
    cap = L.ragraph_topk_cosine_filtered_cap(10)
    seen_levels = set()
    for B in (1, 12, 40, 256, 257, 512, 1024, 4096, 16384, 16385, 100_000):
        for Nk in (4096, 16384, 65536, 70_003, 1_000_000, 4_000_000, 8_000_000, 100_000_000):
            for D in (64, 256):
                for k in (1, 10, 32):
                    nlev = L.ragraph_topk_cosine_filtered_plan(B, Nk, D, k, plan)
                    n0, mode, nl, bound_keys = plan[0], plan[1], plan[2], plan[6]
                    slab0 = mode == 1
                    ends = [plan[3 + i] for i in range(nl)]
                    assert nlev == nl and 1 <= nl <= 3, (B, Nk, D, k)
                    assert k <= n0 <= Nk
                    assert ends[-1] == Nk and all(e % 256 == 0 for e in ends[:-1])
                    if D == 64:  # a stage of the int8 copy holds 512 keys there: a level starts at a whole stage
                        assert all(e % 512 == 0 for e in ends[:-1]), (B, Nk, k, ends)
                    assert all(a < b for a, b in zip(ends, ends[1:])) and (nl == 1 or n0 < ends[0])
                    if Nk >= 65536:  # the shapes KeyIndex sends here
                        prev = n0
                        for e in ends:
                            assert 1.3 * k * e / prev <= cap, (B, Nk, D, k, n0, ends)
                            prev = e
                    assert (bound_keys > 0) == (mode == 2) and (mode != 2 or Nk >= 8192)
                    if mode == 2:  # the k keys behind the bound lie inside the first level; a stage or more per part
                        assert bound_keys % 256 == 0 and bound_keys <= ends[0]
                        assert bound_keys // (32768 // (2 * D)) >= k
                    if slab0:
                        assert B <= 16384
                        assert L.ragraph_topk_cosine_filtered_workspace_bytes(B, Nk, D, k) >= B * n0 * 4 + B * cap * 4
                    seen_levels.add(nl)
    assert seen_levels == {1, 2, 3}
    assert L.ragraph_topk_cosine_filtered_plan(100, 1000, 100, 3, plan) < 0  # unsupported D
    # the bench shape keeps the three-level schedule the measurements in DESIGN.md describe
    assert L.ragraph_topk_cosine_filtered_plan(100_000, 1_000_000, 256, 10, plan) == 3
    assert list(plan) == [15625, 2, 3, 62720, 250880, 1_000_000, 18944]


def test_sharded_workspace_covers_both_schedules():
    """The sharded entry plans for (plan_N, n_shards); its workspace size covers that schedule AND the single bank's (a
    NULL exchange runs the latter), for every shard count."""
    from ragraph_amd import _native as N

    L = N.lib()
    for B in (40, 300, 3000, 20000, 100_000):
        for Nk in (30_000, 125_000, 500_000, 1_000_000):
            for D in (64, 256):
                single = L.ragraph_topk_cosine_filtered_workspace_bytes(B, Nk, D, 10)
                for G in (1, 2, 3, 8):
                    sharded = L.ragraph_topk_cosine_filtered_sharded_workspace_bytes(B, Nk, D, 10, G)
                    assert sharded >= single > 0, (B, Nk, D, G)
    assert L.ragraph_topk_cosine_filtered_sharded_workspace_bytes(10, 1000, 100, 3, 2) == 0  # unsupported D


def test_key_index_overflow_policy_on_host():
    """KeyIndex judges a bank by what its overflowed queries cost (kernels_index.py): more than 1/64 of a call of >= 64
    queries (and at least two) takes the levels off int8 first, then the bank off the filter; a handful of queries per
    call are judged cumulatively (a quarter of at least eight); ordinary counts change nothing."""
    import torch

    from ragraph_amd.kernels_index import KeyIndex

    class Done:
        def query(self):
            return True

    def poll(idx, n_over, B, had_i8):
        idx._pending = (torch.tensor([n_over], dtype=torch.int32), Done(), B, had_i8)
        idx._poll_overflow()

    idx = KeyIndex(torch.zeros(4, 8), ops=object())
    poll(idx, 1, 100_000, True)          # one query of a large call: nothing
    poll(idx, 1500, 100_000, True)       # 1.5 %: below 1/64
    assert not idx._i8_off and not idx._filter_off
    poll(idx, 1600, 100_000, True)       # above 1/64 on int8 levels: int8 goes first
    assert idx._i8_off and not idx._filter_off
    poll(idx, 1600, 100_000, False)      # ... and on bf16 levels the filter
    assert idx._filter_off
    idx = KeyIndex(torch.zeros(4, 8), ops=object())
    poll(idx, 1, 64, True)               # a single query never decides
    assert not idx._i8_off
    poll(idx, 2, 64, True)
    assert idx._i8_off and not idx._filter_off
    idx = KeyIndex(torch.zeros(4, 8), ops=object())
    for _ in range(7):
        poll(idx, 1, 1, False)           # one query per forward, every one overflowing: judged over eight of them
    assert not idx._filter_off
    poll(idx, 1, 1, False)
    assert idx._filter_off and idx.overflowed_queries == 8


def test_key_index_collapses_exact_duplicates_on_host():
    """KeyIndex over a bank of duplicates (host logic with oracle-backed kernels on CPU tensors): the search runs over
    the unique rows and the expansion restores the canonical top-k of every row -- for a reference-shaped bank (three
    quarters one vector, ToyGraphBase.py:91-119 + Augmentation.py:9-20), tied groups, a zero query, k above the number of
    unique rows; banks below the thresholds are searched as they are."""
    import numpy as np
    import torch

    from oracle import cref
    from ragraph_amd.kernels_index import KeyIndex

    class Ops:
        searched = []

        @staticmethod
        def topk_cosine(q, kn, k, idx_base=0):
            Ops.searched.append(kn.shape[0])
            s, i = cref.topk_cosine(q.numpy(), kn.numpy(), k, idx_base)
            return torch.from_numpy(s), torch.from_numpy(i)

        @staticmethod
        def gather_rows(v, idx, idx_base=0):
            return torch.from_numpy(cref.gather_rows(v.numpy(), idx.numpy(), idx_base))

        @staticmethod
        def dedup_rows(kn):
            U, largest, uniq, ptr, mem = cref.dedup_rows(kn.numpy())
            return U, largest, torch.from_numpy(uniq), torch.from_numpy(ptr), torch.from_numpy(mem)

        @staticmethod
        def topk_expand_groups(su, iu, ptr, mem, k, idx_base=0, idx_base_u=0):
            s, i = cref.topk_expand_groups(su.numpy(), iu.numpy(), ptr.numpy(), mem.numpy(), k, idx_base, idx_base_u)
            return torch.from_numpy(s), torch.from_numpy(i)

    rng = np.random.default_rng(3)
    N, D = 4000, 32
    real = cref.normalize_rows(rng.standard_normal((N // 4, D), dtype=np.float32))
    real[rng.random(N // 4) < 0.3] = real[7]
    kn = np.tile(cref.normalize_rows(rng.standard_normal((1, D), dtype=np.float32)), (N, 1))
    kn[np.flatnonzero(np.arange(N) % 40 < 10)] = real
    q = rng.standard_normal((9, D), dtype=np.float32)
    q[0] = 0.0
    q[1] = kn[39]
    q[2] = real[7]
    idx = KeyIndex(torch.from_numpy(kn), ops=Ops)
    for k in (1, 10, 64):
        s, i = idx.topk(torch.from_numpy(q), k, idx_base=5)
        rs, ri = cref.topk_cosine(q, kn, k, idx_base=5)
        assert np.array_equal(i.numpy(), ri) and np.array_equal(s.numpy(), rs)
    n, U, largest = idx.duplicate_stats
    assert n == N and U < N // 4 and largest == 3 * N // 4 and idx.search_index.keys_normalized.shape[0] == U
    assert set(Ops.searched) == {U}                       # every search ran over the unique rows only
    few = KeyIndex(torch.from_numpy(np.tile(kn[:3], (1000, 1))), ops=Ops)   # 3 unique rows, k = 10
    s, i = few.topk(torch.from_numpy(q), 10)
    rs, ri = cref.topk_cosine(q, np.tile(kn[:3], (1000, 1)), 10)
    assert np.array_equal(i.numpy(), ri) and np.array_equal(s.numpy(), rs)
    plain = KeyIndex(torch.from_numpy(cref.normalize_rows(rng.standard_normal((3000, D), dtype=np.float32))), ops=Ops)
    plain.topk(torch.from_numpy(q), 5)
    assert plain._collapsed is False and plain.duplicate_stats == (3000, 3000, 1)
    small = KeyIndex(torch.from_numpy(kn[:1000].copy()), ops=Ops)            # below DEDUP_MIN_ROWS: not even looked at
    small.topk(torch.from_numpy(q), 5)
    assert small._collapsed is False and small.duplicate_stats is None


def test_filtered_dispatch_rule():
    """kernels.filter_helps (pure host arithmetic): which shapes take the bf16-filtered exact top-k.  Banks of >= 65536
    keys at any batch size; >= 32768 keys from 128 queries; >= 8192 keys from 512 queries (D >= 128) or 2048 / 8192
    (D = 64); smaller banks, unsupported D / k and RAGRAPH_EXACT_FP32=1 never."""
    import os

    from ragraph_amd import kernels as K

    assert os.environ.get("RAGRAPH_EXACT_FP32") != "1"
    yes = [(1, 1_000_000, 256, 10), (1, 65536, 256, 10), (40, 70_000, 256, 10), (16, 262144, 64, 3), (128, 32768, 128, 5),
           (512, 10_000, 128, 5), (2708, 10_000, 128, 5), (8192, 8192, 128, 5), (1024, 10_000, 256, 10),
           (8192, 10_000, 64, 10), (2048, 16384, 64, 10), (100_000, 1_000_000, 256, 32)]
    no = [(256, 10_000, 128, 5), (64, 32768, 256, 10), (2708, 10_000, 64, 10), (1024, 16384, 64, 10), (2708, 4096, 128, 5),
          (100_000, 8191, 256, 10), (1, 1_000_000, 96, 10), (1, 1_000_000, 256, 33)]
    for B, N, D, k in yes:
        assert K.filter_helps(B, N, D, k), (B, N, D, k)
    for B, N, D, k in no:
        assert not K.filter_helps(B, N, D, k), (B, N, D, k)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_silent_fallback_without_device():
    from ragraph_amd import kernels as K
    from ragraph_amd.ragraph_utils import Propagation, SimilarityFunctions

    with pytest.raises(K.RagraphNativeError):
        K.normalize_rows(torch.randn(4, 8))
    with pytest.raises(K.RagraphNativeError):
        SimilarityFunctions.calculate_cosine_similarity(torch.randn(2, 64), torch.randn(9, 64))
    with pytest.raises((K.RagraphNativeError, Exception)):
        Propagation.aggregate_k_hop_features(torch.eye(4), torch.randn(4, 8), 1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ragraph_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "libragraph_oracle" not in src


def test_shard_bounds_cover_and_balance():
    from ragraph_amd.sharded import shard_bounds

    for N, G in [(10, 3), (1_000_000, 8), (7, 8), (8_000_000, 8)]:
        spans = [shard_bounds(N, G, r) for r in range(G)]
        assert spans[0][0] == 0 and spans[-1][1] == N
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_data_standins_and_csr_normalisation_match_scipy():
    import scipy.sparse as sp

    from ragraph_amd.data import Batch, DataLoader, synthetic_tu_dataset
    from ragraph_amd.graph import CSRGraph

    ds = synthetic_tu_dataset(num_graphs=5, num_node_attributes=4, num_node_labels=3, seed=1)
    assert ds.num_features == 7 and len(ds[1:3]) == 2 and len(ds.shuffle()) == 5
    b = next(iter(DataLoader(ds, batch_size=3)))
    assert isinstance(b, Batch) and b.num_graphs == 3 and b.ptr[-1] == b.x.shape[0]
    # D^-1/2 (A+I) D^-1/2 in CSR == the reference's scipy recipe (ragraph_utils/utility.py:19-26,45-66)
    n = b.x.shape[0]
    ei = b.edge_index.numpy()
    A = sp.coo_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n)).tocsr() + sp.eye(n)
    d = np.power(np.asarray(A.sum(1)).flatten(), -0.5)
    ref = sp.coo_matrix(A).dot(sp.diags(d)).transpose().dot(sp.diags(d)).todense().astype(np.float32)
    g = CSRGraph.from_edge_index_sym_normalized(b.edge_index, n)
    dense = np.zeros((n, n), dtype=np.float32)
    rows = np.repeat(np.arange(n), np.diff(g.rowptr.numpy()))
    dense[rows, g.col.numpy()] = g.val.numpy()
    assert np.array_equal(dense, np.asarray(ref))
    g2 = CSRGraph.from_dense(torch.from_numpy(np.asarray(ref)))
    assert torch.equal(g2.rowptr, g.rowptr) and torch.equal(g2.col, g.col) and torch.equal(g2.val, g.val)


def test_csr_permuted_and_locality_order_host_logic():
    """CSRGraph.permuted(order) is P A P^T with ascending columns, and locality_order() is a permutation that shrinks the
    mean |row - col| of a shuffled community graph (the structure-only bookkeeping behind tools/spmm_locality.py)."""
    import torch

    from ragraph_amd.data import synthetic_community_graph
    from ragraph_amd.graph import CSRGraph

    n = 600
    ei, member = synthetic_community_graph(n, 8, 50, 0.9, seed=3, device="cpu")
    g = CSRGraph.from_edge_index_sym_normalized(ei, n)
    dense = torch.zeros(n, n)
    rows = torch.repeat_interleave(torch.arange(n), g.rowptr[1:] - g.rowptr[:-1])
    dense[rows, g.col.long()] = g.val
    order = g.locality_order()
    assert sorted(order.tolist()) == list(range(n))
    gp = g.permuted(order)
    dp = torch.zeros(n, n)
    prow = torch.repeat_interleave(torch.arange(n), gp.rowptr[1:] - gp.rowptr[:-1])
    dp[prow, gp.col.long()] = gp.val
    assert torch.equal(dp, dense[order][:, order])
    for r in range(n):  # columns ascending inside every row
        c = gp.col[gp.rowptr[r]:gp.rowptr[r + 1]]
        assert torch.all(c[1:] > c[:-1])
    span = lambda gg, rr: (rr - gg.col.long()).abs().float().mean().item()
    assert span(gp, prow) < 0.6 * span(g, rows)
    truth = torch.sort(member, stable=True).indices
    gt = g.permuted(truth)
    trow = torch.repeat_interleave(torch.arange(n), gt.rowptr[1:] - gt.rowptr[:-1])
    assert span(gt, trow) < 0.3 * span(g, rows)


def test_edge_list_ingestion_matches_reference_recipe(tmp_path):
    """ragraph_amd.edge_data vs a literal restatement of the reference's loader (dict-of-dicts edge times,
    dataloader.py:47-113; scipy bi-normalised adjacency, base_model.py:34-52) on a small TSV in the reference's format."""
    import scipy.sparse as sp

    from ragraph_amd.edge_data import EdgeListData

    rng = np.random.default_rng(4)
    U, I = 40, 30
    lines, t0 = [], 1_452_000_000
    for u in range(U):
        if u % 7 == 3:
            continue  # users without history exist in the real files
        items = rng.integers(0, I, rng.integers(1, 6))
        times = t0 + rng.integers(0, 30 * 86400, items.shape[0])
        lines.append(f"{u}\t{' '.join(map(str, items))}\t{' '.join(map(str, times))}")
    train = tmp_path / "train.txt"
    train.write_text("\n".join(lines) + "\n")
    test = tmp_path / "test.txt"
    test.write_text("\n".join(f"{u}\t{rng.integers(0, I)}" for u in range(0, U, 5)) + "\n")
    ds = EdgeListData(str(train), str(test), hour_interval=1, device="cpu")

    # --- the reference's recipe, literally ---
    edgelist, edge_time = [], []
    for line in lines:
        user, items, times = line.split("\t")
        for it in items.split(" "):
            edgelist.append((int(user), int(it)))
        for tt in times.split(" "):
            edge_time.append(int(tt))
    edgelist = np.array(edgelist, dtype=np.int32)
    ts = np.array(edge_time, dtype=np.int64)
    step = 1 + (ts - ts.min()) // 3600
    nu = max(edgelist[:, 0].max() + 1, max(range(0, U, 5)) + 1)
    ni = ds.num_items
    etd = {}
    for (a, b), s in zip(edgelist, step):
        etd.setdefault(int(a), {})[int(b) + nu] = int(s)
        etd.setdefault(int(b) + nu, {})[int(a)] = int(s)
    g = sp.coo_matrix((np.ones(len(edgelist)), (edgelist[:, 0], edgelist[:, 1])), shape=(nu, ni))
    a, b = sp.csr_matrix((nu, nu)), sp.csr_matrix((ni, ni))
    mat = sp.vstack([sp.hstack([a, g]), sp.hstack([g.transpose(), b])])
    mat = (mat != 0) * 1.0
    deg = np.array(mat.sum(axis=-1))
    with np.errstate(divide="ignore"):
        dinv = np.reshape(np.power(deg, -0.5), [-1])
    dinv[np.isinf(dinv)] = 0.0
    mat = mat.dot(sp.diags(dinv)).transpose().dot(sp.diags(dinv)).tocoo()
    ref_edges = np.stack([mat.row, mat.col], 1).astype(np.int64)
    ref_norm = mat.data.astype(np.float32)
    ref_times = np.array([etd[int(r)][int(c)] for r, c in ref_edges])

    assert ds.num_users == nu
    assert np.array_equal(ds.edges.numpy(), ref_edges)
    assert np.allclose(ds.edge_norm.numpy(), ref_norm, rtol=1e-6)
    assert np.array_equal(ds.edge_times.numpy(), ref_times)
    rp, cols = ds.history_csr([0, 3, 5], device="cpu")
    assert rp.tolist()[0] == 0 and rp.tolist()[-1] == cols.numel()


def test_segment_plan_covers_every_stage_once(tmp_path):
    """The tile kernel's work plan (segment_plan.h) is plain C++: build the exhaustive checker with g++ and run it
    (82k (tiles, stages, workgroups) combinations: exact coverage, slot order, last flags, slot bound)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_segment_plan")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "ragraph_amd", "csrc"),
                           os.path.join(root, "tools", "check_segment_plan.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "segment plans checked" in out.stdout


def test_key_index_reprobes_a_demoted_bank_with_backoff():
    """A demotion is lifted after REPROBE_QUERIES more queries, on a small call only; a probe that fails demotes again and the
    next probe waits four times as long; a probe that holds stays (kernels_index.py)."""
    import torch

    from ragraph_amd.kernels_index import KeyIndex

    class Done:
        def query(self):
            return True

    def poll(idx, n_over, B, had_i8):
        idx._pending = (torch.tensor([n_over], dtype=torch.int32), Done(), B, had_i8)
        idx._poll_overflow()

    idx = KeyIndex(torch.zeros(4, 8), ops=object())
    R = KeyIndex.REPROBE_QUERIES
    poll(idx, 1600, 100_000, True)
    assert idx._i8_off and idx._demoted["i8"] == 0
    idx._queries = R - 1
    idx._maybe_reprobe(512)
    assert idx._i8_off                       # not yet
    idx._queries = R
    idx._maybe_reprobe(100_000)
    assert idx._i8_off                       # not on a large call
    idx._maybe_reprobe(512)
    assert not idx._i8_off and idx._demoted["i8"] is None
    poll(idx, 200, 512, True)                # the probe overflowed: demoted again, the interval quadrupled
    assert idx._i8_off and idx._reprobe_after["i8"] == 4 * R and idx._demoted["i8"] == R
    idx._queries = 2 * R
    idx._maybe_reprobe(512)
    assert idx._i8_off
    idx._queries = 5 * R
    idx._maybe_reprobe(512)
    assert not idx._i8_off
    poll(idx, 0, 512, True)                  # this probe holds
    poll(idx, 0, 4096, True)
    assert not idx._i8_off and idx._reprobe_after["i8"] == 4 * R
    idx._queries = 50 * R                    # a demotion long after a probe that held is a new story: the base interval
    poll(idx, 1600, 100_000, True)
    assert idx._i8_off and idx._reprobe_after["i8"] == R
    idx._queries = 51 * R
    idx._maybe_reprobe(512)
    assert not idx._i8_off
    # the filter itself: lifted before int8 when both are down
    poll(idx, 1600, 100_000, True)
    poll(idx, 1600, 100_000, False)
    assert idx._i8_off and idx._filter_off
    idx._queries = 100 * R
    idx._maybe_reprobe(64)
    assert not idx._filter_off and idx._i8_off
    idx._maybe_reprobe(64)
    assert not idx._i8_off
