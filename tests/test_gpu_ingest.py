"""SURVEY.md section 8(f) row 2 on the GPU: the ingestion kernels (csrc/ingest.hip) against golden g14 produced by the
reference's own process_tu_dataset / EdgeListData + _make_binorm_adj, and against the host restatements at sizes the
reference cannot ingest (its block-diagonal adjacency is dense)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def test_tu_batch_to_csr_matches_reference_g14(dev):
    from ragraph_amd.data import Data
    from ragraph_amd.ragraph_utils import process_tu_dataset

    g = gold("g14_ingestion")
    n = g["tu_x"].shape[0]
    batch = Data(torch.from_numpy(g["tu_x"]), torch.from_numpy(g["tu_edge_index"]))
    feats, adj, labels = process_tu_dataset(batch, int(g["tu_num_node_attributes"]), device=dev)
    dense = torch.zeros(n, n, device=dev)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), adj.rowptr[1:] - adj.rowptr[:-1])
    dense[rows, adj.col.long()] = adj.val
    assert np.allclose(dense.cpu().numpy(), g["tu_adj"], atol=1e-7)
    assert np.array_equal(dense.cpu().numpy() != 0, g["tu_adj"] != 0)
    assert np.array_equal(feats.cpu().numpy(), g["tu_features"]) and np.array_equal(labels.cpu().numpy(), g["tu_node_labels"])
    cols = adj.col.long()
    assert bool(((cols[1:] > cols[:-1]) | (rows[1:] != rows[:-1])).all())   # ascending columns within every row


def test_edge_tsv_to_edges_matches_reference_g14(dev, tmp_path):
    from ragraph_amd.edge_data import EdgeListData

    g = gold("g14_ingestion")
    tr, te = tmp_path / "train.txt", tmp_path / "test.txt"
    tr.write_text(str(g["edge_train_txt"]))
    te.write_text(str(g["edge_test_txt"]))
    ds = EdgeListData(str(tr), str(te), hour_interval=int(g["edge_hour_interval"]), device=dev)
    assert np.array_equal(ds.edges.cpu().numpy(), g["edge_edges"])
    assert np.array_equal(ds.edge_times.cpu().numpy(), g["edge_times"])
    assert np.allclose(ds.edge_norm.cpu().numpy(), g["edge_norm"], atol=1e-7)


def test_ingestion_kernels_match_host_restatement_at_scale(dev):
    """100k-node graph with duplicate edges (the config-2 graph) and 1M interactions with repeated (user, item) pairs:
    HIP == the torch / numpy restatements, bit for bit."""
    from ragraph_amd import kernels as K
    from ragraph_amd.data import synthetic_big_graph
    from ragraph_amd.edge_data import binorm_edges
    from ragraph_amd.graph import CSRGraph

    n = 100_000
    ei = synthetic_big_graph(n, 10, seed=8, device=dev)
    ei = torch.cat([ei, ei[:, :5000]], dim=1)                     # duplicates: multiplicity 2 entries
    rowptr, col, val = K.csr_sym_normalized_from_edges(ei, n)
    ref = CSRGraph.from_edge_index_sym_normalized(ei.cpu(), n)    # CPU tensors -> the torch restatement
    assert torch.equal(rowptr.cpu(), ref.rowptr) and torch.equal(col.cpu(), ref.col)
    assert torch.equal(val.cpu(), ref.val)
    rng = np.random.default_rng(1)
    U, I, E = 30_000, 20_000, 1_000_000
    u = rng.integers(0, U, E)
    i = np.minimum((rng.pareto(1.1, E) * I / 50).astype(np.int64), I - 1)
    step = rng.integers(1, 700, E)
    e_ref, n_ref, t_ref = binorm_edges(U, I, u, i, step)
    e, nm, t = K.binorm_edges(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev), torch.from_numpy(step).to(dev), U, I)
    assert np.array_equal(e.cpu().numpy(), e_ref) and np.array_equal(t.cpu().numpy(), t_ref)
    assert np.array_equal(nm.cpu().numpy(), n_ref)


@pytest.mark.parametrize("n,bits,val_bytes", [(1, 64, 4), (255, 8, 0), (2049, 17, 8), (100_000, 40, 4), (1_000_003, 64, 8), (4097, 64, 0)])
def test_radix_sort_and_scan_match_numpy(dev, n, bits, val_bytes):
    """The library's own stable radix sort (64-bit keys, low `bits` bits, optional 4- / 8-byte values) and prefix sums
    (csrc/sortscan.hip: what ingestion and the duplicate grouping of a bank are built on) against numpy's stable argsort and
    cumsum; inputs untouched; keys drawn from few values so that stability is what is tested."""
    import ctypes

    import numpy as np
    import torch

    from ragraph_amd import kernels as K

    L = K._ready()
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64) >> np.uint64(rng.integers(0, 50))
    keys[rng.random(n) < 0.5] = keys[0]                       # many ties
    mask = np.uint64((1 << bits) - 1) if bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    kd = torch.from_numpy(keys.view(np.int64)).to(dev)
    ko = torch.empty_like(kd)
    vals = np.arange(n, dtype=np.int64 if val_bytes == 8 else np.int32)
    vd = torch.from_numpy(vals).to(dev) if val_bytes else None
    vo = torch.empty_like(vd) if val_bytes else None
    ws = torch.empty(L.ragraph_radix_sort_workspace_bytes(n, val_bytes), dtype=torch.uint8, device=dev)
    K.N.check(L.ragraph_radix_sort_u64(kd.data_ptr(), ko.data_ptr(), vd.data_ptr() if val_bytes else None,
                                       vo.data_ptr() if val_bytes else None, val_bytes, n, bits, ws.data_ptr(), ws.numel(), None), "radix_sort")
    torch.cuda.synchronize()
    assert np.array_equal(ko.cpu().numpy().view(np.uint64), keys[order])
    assert np.array_equal(kd.cpu().numpy().view(np.uint64), keys)          # input untouched
    if val_bytes:
        assert np.array_equal(vo.cpu().numpy(), vals[order])
    x = rng.integers(0, 7, n).astype(np.int32)
    xd = torch.from_numpy(x).to(dev)
    out = torch.empty_like(xd)
    ws2 = torch.empty(L.ragraph_scan_workspace_bytes(n), dtype=torch.uint8, device=dev)
    for inclusive in (0, 1):
        K.N.check(L.ragraph_scan_sum_i32(xd.data_ptr(), out.data_ptr(), n, inclusive, ws2.data_ptr(), ws2.numel(), None), "scan")
        ref = np.cumsum(x, dtype=np.int64)
        ref = ref if inclusive else ref - x
        assert np.array_equal(out.cpu().numpy(), ref.astype(np.int32))


@pytest.mark.parametrize("E,n,sort_cols", [(0, 5, False), (1, 1, True), (1000, 37, False), (1000, 37, True), (300_000, 70_000, False),
                                           (300_000, 70_000, True), (2_000_000, 1000, False), (50_000, 3_000_000, True)])
def test_coo_to_csr_matches_the_stable_sort_formulation(dev, E, n, sort_cols):
    """ragraph_coo_to_csr_i64 (own radix sort; every per-step graph rebuild of the edge flavour, the transposed patterns of
    training and the gather backward) against torch.sort(stable) + bincount + cumsum on the HOST: the same row pointers, the
    same permutation (ties keep the input order: scatter_add_'s accumulation order, modules/utils.py:17-32), the same columns;
    empty rows at both ends and in the middle, duplicate (row, col) pairs."""
    from ragraph_amd import kernels as K
    from ragraph_amd.graph import CSRGraph

    g = torch.Generator().manual_seed(E + n)
    rows = torch.randint(0, n, (E,), generator=g)
    if E > 100:
        rows[rows == 0] = min(1, n - 1)                      # row 0 empty
        rows[rows == n - 1] = max(n - 2, 0)                  # the last row empty
        rows[E // 2:E // 2 + 5] = rows[E // 2]               # duplicates: the stable order decides
    cols = torch.randint(0, n, (E,), generator=g)
    if E > 100:
        cols[E // 2:E // 2 + 5] = cols[E // 2]
    vals = torch.randn(E, generator=g)
    ref, ref_perm = CSRGraph.from_coo(rows, cols, vals, n, sort_cols=sort_cols)           # host tensors: the torch formulation
    got, perm = CSRGraph.from_coo(rows.to(dev), cols.to(dev), vals.to(dev), n, sort_cols=sort_cols)
    assert torch.equal(got.rowptr.cpu(), ref.rowptr) and torch.equal(perm.cpu(), ref_perm)
    assert torch.equal(got.col.cpu(), ref.col) and torch.equal(got.val.cpu(), ref.val)
    assert torch.equal(got.row_ids().cpu(), ref.row_ids())
    if E and sort_cols:
        t_ref = ref.transposed()
        t = got.transposed()
        assert torch.equal(t.rowptr.cpu(), t_ref.rowptr) and torch.equal(t.col.cpu(), t_ref.col) and torch.equal(t.val.cpu(), t_ref.val)


@pytest.mark.parametrize("E,p", [(0, 0.5), (1, 1.0), (1000, 0.0), (100_003, 0.5), (3_000_000, 0.31)])
def test_mask_positions_matches_nonzero(dev, E, p):
    """ragraph_mask_positions_i64 (the edge flavour's per-step edge dropout, modules/utils.py:40-53, on the library's own prefix
    sums) against torch.nonzero: the same positions in the same order, bool and uint8 masks, nothing set, everything set."""
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(E + 7)
    mask = torch.rand(E, device=dev, generator=g) < p
    want = torch.nonzero(mask).reshape(-1)
    assert torch.equal(K.mask_positions(mask), want)
    assert torch.equal(K.mask_positions(mask.to(torch.uint8)), want)


def test_verify_merged_prior_matches_the_formula(dev):
    """ragraph_verify_merged_prior_f32 (the owner's verdict on the merged lists of a sharded call): rows below the prior are
    counted, all-zero queries need no proof, the smallest / largest proven k-th best and the statistics words' candidates."""
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(5)
    R, k = 10_001, 10
    s = torch.sort(torch.rand(R, k, device=dev, generator=g), dim=1, descending=True).values
    s[7] = 0.0                                  # a zero query: every score +0
    s[9, k - 1] = float("-inf")                 # fewer than k candidates: never proven under a prior
    kth = s[:, k - 1]
    prior = float(torch.quantile(kth[torch.isfinite(kth)], 0.2))
    words = torch.zeros(32, dtype=torch.int32, device=dev)
    words[0], words[2], words[3], words[5], words[6] = K.FILTER_STATS_MAGIC, 300, 90, 10, 9
    over = torch.tensor([3], dtype=torch.int32, device=dev)
    for pr in (None, prior):
        out = K.verify_merged_prior(s, pr, words, over).cpu().tolist()
        zero = (s[:, 0] == 0) & (kth == 0)
        ok = zero | (kth >= pr) if pr is not None else torch.ones_like(zero)
        live = ok & ~zero & torch.isfinite(kth)
        assert out[0] == float((~ok).sum()) and out[1] == -float(kth[live].min()) and out[2] == float(kth[live].max())
        assert abs(out[3] - 40.0) < 1e-5 and out[4] == 3.0
    assert K.verify_merged_prior(s[:0], prior).cpu().tolist()[0] == 0.0
