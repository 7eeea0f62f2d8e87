"""Minimal stand-ins for the torch_geometric objects the reference's call surface touches (Data / Batch / DataLoader /
TUDataset attributes: num_graphs, num_features, x, edge_index, y, ptr, num_node_attributes, num_classes, name,
shuffle(), slicing -- ragraph_utils/utility.py:31-52, finetune-rag.py:52-55), plus seeded synthetic generators.
No real TU / amazon data ships with the reference (.MISSING_LARGE_BLOBS) and there is no network: every run here is
on synthetic data of the reference's shapes and statistics.
"""
from __future__ import annotations

import numpy as np
import torch


class Data:
    def __init__(self, x, edge_index, y=None):
        self.x, self.edge_index, self.y = x, edge_index, y
        self.num_graphs = 1
        self.ptr = torch.tensor([0, x.shape[0]])

    @property
    def num_features(self):
        return self.x.shape[1]

    def __getitem__(self, i):
        assert i == 0
        return self


class Batch(Data):
    """Concatenation of graphs with batch-global node ids (what torch_geometric's collate produces)."""

    def __init__(self, graphs):
        off, xs, eis, ys, ptr = 0, [], [], [], [0]
        for g in graphs:
            xs.append(g.x)
            eis.append(g.edge_index + off)
            if g.y is not None:
                ys.append(g.y.reshape(-1))
            off += g.x.shape[0]
            ptr.append(off)
        super().__init__(torch.cat(xs), torch.cat(eis, dim=1), torch.cat(ys) if ys else None)
        self.graphs = list(graphs)
        self.num_graphs = len(graphs)
        self.ptr = torch.tensor(ptr)

    def __getitem__(self, i):
        return self.graphs[i]


class GraphDataset:
    """TUDataset duck type: a list of Data with the dataset-level attributes the drivers read."""

    def __init__(self, graphs, num_node_attributes, num_classes, name="SYNTH"):
        self.graphs = list(graphs)
        self.num_node_attributes = num_node_attributes
        self.num_classes = num_classes
        self.name = name

    @property
    def num_features(self):
        return self.graphs[0].x.shape[1]

    def __len__(self):
        return len(self.graphs)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return GraphDataset(self.graphs[i], self.num_node_attributes, self.num_classes, self.name)
        return self.graphs[i]

    def shuffle(self, generator=None):
        perm = torch.randperm(len(self.graphs), generator=generator).tolist()
        return GraphDataset([self.graphs[i] for i in perm], self.num_node_attributes, self.num_classes, self.name)


class DataLoader:
    def __init__(self, dataset, batch_size=1, shuffle=False):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = torch.randperm(len(self.dataset)).tolist() if self.shuffle else list(range(len(self.dataset)))
        for s in range(0, len(order), self.batch_size):
            yield Batch([self.dataset[i] for i in order[s:s + self.batch_size]])


# ---- synthetic generators (deterministic per seed) -----------------------------------------------------------------
def synthetic_tu_dataset(num_graphs=64, num_node_attributes=18, num_node_labels=3, num_classes=2, min_nodes=10,
                         max_nodes=80, mean_degree=3.7, seed=9, name="SYNTH_TU", attr_dist="uniform") -> GraphDataset:
    """TU-shaped graphs (ENZYMES: 18 attrs + 3 one-hot node labels; PROTEINS: 1 + 3), sizes ~U[min,max].  attr_dist:
    "uniform" [0, 1) attributes (TU-like) or "normal" (the feature distribution of bench.py's query graph)."""
    rng = np.random.default_rng(seed)
    graphs = []
    for _ in range(num_graphs):
        n = int(rng.integers(min_nodes, max_nodes + 1))
        m = max(n - 1, int(n * mean_degree / 2))
        src = np.concatenate([np.arange(n - 1), rng.integers(0, n, m - (n - 1))])   # a path keeps it connected
        dst = np.concatenate([np.arange(1, n), rng.integers(0, n, m - (n - 1))])
        keep = src != dst
        und = np.unique(np.stack([np.minimum(src, dst)[keep], np.maximum(src, dst)[keep]], 1), axis=0)
        ei = np.concatenate([und, und[:, ::-1]], 0).T                                   # both directions, once each
        attrs = (rng.random((n, num_node_attributes), dtype=np.float32) if attr_dist == "uniform" else
                 rng.standard_normal((n, num_node_attributes), dtype=np.float32))
        nl = np.eye(num_node_labels, dtype=np.float32)[rng.integers(0, num_node_labels, n)]
        graphs.append(Data(torch.from_numpy(np.concatenate([attrs, nl], 1)), torch.from_numpy(ei.copy()).long(),
                           torch.tensor([int(rng.integers(0, num_classes))])))
    return GraphDataset(graphs, num_node_attributes, num_classes, name)


def synthetic_big_graph(n=100_000, mean_degree=10, seed=8, device="cuda"):
    """Config-2 graph: ring + Erdos-Renyi extras, undirected, mean degree ~10 -> edge_index [2,E] on `device`."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    extra = n * (mean_degree - 2) // 2
    src = torch.cat([torch.arange(n), torch.randint(0, n, (extra,), generator=g)])
    dst = torch.cat([(torch.arange(n) + 1) % n, torch.randint(0, n, (extra,), generator=g)])
    keep = src != dst
    src, dst = src[keep], dst[keep]
    return torch.stack([torch.cat([src, dst]), torch.cat([dst, src])]).to(device)


def synthetic_community_graph(n=100_000, mean_degree=10, community=512, p_intra=0.9, seed=9, shuffle=True, device="cuda"):
    """A graph WITH structure (the c2 graph above is Erdos-Renyi: nothing to reorder): communities of `community` nodes,
    a fraction p_intra of every node's edges inside its community, node ids shuffled so that the given numbering hides
    the communities (real datasets arrive like that).  Returns (edge_index [2,E] on `device`, membership [n] on the host:
    the community of every node in the RETURNED numbering -- the oracle ordering a reordering is measured against)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    half = n * mean_degree // 2
    src = torch.randint(0, n, (half,), generator=g)
    intra = torch.rand(half, generator=g) < p_intra
    base = src // community * community
    size = torch.full_like(base, community).minimum(n - base)   # (the last community may be short)
    dst_in = base + (torch.rand(half, generator=g) * size).long().minimum(size - 1)
    dst = torch.where(intra, dst_in, torch.randint(0, n, (half,), generator=g))
    keep = src != dst
    src, dst = src[keep], dst[keep]
    member = torch.arange(n) // community
    if shuffle:
        perm = torch.randperm(n, generator=g)   # old id -> new id
        src, dst = perm[src], perm[dst]
        m2 = torch.empty_like(member)
        m2[perm] = member
        member = m2
    return torch.stack([torch.cat([src, dst]), torch.cat([dst, src])]).to(device), member


def synthetic_bank(N, D, C, device="cuda", seeds=(1234, 1235, 1236)):
    """K = normalize(randn), V = randn, L = one_hot(randint).  Generated on the device in slabs
    (a 1M x 256 fp32 host tensor would be a 1 GB PCIe copy)."""
    gk = torch.Generator(device=device).manual_seed(seeds[0])
    gv = torch.Generator(device=device).manual_seed(seeds[1])
    gl = torch.Generator(device=device).manual_seed(seeds[2])
    K = torch.randn(N, D, device=device, generator=gk)
    V = torch.randn(N, D, device=device, generator=gv)
    L = torch.nn.functional.one_hot(torch.randint(0, C, (N,), device=device, generator=gl), C).float()
    return K, V, L


def synthetic_bipartite(num_users=3000, num_items=2000, edges_per_user=10, seed=10, hours=720, device="cuda"):
    """Config-5 style dynamic bipartite graph: Zipf-ish item popularity, hourly time steps.  Returns the reference's
    edge-list form: edges [2E,2] (src,dst; both directions, global ids), edge_norm [2E] (bi-normalised), times [2E]."""
    rng = np.random.default_rng(seed)
    u = np.repeat(np.arange(num_users), edges_per_user)
    i = np.minimum((rng.pareto(1.1, len(u)) * num_items / 50).astype(np.int64), num_items - 1)
    pairs = np.unique(np.stack([u, i], 1), axis=0)
    u, i = pairs[:, 0], pairs[:, 1] + num_users
    t = rng.integers(0, hours, len(u))
    src = np.concatenate([u, i])
    dst = np.concatenate([i, u])
    deg = np.bincount(src, minlength=num_users + num_items).astype(np.float64)
    norm = (deg[src] ** -0.5) * (deg[dst] ** -0.5)   # base_model.py:34-52 (D^-1/2 A D^-1/2 on the 0/1 bipartite graph)
    edges = torch.from_numpy(np.stack([src, dst], 1)).to(device)
    return edges, torch.from_numpy(norm.astype(np.float32)).to(device), torch.from_numpy(np.concatenate([t, t])).to(device)
