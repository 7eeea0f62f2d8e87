"""Mirror of RAGraph_*/ragraph_utils/utility.py: seeding and TU-batch preprocessing, emitting CSR instead of a dense
block-diagonal adjacency (the numpy row_stack loop of utility.py:43-58 is quadratic in the batch's node count)."""
import os
import random

import numpy as np
import torch

from ..graph import CSRGraph


def seed_everything(seed: int):
    """utility.py:5-16."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def process_tu_dataset(data, num_node_attributes, device="cuda"):
    """utility.py:30-72 (node flavour): a PyG-style batch -> (features [n,F], adj CSRGraph, node_labels [n,C]).
    `data` needs .x [n, F + C] (attributes then one-hot node labels) and .edge_index [2,E] with batch-global node ids
    (what torch_geometric's Batch holds; block-diagonal structure is implicit)."""
    x = data.x.to(device)
    features = x[:, :num_node_attributes].float().contiguous()
    node_labels = x[:, num_node_attributes:].float().contiguous()
    adj = CSRGraph.from_edge_index_sym_normalized(data.edge_index.to(device), x.shape[0])
    return features, adj, node_labels
