"""Mirror of RAGraph_*/models/gcnlayers.py: the encoder stack (state_dict keys convs.N.*, g_net.N.*, bns.N.*)."""
import torch
import torch.nn as nn

from .layers import GCN


class GcnLayers(nn.Module):
    def __init__(self, n_in, n_h, num_layers_num, dropout):
        super().__init__()
        self.act = nn.ReLU()
        self.num_layers_num = num_layers_num
        self.convs = nn.ModuleList()
        self.bns = nn.ModuleList()
        for i in range(num_layers_num):  # models/gcnlayers.py:22-37
            self.convs.append(GCN(n_h if i else n_in, n_h))
            self.bns.append(nn.BatchNorm1d(n_h))
        self.g_net = self.convs  # the reference aliases these (gcnlayers.py:16), so checkpoints carry both key sets
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, seq, adj, sparse, LP=False):
        """models/gcnlayers.py:40-67.  LP=True (pre-training: BatchNorm + dropout between layers) is outside the
        inference path and not provided."""
        if LP:
            raise NotImplementedError("GcnLayers(LP=True) is the pre-training branch (BatchNorm+dropout); out of scope")
        out = torch.squeeze(seq, dim=0)
        for i in range(self.num_layers_num):
            out = self.convs[i]((out, adj))
        return out.unsqueeze(dim=0)

    @torch.no_grad()
    def forward_rows(self, seq, adj, lo: int, hi: int):
        """Rows [lo, hi) of forward(seq, adj, False): every layer but the last over all rows (the last layer's rows gather
        from all of its input), the last over the slice.  None when the graph has hub rows (their blocks are spread by the
        whole-graph entry)."""
        from .graph import as_csr
        g = as_csr(adj)
        if g.has_long_rows:
            return None
        out = torch.squeeze(seq, dim=0)
        for i in range(self.num_layers_num - 1):
            out = self.convs[i]((out, g))
        return self.convs[self.num_layers_num - 1].forward_rows(out, g, lo, hi)

    # few-shot split (RAGraph_node_fewshot/models/gcnlayers.py:62-85): encode = layer 0, decode = layer 1
    @torch.no_grad()
    def encode(self, seq, adj):
        return self.convs[0]((torch.squeeze(seq, dim=0), adj))

    def decode(self, seq, adj):
        assert self.num_layers_num >= 2
        return self.convs[1]((seq, adj))
