"""Mirror of RAGraph_*/preprompt.py :: PrePrompt -- the inference side (encoder stack); pre-training losses are out
of scope."""
import torch
import torch.nn as nn

from .gcnlayers import GcnLayers
from .layers.gcn import sparse_features


class PrePrompt(nn.Module):
    def __init__(self, n_in, n_h, activation, num_layers_num, p):
        super().__init__()
        self.gcn = GcnLayers(n_in, n_h, num_layers_num, p)

    @torch.no_grad()   # (the reference detaches the result: nothing upstream of it can train through this call)
    def embed(self, seq, adj, sparse, msk, LP):
        """preprompt.py:57-62.  Returns (h, c).  The node flavour's c is the 3-hop subgraph readout that a Python loop
        over nnz(A^3) computes (preprompt.py:8-27) and inference() throws away; here c is the plain mean readout of h
        (the graph flavour's AvgReadout, RAGraph_graph/preprompt.py:48-54)."""
        sparse_features(seq)  # (bag-of-words features are judged here, once per tensor version: layers/gcn.py)
        h = self.gcn(seq, adj, sparse, LP).squeeze(0)
        return h.detach(), h.mean(dim=0, keepdim=True).detach()

    @torch.no_grad()   # (detached by the reference: always the inference kernels, whatever the caller's grad mode)
    def inference(self, features, adj):
        """preprompt.py:64-66: L GCN layers, detached."""
        sparse_features(features)
        return self.gcn(features, adj, False, False).squeeze(0).detach()

    @torch.no_grad()
    def inference_rows(self, features, adj, lo: int, hi: int):
        """Rows [lo, hi) of inference(features, adj), bit for bit, at the cost of those rows (+ the earlier layers): what a rank
        that answers a slice of the queries needs before its retrieval can start (None: take the whole-graph call)."""
        sparse_features(features)
        h = self.gcn.forward_rows(features, adj, lo, hi)
        return None if h is None else h.detach()

    def encode(self, features, adj):  # RAGraph_node_fewshot/preprompt.py:74-75
        sparse_features(features)
        return self.gcn.encode(features, adj)

    def decode(self, features, adj):  # RAGraph_node_fewshot/preprompt.py:77-78
        return self.gcn.decode(features, adj)

    def load_reference_state_dict(self, state_dict):
        """Load a reference checkpoint (modelset/model_*.pkl): keeps gcn.convs.* / gcn.bns.*, ignores the duplicated
        gcn.g_net.* aliases and the pre-training heads (dgi, graphcledge, graphclmask, lp)."""
        own = self.state_dict()
        picked = {k: v for k, v in state_dict.items() if k in own}
        missing = [k for k in own if k not in picked and not k.startswith("gcn.g_net.") and "bns" not in k]
        if missing:
            raise KeyError(f"reference checkpoint lacks {missing}")
        return self.load_state_dict(picked, strict=False)
