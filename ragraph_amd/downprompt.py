"""Mirror of the GraphPrompt downstream prompt (RAGraph_graph/downprompt.py): w * h, per-graph sum readout, cosine to
class-mean prototypes, log_softmax.  The reference's per-graph / per-sample Python loops (downprompt.py:45-52,104-110)
become two kernels: segment_reduce (prompt multiply fused into the sum) and proto_cosine."""
import torch
import torch.nn as nn

from . import autograd as AG
from . import kernels as K


class downstreamprompt(nn.Module):
    def __init__(self, hid_units):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(1, hid_units))
        torch.nn.init.xavier_uniform_(self.weight)     # downprompt.py:160-161

    def forward(self, graph_embedding):                # downprompt.py:164-168: weight * h
        return AG.mul_cols(graph_embedding, self.weight)  # (inference's fused form: split_and_batchify(..., weight))


def split_and_batchify_graph_feats(batched_graph_feats, graph_sizes, weight=None):
    """downprompt.py:98-112: per-graph SUM of node rows; `weight` fuses downstreamprompt's multiply into the pass."""
    sizes = graph_sizes.to(batched_graph_feats.device, torch.int64).reshape(-1)
    seg = torch.zeros(sizes.numel() + 1, dtype=torch.int64, device=sizes.device)
    seg[1:] = torch.cumsum(sizes, 0)
    return K.segment_reduce(batched_graph_feats, seg, w=weight)


def predict(graphnum, nb_classes, rawret, ave):
    """downprompt.py:41-56: log_softmax over cosine(rawret[g], ave[c])."""
    return K.proto_cosine(rawret[:graphnum], ave[:nb_classes], mode=2)


def averageemb(labels, rawret, nb_class):
    """Class-mean prototypes.  The reference (downprompt.py:59-94) averages over an UNINITIALISED [C, n, D] buffer, so
    its result is garbage-dependent; this is the evident intent: mean of the rows of each class."""
    lab = labels.reshape(-1).long().to(rawret.device)
    order = torch.sort(lab, stable=True).indices
    counts = torch.bincount(lab, minlength=nb_class)
    seg = torch.zeros(nb_class + 1, dtype=torch.int64, device=rawret.device)
    seg[1:] = torch.cumsum(counts, 0)
    out = K.segment_reduce(K.gather_rows(rawret, order), seg, mean_mode=True)    # any D
    out[counts == 0] = 0                                                          # (0 / 0 of an empty class)
    return out


class downprompt(nn.Module):
    """downprompt.py:6-31: forward(seq, graph_len) -> per-graph embedding of the prompted features."""

    def __init__(self, prompt1, prompt2, prompt3, ft_in, nb_classes):
        super().__init__()
        self.downprompt = downstreamprompt(ft_in)
        self.nb_classes = nb_classes

    def forward(self, seq, graph_len):
        return split_and_batchify_graph_feats(seq, graph_len, weight=self.downprompt.weight.reshape(-1))
