"""Mirror of the NODE flavour of the GraphPrompt downstream prompt (RAGraph_node/downprompt.py, byte-identical in
RAGraph_node_fewshot/): ELU(weight * h) -> cosine to the three class prototypes -> softmax.

The reference's per-sample Python loop with three `torch.cosine_similarity` calls per node (downprompt.py:41-44) is
ONE launch of proto_cosine (softmax mode); the prompt multiply + ELU is one launch of mul_cols (differentiable: the
prompt weight is the class's trainable parameter).  Same class names and argument meaning as the reference file; the
graph flavour of the same-named file lives in ragraph_amd/downprompt.py.
"""
import torch
import torch.nn as nn

from . import autograd as AG
from . import kernels as K

NB_CLASSES = 3   # hard-wired in the reference: ret has 3 columns, ave[0..2] (downprompt.py:37,41-44,59)


class downstreamprompt(nn.Module):
    """downprompt.py:118-130: act(weight * graph_embedding), act = nn.ELU() (alpha = 1)."""

    def __init__(self, hid_units):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(1, hid_units))
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.weight)     # downprompt.py:124-125

    def forward(self, graph_embedding):
        return AG.mul_cols(graph_embedding, self.weight, K.ACT_ELU, 1.0)


class weighted_prompt(nn.Module):
    """downprompt.py:80-95: [1, 3] mixing weights (0.9, 0.9, 0.1) times the stacked prompts."""

    def __init__(self, weightednum):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(1, weightednum))
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():                          # downprompt.py:89-91
            self.weight[0][0] = 0.9
            self.weight[0][1] = 0.9
            self.weight[0][2] = 0.1

    def forward(self, graph_embedding):                # torch.mm(weight [1,w], emb [w,D]) -> [1,D]
        return AG.linear(self.weight, graph_embedding.t().contiguous())


class weighted_feature(nn.Module):
    """downprompt.py:100-114: ELU(w0 * a + w1 * b), (w0, w1) initialised to (1, 0)."""

    def __init__(self, weightednum):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(1, weightednum))
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            self.weight[0][0] = 1.0
            self.weight[0][1] = 0.0

    def forward(self, graph_embedding1, graph_embedding2):
        # (the weights stay on the device: no host read-back, and the gradient reaches them)
        mixed = AG.mix2(graph_embedding1, graph_embedding2, self.weight)
        ones = torch.ones(mixed.shape[-1], device=mixed.device)
        return AG.mul_cols(mixed, ones, K.ACT_ELU, 1.0)


def averageemb(labels, rawret):
    """downprompt.py:59-78.  The reference copies the rows of class c into a [3, n/2, D] buffer made by
    `torch.FloatTensor(...)` (UNINITIALISED memory) and takes `mean(dim=1)` over all n/2 slots: with a zero-filled
    buffer that is (sum of the class's rows) / floor(n/2) -- implemented here; anything else the reference returns is
    whatever the allocator left in the buffer.  forward() only uses the prototypes' directions (cosine), so the
    scaling does not reach the class probabilities.  As in the reference, a class with more than n/2 members does not
    fit the buffer: IndexError."""
    n = rawret.shape[0]
    half = int(n / 2)
    lab = labels.reshape(-1).long().to(rawret.device)
    keep = (lab >= 0) & (lab < NB_CLASSES)             # rows of other labels are skipped (the three `if`s)
    lab_k = lab[keep]
    counts = torch.bincount(lab_k, minlength=NB_CLASSES)
    if half == 0 or int(counts.max()) > half:
        raise IndexError(f"averageemb: a class has {int(counts.max())} rows but the reference's buffer holds "
                         f"int({n}/2) = {half} per class (RAGraph_node/downprompt.py:61)")
    order = torch.nonzero(keep).reshape(-1)[torch.sort(lab_k, stable=True).indices]
    seg = torch.zeros(NB_CLASSES + 1, dtype=torch.int64, device=rawret.device)
    seg[1:] = torch.cumsum(counts, 0)
    # (differentiable: a training step keeps the prototypes in the graph of its own embeddings, downprompt.py:24-25)
    sums = AG.segment_sum(AG.gather_rows(rawret, order), seg)                     # rows added in index order
    return AG.mul_cols(sums, torch.full((rawret.shape[1],), 1.0 / half, device=rawret.device))


class downprompt(nn.Module):
    """downprompt.py:6-48: forward(seq, train=0) -> class probabilities [n, 3]."""

    def __init__(self, prompt1, prompt2, prompt3, ft_in, nb_classes, feature, labels):
        super().__init__()
        self.labels = labels
        self.downprompt = downstreamprompt(ft_in)
        self.prompt = torch.cat((prompt1, prompt2, prompt3), 0)
        self.nodelabelprompt = weighted_prompt(3)
        self.dffprompt = weighted_feature(2)
        feature = feature.squeeze()
        self.one = torch.ones(1, ft_in, device=feature.device)
        self.ave = averageemb(labels=self.labels, rawret=feature)

    def forward(self, seq, train=0):
        rawret = self.downprompt(seq)
        if train == 1:
            # as in the reference the prototypes of a training step stay in its autograd graph: the gradient reaches the
            # prompt weight through both sides of the cosine (downprompt.py:24-25)
            self.ave = averageemb(labels=self.labels, rawret=rawret)
        ave = self.ave if train == 1 else self.ave.detach()      # (an evaluation never walks a past step's graph)
        return AG.proto_cosine(rawret, ave, mode=1)              # cosine to ave[0..2], softmax over dim 1
