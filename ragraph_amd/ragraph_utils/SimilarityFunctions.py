"""Mirror of RAGraph_*/ragraph_utils/SimilarityFunctions.py (same names, argument meaning and result)."""
import torch

from .. import kernels as K


class SimilarityFunctions:
    @staticmethod
    def calculate_cosine_similarity(search_keys: torch.Tensor, resource_keys: torch.Tensor) -> torch.Tensor:
        """normalize(search_keys) @ normalize(resource_keys).T -> [B,N] (or [N] for a 1-D query).
        Reference: SimilarityFunctions.py:6-16.  This MATERIALISES the score matrix like the reference does; the hot
        path (ToyGraphBase.retrieve) uses the fused top-k kernel instead and never builds it."""
        one_d = search_keys.dim() == 1
        q = K.normalize_rows(search_keys.reshape(1, -1) if one_d else search_keys)
        kn = K.normalize_rows(resource_keys)
        s = K.linear(q, kn)
        return s.reshape(-1) if one_d else s

    @staticmethod
    def calculate_jaccard_similarity(adj: torch.Tensor, v_c: int, v_m: int) -> float:
        """SimilarityFunctions.py:18-32 (off the hot path; unused by any forward): |N(c) & N(m)| / |N(c) | N(m)|."""
        a = set(adj[v_c].nonzero(as_tuple=False).flatten().tolist())
        b = set(adj[v_m].nonzero(as_tuple=False).flatten().tolist())
        union = len(a | b)
        return 0.0 if union == 0 else len(a & b) / union
