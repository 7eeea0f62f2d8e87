#!/usr/bin/env python3
"""Sampled candidate counts (the statistics block of a filtered call) on the two banks of
tests/test_gpu_kernels.py::test_candidate_statistics_and_cost_aware_int8_demotion."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cref
from ragraph_amd import kernels as K
dev = torch.device("cuda", 0)
rng = np.random.default_rng(97)
N, D, B, k = 70000, 256, 17000, 10
centre = rng.standard_normal((1, D), dtype=np.float32)
for csize in (0, 3000, 4500):
    kn = cref.normalize_rows(np.concatenate([centre + 0.5 * rng.standard_normal((csize, D), dtype=np.float32),
                                             rng.standard_normal((N - csize, D), dtype=np.float32)]))
    q = (centre + 0.5 * rng.standard_normal((B, D), dtype=np.float32)).astype(np.float32) if csize else rng.standard_normal((B, D), dtype=np.float32)
    knd, qd = torch.from_numpy(kn).to(dev), torch.from_numpy(q).to(dev)
    for cap in (-1, 0):
        K.set_max_i8_levels(cap)
        s, i, over, st = K.topk_cosine_filtered(qd, knd, K.keys_to_bf16(knd), k, return_stats=True)
        print(f"cluster {csize} cap {cap}: over {int(over)} levels {K.filter_stats_levels(st.cpu().tolist())} planned i8 {K.expected_i8_candidates(B, N, D, k):.0f}", flush=True)
    K.set_max_i8_levels(-1)
