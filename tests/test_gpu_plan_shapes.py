"""Tile-kernel work plans (ragraph_amd/csrc/segment_plan.h) that a 256-CU device only produces at huge batch sizes are
forced here by planning for fewer workgroups (RAGRAPH_TOPK_CUS, read once per process -> one subprocess per setting):
full rounds, lockstep (Euclid) steps, linear remainders with straddling pieces, in XCD groups and in one group.
Every run is compared bit for bit with the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np, torch
sys.path.insert(0, {root!r})
from oracle import cref
from ragraph_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng({seed})
bad = 0
for (B, N, D, k, packed) in {shapes!r}:
    keys = rng.standard_normal((N, D), dtype=np.float32)
    if N > 8:
        keys[N // 2:] = keys[: N - N // 2]          # exact ties across segments
    kn = cref.normalize_rows(keys)
    q = rng.standard_normal((B, D), dtype=np.float32)
    q[B // 3] = 0.0
    knd = torch.from_numpy(kn).to(dev)
    kp = K.pack_keys(knd) if packed else None
    s, i = K.topk_cosine(torch.from_numpy(q).to(dev), knd, k, idx_base=3, keys_packed=kp)
    rs, ri = cref.topk_cosine(q, kn, k, idx_base=3)
    ok = np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    print("OK " if ok else "BAD", B, N, D, k, packed, flush=True)
    bad += (not ok)
sys.exit(1 if bad else 0)
"""


@pytest.mark.gpu
@pytest.mark.parametrize(
    "cus,shapes",
    [
        # 8 groups of 1 workgroup: every tile is a full round
        (8, [(16500, 300, 256, 5, True), (16500, 300, 64, 5, False)]),
        # 8 groups of 3: 65..70 tiles -> 8 or 9 per group = 2 full rounds + 2/3 leftover tiles (linear remainder)
        (24, [(17000, 700, 256, 10, True), (17900, 1000, 128, 7, False)]),
        # 8 groups of 5 with 9..10 tiles each: one full round + 4..5 leftover (lockstep steps, then remainder)
        (40, [(19000, 2100, 256, 10, True), (20300, 1500, 256, 14, False)]),
        # one group of 40 / 64 workgroups (fewer than 64 tiles): 12, 27, 45 leftover tiles
        (40, [(3000, 4000, 256, 10, True), (6900, 3000, 64, 3, False), (11500, 900, 256, 31, False)]),
        (64, [(7000, 5000, 256, 10, True), (15000, 1200, 128, 10, False)]),
    ],
)
def test_forced_plans_bit_exact(cus, shapes):
    env = dict(os.environ, RAGRAPH_TOPK_CUS=str(cus))
    code = CHILD.format(root=ROOT, seed=cus, shapes=shapes)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("OK ") == len(shapes), r.stdout
