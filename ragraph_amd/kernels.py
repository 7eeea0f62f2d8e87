"""torch.Tensor front end of the C ABI (include/ragraph_hip.h).

PyTorch is plumbing here: it owns HBM allocations and the HIP stream; every number is produced by libragraph_hip.so.
All functions require CUDA(=ROCm) tensors and raise `RagraphNativeError` otherwise -- there is no eager fallback.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _native as N
from ._native import ACT_ELU, ACT_LEAKY, ACT_NONE, ACT_PRELU, ACT_RELU, RagraphNativeError  # noqa: F401

_device_ok = False


def _ready():
    global _device_ok
    if not _device_ok:
        N.require_device()
        _device_ok = True
    return N.lib()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RagraphNativeError(f"{name}: expected a ROCm device tensor (ragraph_amd has no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _idxc(t: torch.Tensor, name: str, dtype=torch.int64) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RagraphNativeError(f"{name}: expected a ROCm device tensor")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current HIP stream of the current device as an integer handle.  (torch.cuda.current_stream().cuda_stream
    builds a Stream object through three Python layers -- 3-4 us, a third of a small forward's host time; the raw
    accessor torch's own compiled graphs use is one C call.)"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------------------------------
def normalize_rows(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """F.normalize(x, p=2, dim=-1) -- SimilarityFunctions.py:8,11."""
    L = _ready()
    x = _f32c(x, "normalize_rows.x")
    x2 = x.reshape(-1, x.shape[-1])
    if out is None:
        out = torch.empty_like(x2)
    N.check(L.ragraph_normalize_rows_f32(x2.data_ptr(), x2.shape[0], x2.shape[1], out.data_ptr(), _stream()),
            "normalize_rows")
    return out.reshape(x.shape)


_ws_cache: dict = {}      # insertion-ordered: least recently used first
_WS_CACHE_MAX = 8


def _workspace(nbytes: int, device) -> torch.Tensor:
    """One grow-only scratch buffer per (device index, current stream OF THAT DEVICE): calls on a stream are ordered, so
    reuse is safe.  The cache is bounded (least recently used entry dropped; torch's allocator keeps a dropped buffer
    alive until the work already enqueued on its stream has used it)."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    stream = _raw_stream(dev) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream
    key = (dev, stream)
    ws = _ws_cache.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    _ws_cache[key] = ws
    while len(_ws_cache) > _WS_CACHE_MAX:
        _ws_cache.pop(next(iter(_ws_cache)))
    return ws


def pack_keys(keys_normalized: torch.Tensor) -> torch.Tensor:
    """Packed copy of the bank (row = [even k | odd k]) for the LDS-DMA ring of the tile kernel; made once per bank
    update and passed to topk_cosine(keys_packed=...)."""
    L = _ready()
    kn = _f32c(keys_normalized, "pack_keys.keys")
    out = torch.empty_like(kn)
    if kn.shape[0] == 0:
        return out
    N.check(L.ragraph_pack_keys_f32(kn.data_ptr(), kn.shape[0], kn.shape[1], out.data_ptr(), _stream()), "pack_keys")
    return out


def packed_keys_help(B: int, D: int, k: int) -> bool:
    """True when topk_cosine(B queries, k) would use a packed bank copy (tile kernel, D = 256, 4-slot ring fits)."""
    return B > 128 and D == 256 and k <= 14


FUSED_WIDTHS = (64, 128, 256)   # row widths the fused / filtered top-k kernels are written for


def padded_dim(D: int):
    """The fused kernels' width that rows of D floats are zero-padded to (None beyond 256: such banks take the score-slab
    path of ragraph_topk_cosine_f32, any D).  Zero columns change no bit of a result: a row norm's lane tree adds +0 terms,
    a score's fmaf chain ends in fmaf(0, 0, acc) = acc (acc is never -0: the chain starts from +0)."""
    for w in FUSED_WIDTHS:
        if D <= w:
            return w
    return None


def pad_cols(x: torch.Tensor, width: int) -> torch.Tensor:
    """[n, D] -> [n, width] with zero columns behind (a copy; no arithmetic)."""
    x = _f32c(x, "pad_cols.x")
    if x.shape[1] == width:
        return x
    out = x.new_zeros((x.shape[0], width))
    out[:, :x.shape[1]] = x
    return out


def topk_cosine(q: torch.Tensor, keys_normalized: torch.Tensor, k: int, idx_base: int = 0,
                keys_packed: torch.Tensor | None = None):
    """Fused normalize(q) @ keys_normalized.T -> top-k.  Returns (scores [B,k] f32, idx [B,k] i64), canonical order.

    SimilarityFunctions.py:6-16 + ToyGraphBase.py:66-67.  `keys_normalized` must come from normalize_rows();
    `keys_packed` (optional) from pack_keys(keys_normalized) -- same bits out, faster key stream for B > 128.
    Any D >= 1: widths other than 64 / 128 / 256 take materialised score slabs (dense kernel + row top-k, same bits);
    KeyIndex pads banks narrower than 256 once and keeps them on the fused kernels."""
    L = _ready()
    q = _f32c(q, "topk_cosine.q")
    kn = _f32c(keys_normalized, "topk_cosine.keys")
    kp = 0
    if keys_packed is not None:
        kpt = _f32c(keys_packed, "topk_cosine.keys_packed")
        if kpt.shape != kn.shape:
            raise RagraphNativeError(f"topk_cosine: keys_packed {tuple(kpt.shape)} != keys {tuple(kn.shape)}")
        kp = kpt.data_ptr()
    if q.dim() != 2 or kn.dim() != 2 or q.shape[1] != kn.shape[1]:
        raise RagraphNativeError(f"topk_cosine: bad shapes {tuple(q.shape)} x {tuple(kn.shape)}")
    B, D = q.shape
    Nk = kn.shape[0]
    scores = torch.empty((B, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=q.device)
    if B == 0:
        return scores, idx
    nbytes = L.ragraph_topk_cosine_workspace_bytes(B, Nk, D, k)
    ws = _workspace(nbytes, q.device)
    N.check(L.ragraph_topk_cosine_bank_f32(q.data_ptr(), B, kn.data_ptr(), kp, Nk, D, k, idx_base, scores.data_ptr(),
                                           idx.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "topk_cosine")
    return scores, idx


def keys_to_bf16(keys_normalized: torch.Tensor) -> torch.Tensor:
    """bf16 copy of the bank for topk_cosine_filtered, in MFMA fragment order (csrc/filter_common.h): rows zero-padded to
    a multiple of 256 + one row holding the largest rounding error; int16 storage, 2 D bytes per key."""
    L = _ready()
    kn = _f32c(keys_normalized, "keys_to_bf16.keys")
    rows = L.ragraph_keys_bf16_rows(kn.shape[0])
    out = torch.empty((rows, kn.shape[1]), dtype=torch.int16, device=kn.device)
    if kn.shape[0]:
        N.check(L.ragraph_keys_to_bf16(kn.data_ptr(), kn.shape[0], kn.shape[1], out.data_ptr(), _stream()), "keys_to_bf16")
    return out


def bank_copy_errors(keys_bf16: torch.Tensor, n_keys: int):
    """(max |dk| of the bf16 copy, max |dk| of the int8 copy, int8 scale) read from the copies' tail rows -- ONE
    synchronisation, for the owner of a bank version (KeyIndex) to judge once whether int8 levels suit this bank."""
    npad = -(-n_keys // 256) * 256
    t16 = keys_bf16[npad, :2].contiguous().view(torch.float32).cpu()
    t8 = keys_bf16[npad + 1 + npad // 2, :4].contiguous().view(torch.float32).cpu()
    return float(t16[0]) ** 0.5, float(t8[0]) ** 0.5, float(t8[1])


def int8_copy_classes(keys_bf16: torch.Tensor, n_keys: int) -> dict:
    """The int8 copy's two classes of granules (csrc/filter_common.h "TWO SCALES"; one synchronisation): NORMAL granules --
    32 KiB of int8 rows whose largest |k_i| is at most `cut` -- are quantised with `scale`, the HEAVY rest with
    `scale_heavy` = the bank's largest |k_i| / 127; each class has its own measured max |dk| and hence its own bound."""
    npad = -(-n_keys // 256) * 256
    row = keys_bf16[npad + 1 + npad // 2, :16].contiguous()
    f, i = row.view(torch.float32).cpu(), row.view(torch.int32).cpu()
    return {"err": float(f[0]) ** 0.5, "scale": float(f[1]), "max_abs": float(f[2]), "err_heavy": float(f[3]) ** 0.5,
            "scale_heavy": float(f[4]), "cut": float(f[5]), "heavy_granules": int(i[6]), "granules": int(i[7])}


def filtered_i8_levels(B: int, n_keys: int, D: int, k: int) -> int:
    """How many trailing levels of a filtered call of this shape run on the int8 copy (under this thread's current cap)."""
    return N.lib().ragraph_topk_cosine_filtered_i8_levels(B, n_keys, D, k)


def sharded_speculates(B: int, plan_n: int, D: int, k: int, n_shards: int) -> bool:
    """Would a sharded filtered call of this shape skip its bound pass and exchange 0 under a speculative prior?  (The same
    answer on every rank: computed from the shared plan.)"""
    return bool(N.lib().ragraph_topk_cosine_filtered_sharded_speculates(B, plan_n, D, k, n_shards))


def verify_merged_prior(merged_scores: torch.Tensor, prior, stats=None, overflow=None) -> torch.Tensor:
    """The owner's verdict on the merged lists of a sharded call (ragraph_verify_merged_prior_f32): a [5] float32 device
    tensor [missed rows, -(lowest proven k-th best), highest, candidates per query of this shard, overflowed lists]."""
    L = _ready()
    ms = _f32c(merged_scores, "verify_merged_prior.scores")
    R, k = ms.shape
    out = torch.empty(5, dtype=torch.float32, device=ms.device)
    N.check(L.ragraph_verify_merged_prior_f32(ms.data_ptr() if R else None, R, k, 0.0 if prior is None else float(prior),
                                              0 if prior is None else 1, stats.data_ptr() if stats is not None else None,
                                              overflow.data_ptr() if overflow is not None else None, out.data_ptr(), _stream()),
            "verify_merged_prior")
    return out


def set_filter_prior(theta_prior) -> float:
    """A speculative first bound for this thread's following filtered calls (None / NaN: none); returns the old one.
    ragraph_topk_cosine_filtered_set_prior: exact for any value -- queries it is too high for take the exact scan."""
    return N.lib().ragraph_topk_cosine_filtered_set_prior(float("nan") if theta_prior is None else float(theta_prior))


def ord2f(v: int) -> float:
    """The float behind an order-preserving int of the statistics words (csrc/filter_common.h f2ord: non-negative floats
    keep their bits, negative ones have the 31 low bits flipped)."""
    import struct

    v = int(v)
    if v < 0:
        v ^= 0x7FFFFFFF
    return struct.unpack("<f", struct.pack("<i", v))[0]


def set_max_i8_levels(n: int) -> int:
    """Cap the int8 levels of this thread's following filtered calls (-1: the library's rule); returns the old cap."""
    return N.lib().ragraph_topk_cosine_filtered_max_i8_levels(int(n))


def expected_i8_candidates(B: int, n_keys: int, D: int, k: int) -> float:
    """Candidates per query the schedule's cost model expects from the int8 levels of a filtered call of this shape (under
    this thread's int8 cap): ~3.9 k (level end / previous end) per level."""
    L = N.lib()
    plan = (ctypes.c_int64 * 7)()
    if L.ragraph_topk_cosine_filtered_plan(B, n_keys, D, k, plan) <= 0:
        return 0.0
    nlev = int(plan[2])
    n_i8 = L.ragraph_topk_cosine_filtered_i8_levels(B, n_keys, D, k)
    ends = [max(int(plan[0]), 1)] + [int(plan[3 + l]) for l in range(nlev)]
    return sum(3.9 * k * ends[l + 1] / ends[l] for l in range(nlev) if l >= nlev - n_i8)


def filter_helps(B: int, n_keys: int, D: int, k: int) -> bool:
    """True when the bf16-filtered exact top-k is the faster way to the same bits.  Measured on MI355X (ms, filtered vs
    fp32 kernels, D = 256, k = 10): see DESIGN.md section 4.0 (batch-size table).  Small score matrices stay on the
    fp32 slab / tile kernels."""
    if os.environ.get("RAGRAPH_EXACT_FP32") == "1":
        return False
    if D not in (64, 128, 256) or k > 32:
        return False
    # Measured on MI355X after the round-2 kernels (ms, filtered vs the fp32 dispatch; tools/quick_topk_bench.py):
    #   N >= 65536: filtered at every batch size (1 x 65536 x 256: 0.036 vs 0.055; 64 x 65536: 0.048 vs 0.080; D = 64,
    #     16 x 65536: 0.036 vs 0.056);
    #   32768 <= N < 65536: from 128 queries (128 x 32768 x 256: 0.046 vs 0.076; 64 x 32768: 0.068 vs 0.058);
    #   8192 <= N < 32768: D >= 128 from 512 queries (512 x 10000 x 128: 0.049 vs 0.056; 1024 x 10000 x 256: 0.075 vs
    #     0.114; 256 x 10000 x 128: 0.048 vs 0.042); D = 64 from 2048 (N >= 16384: 1024 x 16384: 0.089 vs 0.085) or 8192
    #     queries (2708 x 10000 x 64: 0.155 vs 0.138; 8192 x 10000: 0.284 vs 0.418).
    if n_keys >= 65536:
        return B >= FILTER_MIN_B
    if n_keys >= 32768:
        return B >= 128
    if n_keys < 8192:
        return False
    if D >= 128:
        return B >= 512
    return B >= (2048 if n_keys >= 16384 else 8192)


FILTER_MIN_B = int(os.environ.get("RAGRAPH_FILTER_MIN_B", "1"))  # banks of >= 64 k keys: filtered from this many queries


_EXCHANGE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int)

# Candidate statistics of a filtered call: the 16 ints its launches leave at the end of its workspace (include/ragraph_hip.h:
# ragraph_topk_cosine_filtered_stats_offset).  topk_cosine_filtered(..., return_stats=True) hands the caller a view of
# THAT call's words (valid until the next filtered call on the same stream reuses the workspace); there is no module-level
# "last call" state -- another thread, stream or index would overwrite it.
FILTER_STATS = True   # (KeyIndex: this ops object supports return_stats)
FILTER_STATS_MAGIC = 0x52414753


def filter_stats_levels(stats) -> list:
    """[(dtype, keys, candidates per query or None)] per level from a HOST copy of a call's statistics words."""
    st = [int(x) for x in stats]
    if len(st) < 16 or st[0] != FILTER_STATS_MAGIC:
        return []
    return [("int8" if st[8 + l] else "bf16", st[11 + l], (st[2 + l] / st[5 + l]) if st[5 + l] else None)
            for l in range(min(st[1], 3))]


def topk_cosine_filtered(q: torch.Tensor, keys_normalized: torch.Tensor, keys_bf16: torch.Tensor, k: int,
                         idx_base: int = 0, keys_packed: torch.Tensor | None = None, exchange=None, plan_n: int = 0,
                         return_stats: bool = False):
    """Exact top-k (same bits as topk_cosine) through the bf16 MFMA filter.  Returns (scores, idx, overflow): `overflow`
    is a 1-element int32 DEVICE tensor counting the queries whose candidate list overflowed (none on ordinary banks;
    all-zero queries are answered without a scan and never counted);
    the call itself recomputed those rows with an exact fp32 scan on the device, so the result is complete and nothing
    is read back: the call is asynchronous and HIP-graph capturable.  (`int(overflow)` synchronises.)
    return_stats=True: a fourth result, a [32] int32 device view of THIS call's candidate statistics (filter_stats_levels).

    Row-sharded banks: `exchange(phase, theta, scores)` is called between the phases of the call
    (ragraph_topk_cosine_filtered_sharded_f32) with theta [B] = this shard's lower bound of every query's final k-th
    best score and scores [B,k] = its running top-k; it sharpens theta in place across the shards.  `plan_n` = the
    largest shard's size (the same schedule, hence the same collectives, on every rank).  At phase 0 `scores` holds k
    lower bounds of distinct keys' exact scores (the bound pass's group maxima minus eps, or an exact level 0's top-k):
    an exchange that pools them across the shards announces the shard count as `exchange.n_shards` (default 1), and
    every shard then scans 1 / n_shards of the first sample.  The result is then the shard's list of what can still be
    in the global top-k (padded with -inf / INT64_MAX), to be merged by topk_merge."""
    L = _ready()
    q = _f32c(q, "topk_cosine_filtered.q")
    kn = _f32c(keys_normalized, "topk_cosine_filtered.keys")
    B, D = q.shape
    Nk = kn.shape[0]
    if keys_bf16.dtype != torch.int16 or not keys_bf16.is_contiguous() or keys_bf16.shape[1] != D or \
            keys_bf16.shape[0] != L.ragraph_keys_bf16_rows(Nk):
        raise RagraphNativeError("topk_cosine_filtered: keys_bf16 must come from keys_to_bf16(keys_normalized)")
    kp = 0
    if keys_packed is not None:
        kpt = _f32c(keys_packed, "topk_cosine_filtered.keys_packed")
        if kpt.shape != kn.shape:
            raise RagraphNativeError("topk_cosine_filtered: keys_packed shape mismatch")
        kp = kpt.data_ptr()
    scores = torch.empty((B, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=q.device)
    overflow = torch.empty(1, dtype=torch.int32, device=q.device)
    if exchange is None:
        nbytes = L.ragraph_topk_cosine_filtered_workspace_bytes(B, max(plan_n, Nk), D, k)
    else:  # (the sharded schedule -- planned for (plan_n, n_shards) -- can need another level-0 scratch)
        nbytes = L.ragraph_topk_cosine_filtered_sharded_workspace_bytes(B, max(plan_n, Nk), D, k,
                                                                        int(getattr(exchange, "n_shards", 1)))
    if nbytes == 0:
        raise RagraphNativeError(f"topk_cosine_filtered: unsupported shape B={B} N={Nk} D={D} k={k}")
    ws = _workspace(nbytes, q.device)
    off = L.ragraph_topk_cosine_filtered_stats_offset(ws.numel())
    stats = ws[off:off + 128].view(torch.int32)   # (32 words; valid until the next filtered call on this stream)
    if exchange is None:
        N.check(L.ragraph_topk_cosine_filtered_f32(q.data_ptr(), B, kn.data_ptr(), kp, keys_bf16.data_ptr(), Nk, D, k,
                                                   idx_base, scores.data_ptr(), idx.data_ptr(), overflow.data_ptr(), None,
                                                   ws.data_ptr(), ws.numel(), _stream()),
                "topk_cosine_filtered")
        return (scores, idx, overflow, stats) if return_stats else (scores, idx, overflow)
    theta = torch.empty(B, dtype=torch.float32, device=q.device)
    errors = []

    def _cb(_ctx, phase):  # runs on this thread, between the call's launches
        try:
            exchange(int(phase), theta, scores)
        except BaseException as e:  # (an exception must not unwind through the C frames)
            errors.append(e)

    cb = _EXCHANGE_FN(_cb)
    rc = L.ragraph_topk_cosine_filtered_sharded_f32(q.data_ptr(), B, kn.data_ptr(), kp, keys_bf16.data_ptr(), Nk, D, k,
                                                    idx_base, scores.data_ptr(), idx.data_ptr(), overflow.data_ptr(), None,
                                                    ws.data_ptr(), ws.numel(), _stream(), max(plan_n, Nk),
                                                    theta.data_ptr(), cb, None, int(getattr(exchange, "n_shards", 1)))
    if errors:
        raise errors[0]
    N.check(rc, "topk_cosine_filtered(sharded)")
    return (scores, idx, overflow, stats) if return_stats else (scores, idx, overflow)


def fused_helps(B: int, n_keys: int, D: int, k: int) -> bool:
    """True when the single-launch small-bank kernel (csrc/topk_fused.hip) is the fastest way to the exact top-k.  Its time
    is nearly flat in the batch size (38 - 95 us: a chain of per-workgroup phases, every tile streaming the L2-resident
    bf16 copy), while the score-slab / fp32 paths that small banks otherwise take grow with B.  Measured on MI355X
    (us, this kernel vs the dispatch without it; tools/quick_fused_bench.py, profiles/r3_fused_sweep.txt):
      8192 x 2000 x 64: 38 vs 136;  8192 x 5000 x 128: 76 vs 210;  8192 x 5000 x 256: 135 vs 290;  2708 x 5000 x 64: 69 vs 89;
      2708 x 20000 x 64: 83 vs 120;  1024 x 20000 x 64: 68 vs 92;  2708 x 5000 x 256: 96 vs 140;
      not: 1024 x 5000 x 128: 63 vs 52;  256 x 2000 x 64: 38 vs 25;  and banks of >= 8192 keys at D >= 128, where the
      multi-launch filtered path is as fast (2708 x 10000 x 128 -- BASELINE config 1 -- 89 vs 87; 8192 x 10000 x 128: 105 vs 107)."""
    if os.environ.get("RAGRAPH_EXACT_FP32") == "1" or os.environ.get("RAGRAPH_TOPK_FUSED", "1") == "0":
        return False
    if D not in (64, 128, 256) or k > 16 or n_keys < 128 * k:
        return False
    if 2 * n_keys * D > FUSED_MAX_COPY_BYTES or (D >= 128 and n_keys >= 8192):
        return False
    return B * n_keys >= FUSED_MIN_SCORES


FUSED_MAX_COPY_BYTES = int(os.environ.get("RAGRAPH_FUSED_MAX_COPY", str(3 << 20)))
FUSED_MIN_SCORES = int(os.environ.get("RAGRAPH_FUSED_MIN_SCORES", "12000000"))


_fused_tickets: dict = {}


def _tickets(n: int, device) -> torch.Tensor:
    """The fused kernel's per-tile tickets: zero before the first call, left zero by every call; one buffer per
    (device, stream) -- calls on a stream are ordered."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, _raw_stream(dev) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream)
    t = _fused_tickets.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 1024), dtype=torch.int32, device=device)
        _fused_tickets[key] = t
    return t


def topk_cosine_fused(q: torch.Tensor, keys_normalized: torch.Tensor, keys_bf16: torch.Tensor, k: int, idx_base: int = 0):
    """Exact top-k (same bits as topk_cosine) in ONE launch: small banks (ragraph_topk_cosine_fused_f32)."""
    L = _ready()
    q = _f32c(q, "topk_cosine_fused.q")
    kn = _f32c(keys_normalized, "topk_cosine_fused.keys")
    B, D = q.shape
    Nk = kn.shape[0]
    if keys_bf16.dtype != torch.int16 or not keys_bf16.is_contiguous() or keys_bf16.shape[1] != D or \
            keys_bf16.shape[0] != L.ragraph_keys_bf16_rows(Nk):
        raise RagraphNativeError("topk_cosine_fused: keys_bf16 must come from keys_to_bf16(keys_normalized)")
    scores = torch.empty((B, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=q.device)
    if B:
        nbytes = L.ragraph_topk_cosine_fused_workspace_bytes(B, Nk, D, k)
        if nbytes == 0:
            raise RagraphNativeError(f"topk_cosine_fused: unsupported shape B={B} N={Nk} D={D} k={k}")
        ws = _workspace(nbytes, q.device)
        N.check(L.ragraph_topk_cosine_fused_f32(q.data_ptr(), B, kn.data_ptr(), keys_bf16.data_ptr(), Nk, D, k, idx_base,
                                                scores.data_ptr(), idx.data_ptr(), _tickets((B + 31) // 32, q.device).data_ptr(),
                                                ws.data_ptr(), ws.numel(), _stream()), "topk_cosine_fused")
    return scores, idx


SMALL_MAX_B = 16   # (the kernel takes up to 32; measured against the four-launch call on the 1M x 256 bank, ms: 1 query 0.067 vs
                   # 0.093, 8: 0.073 vs 0.097, 16: 0.087 vs 0.094, 24: 0.104 vs 0.100, 32: 0.113 vs 0.096 -- profiles/r4_small_ab.txt)


def small_helps(B: int, n_keys: int, D: int, k: int) -> bool:
    """True when the single-launch kernel for a handful of queries against a large bank (csrc/topk_small.hip) is the way
    to the exact top-k: up to 16 queries, banks the filtered path takes at any batch size (>= 65536 keys).  Measured on
    MI355X, 1M x 256 bank, k = 10 (ms per call, this kernel vs the four-launch filtered call): DESIGN.md section 4.0c."""
    if os.environ.get("RAGRAPH_EXACT_FP32") == "1" or os.environ.get("RAGRAPH_TOPK_SMALL", "1") == "0":
        return False
    return 1 <= B <= SMALL_MAX_B and D in (64, 128, 256) and k <= 32 and 65536 <= n_keys < 2 ** 31


_small_state: dict = {}   # insertion-ordered: least recently used first
_small_state_pinned: set = set()   # keys whose buffer a captured HIP graph replays against
_SMALL_STATE_MAX = 8


def _small_state_buf(device, create: bool = True):
    """The single-launch kernel's state words: zero before the first call, left zero by every call; one buffer per
    (device, stream) -- calls on a stream are ordered.  Never ALLOCATED while a HIP graph is being captured (the buffer
    would come from the capture's private pool and be reused by later eager calls on a recycled stream handle): returns
    None then, and the caller takes another path (small_helps_now).  The cache is bounded like the workspace cache -- except that a
    buffer handed out DURING a capture stays for good: the graph replays against its address, and the kernel both reads it as
    zeros and zeroes it on exit."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, _raw_stream(dev) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream)
    t = _small_state.pop(key, None)
    if t is None:
        if not create or torch.cuda.is_current_stream_capturing():
            return None
        t = torch.zeros(N.lib().ragraph_topk_cosine_small_state_bytes() // 4, dtype=torch.int32, device=device)
    _small_state[key] = t
    if torch.cuda.is_current_stream_capturing():
        _small_state_pinned.add(key)   # a captured graph holds the RAW address: this buffer is never evicted (a few KB)
    for old_key in [k_ for k_ in _small_state if k_ not in _small_state_pinned][:-_SMALL_STATE_MAX]:
        _small_state.pop(old_key)      # least recently used first; pinned entries do not count against the bound
    return t


def small_state_ready(device) -> bool:
    """May the single-launch kernel run on the current stream right now?  (Always outside a capture; inside one only when
    the stream's state buffer already exists -- e.g. from the warm-up runs every capture is preceded by.)"""
    return _small_state_buf(device) is not None


def topk_cosine_small(q: torch.Tensor, keys_normalized: torch.Tensor, keys_bf16: torch.Tensor, k: int, idx_base: int = 0,
                      return_stats: bool = False):
    """Exact top-k (same bits as topk_cosine) of up to 32 queries in ONE launch (ragraph_topk_cosine_small_f32).  Returns
    (scores, idx, overflow): overflow = 1-element int32 device tensor, the queries answered by an exact scan.  Under this
    thread's speculative first bound (set_filter_prior) the launch has no bound phase and a second, normally empty launch
    scans for the queries the prior was too high for.  return_stats=True: + a [32] int32 view of the call's statistics words
    ([16] speculative, [17] misses, [18] / [19] smallest / largest k-th best score: filter_stats / ord2f)."""
    L = _ready()
    q = _f32c(q, "topk_cosine_small.q")
    kn = _f32c(keys_normalized, "topk_cosine_small.keys")
    B, D = q.shape
    Nk = kn.shape[0]
    if keys_bf16.dtype != torch.int16 or not keys_bf16.is_contiguous() or keys_bf16.shape[1] != D or \
            keys_bf16.shape[0] != L.ragraph_keys_bf16_rows(Nk):
        raise RagraphNativeError("topk_cosine_small: keys_bf16 must come from keys_to_bf16(keys_normalized)")
    scores = torch.empty((B, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=q.device)
    overflow = torch.empty(1, dtype=torch.int32, device=q.device)
    nbytes = L.ragraph_topk_cosine_small_workspace_bytes(B, D, k)
    if nbytes == 0:
        raise RagraphNativeError(f"topk_cosine_small: unsupported shape B={B} N={Nk} D={D} k={k}")
    ws = _workspace(nbytes, q.device)
    state = _small_state_buf(q.device)
    if state is None:
        raise RagraphNativeError("topk_cosine_small: no state buffer for this stream yet and a HIP graph is being captured "
                                 "(run the call once on this stream before capturing, or take topk_cosine_filtered)")
    rc = L.ragraph_topk_cosine_small_f32(q.data_ptr(), B, kn.data_ptr(), keys_bf16.data_ptr(), Nk, D, k, idx_base,
                                         scores.data_ptr(), idx.data_ptr(), overflow.data_ptr(), state.data_ptr(), ws.data_ptr(),
                                         ws.numel(), _stream())
    if rc != 0:   # (a launch that did not happen leaves the words as they were; after any error: zero them again)
        state.zero_()
    N.check(rc, "topk_cosine_small")
    if return_stats:
        return scores, idx, overflow, ws[nbytes - 128:nbytes].view(torch.int32)   # (the workspace's last 128 bytes)
    return scores, idx, overflow


def theta_sharpen(gathered: torch.Tensor, theta: torch.Tensor, k: int) -> torch.Tensor:
    """In place: theta[b] = max(theta[b], k-th largest of gathered[:, b, :]) -- the per-level exchange of a filtered
    retrieval over a row-sharded bank (gathered = all_gather of every shard's best m exact scores, [G, B, m])."""
    L = _ready()
    g = _f32c(gathered, "theta_sharpen.gathered")
    G, B, m = g.shape
    if theta.dtype != torch.float32 or not theta.is_contiguous() or theta.numel() != B:
        raise RagraphNativeError("theta_sharpen: theta must be a contiguous fp32 [B] tensor")
    N.check(L.ragraph_theta_sharpen_f32(g.data_ptr(), G, B, m, k, theta.data_ptr(), _stream()), "theta_sharpen")
    return theta


from .kernels_index import KeyIndex  # noqa: E402,F401  (bank copies + dispatch to the fastest exact top-k)


def topk_merge(scores: torch.Tensor, idx: torch.Tensor):
    """[G,B,k] per-shard lists -> canonical [B,k]."""
    L = _ready()
    scores = _f32c(scores, "topk_merge.scores")
    idx = _idxc(idx, "topk_merge.idx")
    G, B, k = scores.shape
    out_s = torch.empty((B, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((B, k), dtype=torch.int64, device=scores.device)
    N.check(L.ragraph_topk_merge_f32(scores.data_ptr(), idx.data_ptr(), G, B, k, out_s.data_ptr(), out_i.data_ptr(),
                                     _stream()), "topk_merge")
    return out_s, out_i


def dedup_rows(keys_normalized: torch.Tensor):
    """Groups of bit-identical bank rows (ragraph_dedup_rows_f32; the reference's bank recipe stores most rows several
    hundred thousand times over, ToyGraphBase.py:91-119 + Augmentation.py:9-20).  Returns (U, largest group, uniq_row [U]
    int64 = every group's lowest row, ascending; group_ptr [U+1] int32; members [N] int32 = the groups' rows, ascending).
    Reads the two counts back: ONE synchronisation, for the owner of a bank version (KeyIndex)."""
    L = _ready()
    kn = _f32c(keys_normalized, "dedup_rows.keys")
    Nk, D = kn.shape
    dev = kn.device
    stats = torch.empty(2, dtype=torch.int64, device=dev)
    uniq_row = torch.empty(Nk, dtype=torch.int64, device=dev)
    group_ptr = torch.empty(Nk + 1, dtype=torch.int32, device=dev)
    members = torch.empty(Nk, dtype=torch.int32, device=dev)
    nbytes = L.ragraph_dedup_rows_workspace_bytes(Nk)
    if nbytes == 0:
        raise RagraphNativeError(f"dedup_rows: unsupported bank of {Nk} rows")
    ws = _workspace(nbytes, dev)
    N.check(L.ragraph_dedup_rows_f32(kn.data_ptr(), Nk, D, stats.data_ptr(), uniq_row.data_ptr(), group_ptr.data_ptr(),
                                     members.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "dedup_rows")
    U, largest = (int(x) for x in stats.tolist())
    return U, largest, uniq_row[:U], group_ptr[:U + 1], members


def topk_expand_groups(scores_u: torch.Tensor, idx_u: torch.Tensor, group_ptr: torch.Tensor, members: torch.Tensor, k: int,
                       idx_base: int = 0, idx_base_u: int = 0):
    """Canonical top-k of a bank from the canonical top-ku of its unique rows (dedup_rows): [B,ku] -> [B,k], same bits
    as the search over every row (ragraph_topk_expand_groups_f32)."""
    L = _ready()
    su = _f32c(scores_u, "topk_expand_groups.scores")
    iu = _idxc(idx_u, "topk_expand_groups.idx")
    B, ku = su.shape
    out_s = torch.empty((B, k), dtype=torch.float32, device=su.device)
    out_i = torch.empty((B, k), dtype=torch.int64, device=su.device)
    N.check(L.ragraph_topk_expand_groups_f32(su.data_ptr(), iu.data_ptr(), ku, idx_base_u, group_ptr.data_ptr(),
                                             members.data_ptr(), group_ptr.numel() - 1, B, k, idx_base, out_s.data_ptr(),
                                             out_i.data_ptr(), _stream()), "topk_expand_groups")
    return out_s, out_i


def gather_rows(v: torch.Tensor, idx: torch.Tensor, idx_base: int = 0) -> torch.Tensor:
    """v[idx] -- ToyGraphBase.py:70-71.  Rows outside [idx_base, idx_base+N) come back as zeros."""
    L = _ready()
    v = _f32c(v, "gather_rows.v")
    idx = _idxc(idx, "gather_rows.idx")
    out = torch.empty(tuple(idx.shape) + (v.shape[1],), dtype=torch.float32, device=v.device)
    N.check(L.ragraph_gather_rows_f32(v.data_ptr(), v.shape[0], v.shape[1], idx.data_ptr(), idx.numel(), idx_base,
                                      out.data_ptr(), _stream()), "gather_rows")
    return out


def gather_reduce(v: torch.Tensor, labels: torch.Tensor | None, idx: torch.Tensor, idx_base: int = 0,
                  v_scale: float = 1.0):
    """(v_scale * sum_k v[idx], mean_k labels[idx]) -- RAGraph.py:48-49; edge modules/RAGraph.py:321 (v_scale=1/k)."""
    L = _ready()
    v = _f32c(v, "gather_reduce.v")
    idx = _idxc(idx, "gather_reduce.idx")
    B, k = idx.shape
    sum_v = torch.empty((B, v.shape[1]), dtype=torch.float32, device=v.device)
    mean_l = None
    C = 0
    if labels is not None:
        labels = _f32c(labels, "gather_reduce.labels")
        C = labels.shape[1]
        mean_l = torch.empty((B, C), dtype=torch.float32, device=v.device)
    N.check(L.ragraph_gather_reduce_f32(v.data_ptr(), v.shape[1], _ptr(labels), C, v.shape[0], idx.data_ptr(), B, k,
                                        idx_base, float(v_scale), sum_v.data_ptr(), _ptr(mean_l), _stream()),
            "gather_reduce")
    return sum_v, mean_l


def gather_reduce_mix(v: torch.Tensor, labels: torch.Tensor | None, idx: torch.Tensor, a: torch.Tensor, wa: float, wb: float,
                      idx_base: int = 0, v_scale: float = 1.0):
    """(a * wa + (v_scale * sum_k v[idx]) * wb, mean_k labels[idx]) -- RAGraph.py:48-49 + :53 in one launch: the bits of
    gather_reduce followed by axpby(a, wa, sum, wb), without the sum in memory."""
    L = _ready()
    v = _f32c(v, "gather_reduce_mix.v")
    idx = _idxc(idx, "gather_reduce_mix.idx")
    a = _f32c(a, "gather_reduce_mix.a")
    B, k = idx.shape
    if tuple(a.shape) != (B, v.shape[1]):
        raise RagraphNativeError(f"gather_reduce_mix: a is {tuple(a.shape)}, the reduction [{B}, {v.shape[1]}]")
    out = torch.empty((B, v.shape[1]), dtype=torch.float32, device=v.device)
    mean_l = None
    C = 0
    if labels is not None:
        labels = _f32c(labels, "gather_reduce_mix.labels")
        C = labels.shape[1]
        mean_l = torch.empty((B, C), dtype=torch.float32, device=v.device)
    N.check(L.ragraph_gather_reduce_mix_f32(v.data_ptr(), v.shape[1], _ptr(labels), C, v.shape[0], idx.data_ptr(), B, k, idx_base,
                                            float(v_scale), a.data_ptr(), float(wa), float(wb), out.data_ptr(), _ptr(mean_l),
                                            _stream()), "gather_reduce_mix")
    return out, mean_l


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, act: int = ACT_NONE,
           alpha: float = 0.0) -> torch.Tensor:
    """act(x @ weight.T + bias) -- layers/gcn.py:32, TaskDecoder.py:15-16."""
    L = _ready()
    x = _f32c(x, "linear.x")
    w = _f32c(weight, "linear.weight")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if w.dim() != 2 or w.shape[1] != x2.shape[1]:
        raise RagraphNativeError(f"linear: bad shapes {tuple(x.shape)} x {tuple(w.shape)}")
    b = None if bias is None else _f32c(bias, "linear.bias")
    y = torch.empty((x2.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
    if x2.shape[0] > 0:
        N.check(L.ragraph_linear_f32(x2.data_ptr(), x2.shape[0], x2.shape[1], w.data_ptr(), w.shape[0], _ptr(b), act,
                                     float(alpha), y.data_ptr(), _stream()), "linear")
    return y.reshape(tuple(lead) + (w.shape[0],))


def linear_tn(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a^T @ b for tall row-major operands: a [n, M], b [n, N] -> [M, N] (ragraph_linear_tn_f32: the weight gradient of a
    dense layer without transposed copies, the rows cut into ranges summed in order -- deterministic)."""
    L = _ready()
    a, b = _f32c(a, "linear_tn.a"), _f32c(b, "linear_tn.b")
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise RagraphNativeError(f"linear_tn: bad shapes {tuple(a.shape)} / {tuple(b.shape)}")
    n, M, Nn = a.shape[0], a.shape[1], b.shape[1]
    out = torch.empty((M, Nn), dtype=torch.float32, device=a.device)
    if n == 0:
        return out.zero_()
    ws = _workspace(L.ragraph_linear_tn_workspace_bytes(n, M, Nn), a.device)
    N.check(L.ragraph_linear_tn_f32(a.data_ptr(), b.data_ptr(), n, M, Nn, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
            "linear_tn")
    return out


def column_sums(x: torch.Tensor) -> torch.Tensor:
    """Sum over the rows of x [n, D] -> [D] (a bias gradient): ranges of COLSUM_ROWS rows summed sequentially by one lane
    group each (ragraph_segment_reduce_f32), then the range sums in order -- deterministic, and chip-wide where one segment
    of 100 000 rows was one workgroup's 7.5-ms chain."""
    x = _f32c(x, "column_sums.x")
    n = x.shape[0]
    if n <= COLSUM_SINGLE_MAX:   # (a batch of a few hundred nodes: one launch beats ranges + their pointer bookkeeping)
        return segment_reduce(x, torch.tensor([0, n], dtype=torch.int64, device=x.device)).reshape(-1)
    ptr = torch.arange(0, n + COLSUM_ROWS, COLSUM_ROWS, dtype=torch.int64, device=x.device).clamp_(max=n)
    part = segment_reduce(x, ptr)
    return segment_reduce(part, torch.tensor([0, part.shape[0]], dtype=torch.int64, device=x.device)).reshape(-1)


COLSUM_ROWS = 256
COLSUM_SINGLE_MAX = 2048
LINEAR_TN_MIN_ROWS = 2048   # below: gY^T X through the dense kernel on transposed copies (tiny copies, no range sums)

ROW_BLOCK = 4096  # rows / segments longer than this are summed in blocks (csrc/sparse.hip, oracle ORACLE_ROW_BLOCK)


def spmm_csr(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, x: torch.Tensor,
             bias: torch.Tensor | None = None, act: int = ACT_NONE, alpha: float = 0.0, beta: float = 0.0,
             y_in: torch.Tensor | None = None, long_rows: bool = False) -> torch.Tensor:
    """act(A @ x + bias) + beta*y_in for CSR A -- layers/gcn.py:36-40, Propagation.py:22-25, edge _agg.
    `long_rows`: the graph has rows of more than ROW_BLOCK edges (CSRGraph.has_long_rows): their blocks are spread over
    the chip through a workspace -- same bits, three more launches."""
    L = _ready()
    rowptr = _idxc(rowptr, "spmm_csr.rowptr")
    col = _idxc(col, "spmm_csr.col", torch.int32)
    val = _f32c(val, "spmm_csr.val")
    x = _f32c(x, "spmm_csr.x")
    n = rowptr.numel() - 1
    D = x.shape[1]
    b = None if bias is None else _f32c(bias, "spmm_csr.bias")
    yi = None if y_in is None else _f32c(y_in, "spmm_csr.y_in")
    y = torch.empty((n, D), dtype=torch.float32, device=x.device)
    if long_rows:
        nnz = col.numel()
        ws = _workspace(L.ragraph_sparse_workspace_bytes(nnz, D), x.device)
        N.check(L.ragraph_spmm_csr_ws_f32(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), n, x.data_ptr(), D, _ptr(b),
                                          act, float(alpha), float(beta), _ptr(yi), y.data_ptr(), nnz, ws.data_ptr(),
                                          ws.numel(), _stream()), "spmm_csr")
        return y
    N.check(L.ragraph_spmm_csr_f32(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), n, x.data_ptr(), D, _ptr(b), act,
                                   float(alpha), float(beta), _ptr(yi), y.data_ptr(), _stream()), "spmm_csr")
    return y


def spmm_csr_panels(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, x: torch.Tensor, x_panels: bool, y_panels: bool,
                    act: int = ACT_NONE, alpha: float = 0.0) -> torch.Tensor:
    """act(A @ x) with x and / or the result PANEL-major ([D/32][n][32] floats, stored as an [n, D]-sized tensor) --
    the hops of a k-hop propagation between which the features never need to be row-major (ragraph_spmm_csr_panels_f32:
    an XCD gathers from one panel; same bits as spmm_csr in every layout)."""
    L = _ready()
    rowptr = _idxc(rowptr, "spmm_csr_panels.rowptr")
    col = _idxc(col, "spmm_csr_panels.col", torch.int32)
    val = _f32c(val, "spmm_csr_panels.val")
    x = _f32c(x, "spmm_csr_panels.x")
    n, D = rowptr.numel() - 1, x.shape[1]
    y = torch.empty((n, D), dtype=torch.float32, device=x.device)
    N.check(L.ragraph_spmm_csr_panels_f32(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), n, x.data_ptr(), x.shape[0], int(x_panels),
                                          D, act, float(alpha), y.data_ptr(), int(y_panels), _stream()), "spmm_csr_panels")
    return y


def spmm_linear(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, x: torch.Tensor, weight: torch.Tensor,
                bias: torch.Tensor | None = None, act: int = ACT_NONE, alpha: float = 0.0) -> torch.Tensor:
    """act((A @ x) @ weight^T + bias) in ONE launch (ragraph_spmm_linear_f32: the aggregate-first GCN layer; x of 64 or 128
    columns): the bits of spmm_csr followed by linear, without the aggregated table in HBM."""
    L = _ready()
    rowptr = _idxc(rowptr, "spmm_linear.rowptr")
    col = _idxc(col, "spmm_linear.col", torch.int32)
    val = _f32c(val, "spmm_linear.val")
    x = _f32c(x, "spmm_linear.x")
    w = _f32c(weight, "spmm_linear.weight")
    b = _f32c(bias, "spmm_linear.bias") if bias is not None else None
    m, n_out = rowptr.numel() - 1, w.shape[0]
    if w.shape[1] != x.shape[1]:
        raise ValueError(f"spmm_linear: weight [{w.shape[0]}, {w.shape[1]}] does not take x of {x.shape[1]} columns")
    y = torch.empty((m, n_out), dtype=torch.float32, device=x.device)
    if m:
        N.check(L.ragraph_spmm_linear_f32(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), m, x.data_ptr(), x.shape[1], w.data_ptr(),
                                          n_out, b.data_ptr() if b is not None else None, act, float(alpha), y.data_ptr(), _stream()),
                 "spmm_linear")
    return y


def spmm_linear_helps(n_rows: int, f_in: int, f_out: int) -> bool:
    """The aggregate-first layer in one launch: widths the fused kernel takes, enough rows to fill the chip, one block of output
    columns (every 256-column block would aggregate its rows again).  RAGRAPH_SPMM_LINEAR=0: the two launches (A/B)."""
    return f_in in (64, 128) and f_out <= 256 and n_rows >= 4096 and os.environ.get("RAGRAPH_SPMM_LINEAR", "1") != "0"


def spmm_csr_tiled(plan, val2: torch.Tensor, x: torch.Tensor, n: int, x_panels: bool, y_panels: bool, act: int = ACT_NONE,
                   alpha: float = 0.0) -> torch.Tensor:
    """act(A @ x) over a TILED graph (ragraph_spmm_csr_tiled_f32; plan = CSRGraph.tile_plan(...), val2 = the edge values in
    plan order): the layouts of spmm_csr_panels, the same bits, the X traffic of a hop cut to (passes) x the table."""
    L = _ready()
    x = _f32c(x, "spmm_csr_tiled.x")
    D = x.shape[1]
    y = torch.empty((n, D), dtype=torch.float32, device=x.device)
    N.check(L.ragraph_spmm_csr_tiled_f32(plan.wp.data_ptr(), plan.col3.data_ptr(), val2.data_ptr(), plan.row3.data_ptr(), plan.RG,
                                         plan.C, n, x.data_ptr(), x.shape[0], int(x_panels), D, act, float(alpha),
                                         y.data_ptr(), int(y_panels), _stream()), "spmm_csr_tiled")
    return y


def tiles_help(n: int, D: int) -> bool:
    """A hop over a table much larger than the eight L2s on the graph-tiled kernel -- OPT-IN (RAGRAPH_SPMM_TILED=1: D a multiple
    of 256; =2: also D = 64 / 128).  Measured on c2's graph (tools/gnn_probe.py, us per hop, tiled / panel kernel): row-major in
    119 / 134, panel-major in 114 / 118, the GNN forward alone 0.534 / 0.545 ms; on a uniformly random graph 125 / 135
    (tools/microbench/spmm_tiled_bench.hip).  Not the default because inside the full c2 step the hops run on a side stream UNDER
    the retrieval, and a tiled workgroup holds a CU's whole LDS for the length of the launch: the filter kernel's workgroup for
    that CU starts late -- 20.54 - 20.59 ms per step on the panel kernels, 20.55 - 20.69 tiled (A/B on one box,
    profiles/r6_spmm_tiled.txt)."""
    mode = os.environ.get("RAGRAPH_SPMM_TILED", "0")
    if mode == "0" or n * D * 4 < (32 << 20):
        return False
    return (D % 256 == 0 and D <= 2048) or (mode == "2" and D in (64, 128))


def panels_help(n: int, D: int, k: int) -> bool:
    """A k-hop propagation whose intermediate features are worth keeping panel-major: at least two hops over a table much
    larger than the eight L2s (c2: 100 000 x 256 -- 102 MB; a table that fits the L2s gains nothing), D = 256 (one panel per XCD)."""
    return k >= 2 and D == 256 and n * D * 4 >= (64 << 20) and os.environ.get("RAGRAPH_SPMM_PANELS", "1") != "0"


def slices_help(n: int, D: int) -> bool:
    """A plain aggregation act(A @ x) of a NARROW row-major table much larger than the L2s (c2's encoder input: 100 000 x 128,
    51 MB) on the XCD-sliced kernel -- every XCD gathers its 128-byte slice of each neighbour row: 73.6 -> 66.8 us
    (tools/gnn_probe.py).  Not at D = 256: row-major slices 1 KiB apart fall on an eighth of an L2's channels (136 us, the
    row kernel's time; the panel-major layout between the hops is what helps there)."""
    return D == 128 and n * D * 4 >= (32 << 20) and os.environ.get("RAGRAPH_SPMM_ROW_SLICES", "1") != "0"


def csr_row_normalize(rowptr: torch.Tensor, val: torch.Tensor) -> torch.Tensor:
    """val / rowsum -- Propagation.py:15-16."""
    L = _ready()
    rowptr = _idxc(rowptr, "csr_row_normalize.rowptr")
    val = _f32c(val, "csr_row_normalize.val")
    out = torch.empty_like(val)
    N.check(L.ragraph_csr_row_normalize_f32(rowptr.data_ptr(), val.data_ptr(), rowptr.numel() - 1, out.data_ptr(),
                                            _stream()), "csr_row_normalize")
    return out


def segment_softmax(rowptr: torch.Tensor, x: torch.Tensor, long_rows: bool = False) -> torch.Tensor:
    """scatter_softmax over CSR rows -- RAGraph_edge/modules/RAGraph.py:261 (`long_rows`: see spmm_csr)."""
    L = _ready()
    rowptr = _idxc(rowptr, "segment_softmax.rowptr")
    x = _f32c(x, "segment_softmax.x")
    out = torch.zeros_like(x)
    if long_rows:
        nnz = x.numel()
        ws = _workspace(L.ragraph_sparse_workspace_bytes(nnz, 0), x.device)
        N.check(L.ragraph_segment_softmax_ws_f32(rowptr.data_ptr(), x.data_ptr(), rowptr.numel() - 1, nnz, out.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _stream()), "segment_softmax")
        return out
    N.check(L.ragraph_segment_softmax_f32(rowptr.data_ptr(), x.data_ptr(), rowptr.numel() - 1, out.data_ptr(),
                                          _stream()), "segment_softmax")
    return out


def axpby(a: torch.Tensor, wa: float, b: torch.Tensor, wb: float) -> torch.Tensor:
    """a*wa + b*wb (uncontracted) -- RAGraph.py:53."""
    L = _ready()
    a = _f32c(a, "axpby.a")
    b = _f32c(b, "axpby.b")
    if a.shape != b.shape:
        raise RagraphNativeError(f"axpby: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
    out = torch.empty_like(a)
    N.check(L.ragraph_axpby_f32(a.data_ptr(), float(wa), b.data_ptr(), float(wb), a.numel(), out.data_ptr(),
                                _stream()), "axpby")
    return out


def softmax_mix(logits: torch.Tensor, rag_label: torch.Tensor | None, lam: float = 0.0,
                log_mode: bool = False) -> torch.Tensor:
    """softmax(logits)*(1-lam) + rag_label*lam -- RAGraph.py:55-57."""
    L = _ready()
    logits = _f32c(logits, "softmax_mix.logits")
    lg = logits.reshape(-1, logits.shape[-1])
    rl = None if rag_label is None else _f32c(rag_label, "softmax_mix.rag_label").reshape(lg.shape)
    out = torch.empty_like(lg)
    N.check(L.ragraph_softmax_mix_f32(lg.data_ptr(), _ptr(rl), lg.shape[0], lg.shape[1], float(lam), int(log_mode),
                                      out.data_ptr(), _stream()), "softmax_mix")
    return out.reshape(logits.shape)


FUSE_DECODE_MAX_ROWS = 8192   # beyond: fc1 is faster on the MFMA tile kernel than in the fused launch's scalar chains
FUSE_DECODE_LDS = 64 * 1024


def fuse_decode_fits(n: int, D: int, H: int, C: int) -> bool:
    """Does ragraph_fuse_decode_f32 take this shape (and is it the faster form)?"""
    return 0 < n <= FUSE_DECODE_MAX_ROWS and 4 * 4 * ((D + 3) // 4 * 4 + (H + 3) // 4 * 4 + C) <= FUSE_DECODE_LDS


def fuse_decode(query: torch.Tensor, rag: torch.Tensor, wq: float, wr: float, W1: torch.Tensor, b1: torch.Tensor | None,
                slope: float, W2: torch.Tensor, b2: torch.Tensor | None, rag_label: torch.Tensor | None,
                lam: float) -> torch.Tensor:
    """K5 + K6 in one launch -- RAGraph_node/RAGraph.py:53-57 + TaskDecoder.py:14-17:
    softmax(fc2(LeakyReLU(fc1(query*wq + rag*wr)))) * (1-lam) + rag_label*lam.  Same bits as axpby -> linear(LEAKY) ->
    linear -> softmax_mix."""
    L = _ready()
    query, rag = _f32c(query, "fuse_decode.query"), _f32c(rag, "fuse_decode.rag")
    W1, W2 = _f32c(W1, "fuse_decode.W1"), _f32c(W2, "fuse_decode.W2")
    b1 = None if b1 is None else _f32c(b1, "fuse_decode.b1")
    b2 = None if b2 is None else _f32c(b2, "fuse_decode.b2")
    n, D = query.shape
    H, C = W1.shape[0], W2.shape[0]
    if rag.shape != query.shape or W1.shape[1] != D or W2.shape[1] != H:
        raise ValueError(f"fuse_decode: shapes query {tuple(query.shape)} rag {tuple(rag.shape)} W1 {tuple(W1.shape)} "
                         f"W2 {tuple(W2.shape)}")
    rl = None if rag_label is None else _f32c(rag_label, "fuse_decode.rag_label").reshape(n, C)
    out = torch.empty((n, C), dtype=torch.float32, device=query.device)
    N.check(L.ragraph_fuse_decode_f32(query.data_ptr(), rag.data_ptr(), n, D, float(wq), float(wr), W1.data_ptr(),
                                      _ptr(b1), H, float(slope), W2.data_ptr(), _ptr(b2), C, _ptr(rl), float(lam),
                                      out.data_ptr(), _stream()), "fuse_decode")
    return out


def segment_reduce(x: torch.Tensor, seg_ptr: torch.Tensor, w: torch.Tensor | None = None,
                   mean_mode: bool = False) -> torch.Tensor:
    """Per-segment sum (or mean) of rows, optionally of w*x -- RAGraph_graph/RAGraph.py:50,63; downprompt.py:98-112."""
    L = _ready()
    x = _f32c(x, "segment_reduce.x")
    seg_ptr = _idxc(seg_ptr, "segment_reduce.seg_ptr")
    wv = None if w is None else _f32c(w, "segment_reduce.w").reshape(-1)
    G = seg_ptr.numel() - 1
    out = torch.empty((G, x.shape[1]), dtype=torch.float32, device=x.device)
    N.check(L.ragraph_segment_reduce_f32(x.data_ptr(), x.shape[1], seg_ptr.data_ptr(), G, _ptr(wv), int(mean_mode),
                                         out.data_ptr(), _stream()), "segment_reduce")
    return out


def proto_cosine(emb: torch.Tensor, proto: torch.Tensor, mode: int = 0) -> torch.Tensor:
    """cosine(emb_g, proto_c) (+softmax / log_softmax) -- downprompt.py:41-56."""
    L = _ready()
    emb = _f32c(emb, "proto_cosine.emb")
    proto = _f32c(proto, "proto_cosine.proto")
    G, D = emb.shape
    C = proto.shape[0]
    out = torch.empty((G, C), dtype=torch.float32, device=emb.device)
    N.check(L.ragraph_proto_cosine_f32(emb.data_ptr(), G, D, proto.data_ptr(), C, mode, out.data_ptr(), _stream()),
            "proto_cosine")
    return out


def proto_cosine_grad(emb: torch.Tensor, proto: torch.Tensor, mode: int, out: torch.Tensor, gout: torch.Tensor):
    """Gradient of proto_cosine with respect to `emb` (prototypes constant)."""
    L = _ready()
    emb, proto = _f32c(emb, "proto_cosine_grad.emb"), _f32c(proto, "proto_cosine_grad.proto")
    out, gout = _f32c(out, "proto_cosine_grad.out"), _f32c(gout, "proto_cosine_grad.gout")
    G, D = emb.shape
    gemb = torch.empty_like(emb)
    N.check(L.ragraph_proto_cosine_grad_f32(emb.data_ptr(), G, D, proto.data_ptr(), proto.shape[0], mode, out.data_ptr(),
                                            gout.data_ptr(), gemb.data_ptr(), _stream()), "proto_cosine_grad")
    return gemb


def proto_cosine_grad_proto(emb: torch.Tensor, proto: torch.Tensor, mode: int, out: torch.Tensor, gout: torch.Tensor):
    """Gradient of proto_cosine with respect to the PROTOTYPES [C, D] (a training step that keeps them in the graph:
    RAGraph_node/downprompt.py:24-25); sums in a fixed order."""
    L = _ready()
    emb, proto = _f32c(emb, "proto_cosine_grad_proto.emb"), _f32c(proto, "proto_cosine_grad_proto.proto")
    out, gout = _f32c(out, "proto_cosine_grad_proto.out"), _f32c(gout, "proto_cosine_grad_proto.gout")
    G, D = emb.shape
    C = proto.shape[0]
    nbytes = int(L.ragraph_proto_cosine_grad_proto_workspace_bytes(G, C, D))
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=emb.device)
    gproto = torch.empty_like(proto)
    N.check(L.ragraph_proto_cosine_grad_proto_f32(emb.data_ptr(), G, D, proto.data_ptr(), C, mode, out.data_ptr(), gout.data_ptr(),
                                                  gproto.data_ptr(), ws.data_ptr(), nbytes, _stream()), "proto_cosine_grad_proto")
    return gproto


def axpby_dev(a: torch.Tensor, b: torch.Tensor, w: torch.Tensor, ia: int, ib: int) -> torch.Tensor:
    """a * w[ia] + b * w[ib], the weights read on the device (an index < 0: weight 0) -- RAGraph_node/downprompt.py:113."""
    L = _ready()
    a, b = _f32c(a, "axpby_dev.a"), _f32c(b, "axpby_dev.b")
    w = _f32c(w, "axpby_dev.w").reshape(-1)
    if a.shape != b.shape or max(ia, ib) >= w.numel():
        raise RagraphNativeError(f"axpby_dev: shapes {tuple(a.shape)} / {tuple(b.shape)}, weights {w.numel()}")
    out = torch.empty_like(a)
    N.check(L.ragraph_axpby_dev_f32(a.data_ptr(), b.data_ptr(), w.data_ptr(), int(ia), int(ib), a.numel(), out.data_ptr(),
                                    _stream()), "axpby_dev")
    return out


def topk_rows(scores: torch.Tensor, k: int):
    """torch.topk(scores, k) over a materialised [B,N] matrix, canonical tie order -- few-shot retrieve
    (RAGraph_node_fewshot/.../ToyGraphBase.py:64), edge evaluation (RAGraph_edge/utils/metrics.py:116)."""
    L = _ready()
    s = _f32c(scores, "topk_rows.scores")
    if s.dim() != 2:
        raise RagraphNativeError(f"topk_rows: expected [B,N], got {tuple(s.shape)}")
    B, Nn = s.shape
    out_s = torch.empty((B, k), dtype=torch.float32, device=s.device)
    out_i = torch.empty((B, k), dtype=torch.int64, device=s.device)
    N.check(L.ragraph_topk_rows_f32(s.data_ptr(), B, Nn, Nn, k, out_s.data_ptr(), out_i.data_ptr(), _stream()),
            "topk_rows")
    return out_s, out_i


def topk_select_rows(scores: torch.Tensor, k: int):
    """The canonical top-k SET of every row for any k <= N: (kth [B], idx [B,k] in ascending index order) -- the edge
    flavour's vanilla-phase retrieval (RAGraph_edge/modules/RAGraph.py:57,73,308-321), which only averages the winners."""
    L = _ready()
    s = _f32c(scores, "topk_select_rows.scores")
    if s.dim() != 2:
        raise RagraphNativeError(f"topk_select_rows: expected [B,N], got {tuple(s.shape)}")
    B, Nn = s.shape
    kth = torch.empty(B, dtype=torch.float32, device=s.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=s.device)
    if Nn >= 65536 and B <= 65535:  # long rows: every pass of the selection spread over (chunk, row) workgroups
        ws = _workspace(L.ragraph_topk_select_rows_workspace_bytes(B, Nn), s.device)
        N.check(L.ragraph_topk_select_rows_ws_f32(s.data_ptr(), B, Nn, Nn, k, kth.data_ptr(), idx.data_ptr(), ws.data_ptr(),
                                                  ws.numel(), _stream()), "topk_select_rows")
        return kth, idx
    N.check(L.ragraph_topk_select_rows_f32(s.data_ptr(), B, Nn, Nn, k, kth.data_ptr(), idx.data_ptr(), _stream()),
            "topk_select_rows")
    return kth, idx


def retrieve_mean_large_k(q: torch.Tensor, keys_normalized: torch.Tensor, values: torch.Tensor, k: int,
                          slab_bytes: int = 1 << 30) -> torch.Tensor:
    """mean_k V[top-k(q)] for k beyond the fused kernels' lists (k > 64): score slabs by the dense kernel (the same
    fmaf chains as every other path), topk_select_rows, and the winners' sum in ascending index order times 1/k.
    RAGraph_edge/modules/RAGraph.py:298-324 with retrieve_num = 50 ... 100000.
    Up to ROW_BLOCK winners the sum is gather_reduce's sequential chain.  Beyond (retrieve_num = 100000: one lane group
    walking 100 k rows one after the other took 87 ms per 64 queries) the winners of a query are a CSR row of ones and the
    sum is the SpMM's hub-row path: blocks of ROW_BLOCK consecutive winners, each its own chain, spread over the chip,
    the block sums added in order -- the blocked order of oracle_spmm_csr, deterministic."""
    q = _f32c(q, "retrieve_mean_large_k.q")
    kn = _f32c(keys_normalized, "retrieve_mean_large_k.keys")
    values = _f32c(values, "retrieve_mean_large_k.values")
    B, Nk = q.shape[0], kn.shape[0]
    qn = normalize_rows(q)
    rows = max(1, min(B, slab_bytes // (4 * Nk)))
    out = torch.empty((B, values.shape[1]), dtype=torch.float32, device=q.device)
    ones = rowptr = None
    for b0 in range(0, B, rows):
        S = linear(qn[b0:b0 + rows], kn)
        _, idx = topk_select_rows(S, k)
        nb = idx.shape[0]
        if k <= ROW_BLOCK:
            out[b0:b0 + nb], _ = gather_reduce(values, None, idx, v_scale=1.0 / k)
            continue
        if ones is None or ones.numel() != nb * k:
            ones = torch.ones(nb * k, dtype=torch.float32, device=q.device)
            rowptr = torch.arange(0, (nb + 1) * k, k, dtype=torch.int64, device=q.device)
        total = spmm_csr(rowptr, idx.reshape(-1).to(torch.int32), ones, values, long_rows=True)
        out[b0:b0 + nb] = axpby(total, 1.0 / k, total, 0.0)
    return out


def scatter_fill_(scores: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, value: float) -> torch.Tensor:
    """In place: scores[b, col[rowptr[b]:rowptr[b+1]]] = value -- metrics.py:210-214 (_mask_history_pos)."""
    L = _ready()
    if not (scores.is_cuda and scores.dtype == torch.float32 and scores.is_contiguous() and scores.dim() == 2):
        raise RagraphNativeError("scatter_fill_: scores must be a contiguous fp32 [B,N] device tensor")
    rowptr = _idxc(rowptr, "scatter_fill_.rowptr")
    col = _idxc(col, "scatter_fill_.col")
    B, Nn = scores.shape
    N.check(L.ragraph_scatter_fill_f32(scores.data_ptr(), B, Nn, Nn, rowptr.data_ptr(), col.data_ptr(), float(value),
                                       _stream()), "scatter_fill")
    return scores


def floyd_warshall(adj_dense: torch.Tensor) -> torch.Tensor:
    """All-pairs shortest paths with the adjacency VALUES as edge lengths -- PositionAwareEncoder.py:27-48."""
    L = _ready()
    a = _f32c(adj_dense, "floyd_warshall.adj")
    n = a.shape[0]
    d = torch.empty_like(a)
    N.check(L.ragraph_floyd_warshall_f32(a.data_ptr(), n, d.data_ptr(), _stream()), "floyd_warshall")
    return d


def position_code(dist: torch.Tensor, anchors: torch.Tensor, dis_q: float = 10.0) -> torch.Tensor:
    """1/(d+1) if d < dis_q else 0 to each anchor -- PositionAwareEncoder.py:6-24."""
    L = _ready()
    d = _f32c(dist, "position_code.dist")
    anchors = _idxc(anchors, "position_code.anchors")
    out = torch.empty((d.shape[0], anchors.numel()), dtype=torch.float32, device=d.device)
    N.check(L.ragraph_position_code_f32(d.data_ptr(), d.shape[0], anchors.data_ptr(), anchors.numel(), float(dis_q),
                                        out.data_ptr(), _stream()), "position_code")
    return out


def position_codes_csr(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, anchors: torch.Tensor,
                       dis_q: float = 10.0, return_dist: bool = False):
    """Position-aware codes [n, A] from the CSR of the query batch: shortest-path distances to the A anchors only (one
    workgroup per anchor, no n x n matrix, one launch) -- PositionAwareEncoder.py:6-24 as the few-shot retrieve uses it
    on every forward (RAGraph_node_fewshot/ragraph_utils/ToyGraphBase.py:49-50)."""
    L = _ready()
    rowptr = _idxc(rowptr, "position_codes_csr.rowptr")
    col = _idxc(col, "position_codes_csr.col", torch.int32)
    val = _f32c(val, "position_codes_csr.val")
    anchors = _idxc(anchors, "position_codes_csr.anchors").reshape(-1)
    n, A = rowptr.numel() - 1, anchors.numel()
    codes = torch.empty((n, A), dtype=torch.float32, device=val.device)
    dist = torch.empty((n, A), dtype=torch.float32, device=val.device) if return_dist else None
    N.check(L.ragraph_position_codes_csr_f32(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), n, anchors.data_ptr(), A,
                                             float(dis_q), codes.data_ptr(), _ptr(dist), _stream()), "position_codes_csr")
    return (codes, dist) if return_dist else codes


def sigmoid_gate(x: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(z) -- the embedding gate of RAGraph_edge/modules/RAGraph.py:168."""
    L = _ready()
    x = _f32c(x, "sigmoid_gate.x")
    z = _f32c(z, "sigmoid_gate.z")
    out = torch.empty_like(x)
    N.check(L.ragraph_sigmoid_gate_f32(x.data_ptr(), z.data_ptr(), x.numel(), out.data_ptr(), _stream()), "sigmoid_gate")
    return out


def time_rescale(t: torch.Tensor, t_min: float, t_max: float) -> torch.Tensor:
    """(t - t_min) / (t_max - t_min) on int64 time steps -- RAGraph_edge/modules/RAGraph.py:254-257."""
    L = _ready()
    t = _idxc(t, "time_rescale.t")
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    N.check(L.ragraph_time_rescale_f32(t.data_ptr(), t.numel(), float(t_min), float(t_max), out.data_ptr(), _stream()),
            "time_rescale")
    return out


# ---- toy-bank construction --------------------------------------------------------------
def csr_row_sums(rowptr: torch.Tensor, val: torch.Tensor) -> torch.Tensor:
    """Row sums of a CSR matrix (torch.sum(adj, dim=1); dim=0 on the transposed CSR) -- InverseSampling.py:25,53."""
    L = _ready()
    rowptr = _idxc(rowptr, "csr_row_sums.rowptr")
    val = _f32c(val, "csr_row_sums.val")
    out = torch.empty(rowptr.numel() - 1, dtype=torch.float32, device=val.device)
    N.check(L.ragraph_csr_row_sums_f32(rowptr.data_ptr(), val.data_ptr(), out.numel(), out.data_ptr(), _stream()),
            "csr_row_sums")
    return out


def pagerank(rowptr_t: torch.Tensor, col_t: torch.Tensor, val_t: torch.Tensor, out_deg: torch.Tensor,
             graph_ptr: torch.Tensor, d: float = 0.85, eps: float = 1e-6, max_iter: int = 128):
    """InverseSampling.pagerank_algorithm for a batch of graphs (segments of one block-diagonal transposed CSR), all
    power iterations enqueued without a host round trip.  Returns (p [n], iters [G] int32)."""
    L = _ready()
    rowptr_t = _idxc(rowptr_t, "pagerank.rowptr")
    col_t = _idxc(col_t, "pagerank.col", torch.int32)
    val_t = _f32c(val_t, "pagerank.val")
    out_deg = _f32c(out_deg, "pagerank.out_deg")
    graph_ptr = _idxc(graph_ptr, "pagerank.graph_ptr")
    n, G = out_deg.numel(), graph_ptr.numel() - 1
    sizes = graph_ptr[1:] - graph_ptr[:-1]
    graph_of = torch.repeat_interleave(torch.arange(G, device=out_deg.device, dtype=torch.int32), sizes)
    p = torch.empty(n, dtype=torch.float32, device=out_deg.device)
    iters = torch.empty(G, dtype=torch.int32, device=out_deg.device)
    ws = _workspace(L.ragraph_pagerank_workspace_bytes(n, G), out_deg.device)
    N.check(L.ragraph_pagerank_f32(rowptr_t.data_ptr(), col_t.data_ptr(), val_t.data_ptr(), out_deg.data_ptr(),
                                   graph_ptr.data_ptr(), graph_of.data_ptr(), G, n, float(d), float(eps), int(max_iter),
                                   p.data_ptr(), iters.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "pagerank")
    return p, iters


def sample_prob(pagerank_p: torch.Tensor, col_sum: torch.Tensor, graph_ptr: torch.Tensor, alpha: float = 0.5,
                eps: float = 1e-6) -> torch.Tensor:
    """InverseSampling.compute_sample_prob (:6-19): inverse-importance sampling probabilities per graph."""
    L = _ready()
    pr = _f32c(pagerank_p, "sample_prob.pagerank")
    cs = _f32c(col_sum, "sample_prob.col_sum")
    graph_ptr = _idxc(graph_ptr, "sample_prob.graph_ptr")
    out = torch.empty_like(pr)
    N.check(L.ragraph_sample_prob_f32(pr.data_ptr(), cs.data_ptr(), graph_ptr.data_ptr(), graph_ptr.numel() - 1, float(alpha),
                                      float(eps), out.data_ptr(), _stream()), "sample_prob")
    return out


def position_codes_batch(adj: torch.Tensor, anchors: torch.Tensor, dis_q: float = 10.0, return_dist: bool = False):
    """PositionAwareEncoder.encode_position_aware_code for G graphs of n <= 64 nodes: adj [G,n,n], anchors [G,A] ->
    codes [G,n,A] (and the all-pairs distances [G,n,n])."""
    L = _ready()
    a = _f32c(adj, "position_codes_batch.adj")
    anchors = _idxc(anchors, "position_codes_batch.anchors")
    G, n, _ = a.shape
    A = anchors.shape[1]
    codes = torch.empty((G, n, A), dtype=torch.float32, device=a.device)
    dist = torch.empty_like(a) if return_dist else None
    N.check(L.ragraph_position_codes_batch_f32(a.data_ptr(), G, n, anchors.data_ptr(), A, float(dis_q), _ptr(dist),
                                               codes.data_ptr(), _stream()), "position_codes_batch")
    return (codes, dist) if return_dist else codes


# ---- graph ingestion --------------------------------------------------------------------
def csr_sym_normalized_from_edges(edge_index: torch.Tensor, n: int):
    """D^-1/2 (A + I) D^-1/2 as CSR from an edge list [2,E] -- ragraph_utils/utility.py:19-26,45-66.  Returns (rowptr
    int64 [n+1], col int32 [nnz], val fp32 [nnz]).  Reads the entry count back (one synchronisation: ingestion is a
    per-graph step, not part of the captured forward)."""
    L = _ready()
    ei = _idxc(edge_index, "csr_sym_normalized.edge_index")
    E = ei.shape[1]
    dev = ei.device
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    col = torch.empty(E + n, dtype=torch.int32, device=dev)
    val = torch.empty(E + n, dtype=torch.float32, device=dev)
    nnz = torch.empty(1, dtype=torch.int64, device=dev)
    ws = _workspace(L.ragraph_ingest_workspace_bytes(E + n, n), dev)
    N.check(L.ragraph_csr_sym_normalized_f32(ei[0].data_ptr() if E else None, ei[1].data_ptr() if E else None, E, n,
                                             rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), nnz.data_ptr(), ws.data_ptr(),
                                             ws.numel(), _stream()), "csr_sym_normalized")
    m = int(nnz.item())
    return rowptr, col[:m].contiguous(), val[:m].contiguous()


def coo_to_csr(rows: torch.Tensor, cols: torch.Tensor, n: int, sort_cols: bool = False):
    """COO -> CSR, stable (ragraph_coo_to_csr_i64): returns (rowptr int64 [n+1], col int32 [E] in CSR order, perm int64 [E] =
    the input position of every CSR slot).  No read-back, no other library: the per-step graph rebuilds of the edge flavour,
    the transposed patterns of training and the gather backward come through here."""
    L = _ready()
    r, c = _idxc(rows, "coo_to_csr.rows"), _idxc(cols, "coo_to_csr.cols")
    E = r.numel()
    if c.numel() != E:
        raise RagraphNativeError("coo_to_csr: rows and cols differ in length")
    dev = r.device
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    perm = torch.empty(E, dtype=torch.int64, device=dev)
    col = torch.empty(E, dtype=torch.int32, device=dev)
    ws = _workspace(L.ragraph_coo_to_csr_workspace_bytes(E, n), dev)
    N.check(L.ragraph_coo_to_csr_i64(r.data_ptr() if E else None, c.data_ptr() if E else None, E, n, 1 if sort_cols else 0,
                                     rowptr.data_ptr(), perm.data_ptr() if E else None, col.data_ptr() if E else None,
                                     ws.data_ptr(), ws.numel(), _stream()), "coo_to_csr")
    return rowptr, col, perm


def csr_row_ids(rowptr: torch.Tensor, nnz: int) -> torch.Tensor:
    """rows[e] = the row of CSR slot e (ragraph_csr_row_ids_i64)."""
    L = _ready()
    rp = _idxc(rowptr, "csr_row_ids.rowptr")
    rows = torch.empty(nnz, dtype=torch.int64, device=rp.device)
    N.check(L.ragraph_csr_row_ids_i64(rp.data_ptr(), rp.numel() - 1, nnz, rows.data_ptr() if nnz else None, _stream()),
            "csr_row_ids")
    return rows


def mask_positions(mask: torch.Tensor) -> torch.Tensor:
    """Positions of the set elements of a bool / uint8 mask, ascending (ragraph_mask_positions_i64) -- x[mask] as
    x[mask_positions(mask)] without another library's select.  Reads the count back (one synchronisation, as boolean
    indexing does)."""
    L = _ready()
    if not mask.is_cuda or mask.dtype not in (torch.bool, torch.uint8):
        raise RagraphNativeError("mask_positions: expected a bool / uint8 ROCm device tensor")
    m = mask.contiguous().view(torch.uint8).reshape(-1)
    E = m.numel()
    pos = torch.empty(E, dtype=torch.int64, device=m.device)
    count = torch.empty(1, dtype=torch.int64, device=m.device)
    ws = _workspace(L.ragraph_mask_positions_workspace_bytes(E), m.device)
    N.check(L.ragraph_mask_positions_i64(m.data_ptr() if E else None, E, pos.data_ptr() if E else None, count.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _stream()), "mask_positions")
    return pos[:int(count.item())]


def binorm_edges(users: torch.Tensor, items: torch.Tensor, step: torch.Tensor, num_users: int, num_items: int):
    """Bi-normalised bipartite adjacency as a (dst, src)-sorted edge list with per-edge time steps --
    RAGraph_edge/modules/base_model.py:34-52 + utils/dataloader.py:94,108-113.  Returns (edges [M,2] int64, norm [M],
    times [M] int64)."""
    L = _ready()
    u, i, t = (_idxc(x, "binorm_edges") for x in (users, items, step))
    E = u.numel()
    dev = u.device
    edges = torch.empty((2 * E, 2), dtype=torch.int64, device=dev)
    norm = torch.empty(2 * E, dtype=torch.float32, device=dev)
    times = torch.empty(2 * E, dtype=torch.int64, device=dev)
    ne = torch.empty(1, dtype=torch.int64, device=dev)
    ws = _workspace(L.ragraph_ingest_workspace_bytes(2 * E, num_users + num_items), dev)
    N.check(L.ragraph_binorm_edges_f32(u.data_ptr(), i.data_ptr(), t.data_ptr(), E, num_users, num_items, edges.data_ptr(),
                                       norm.data_ptr(), times.data_ptr(), ne.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
            "binorm_edges")
    m = int(ne.item())
    return edges[:m].contiguous(), norm[:m].contiguous(), times[:m].contiguous()


# ---- element-wise derivatives of the fused epilogues (fine-tuning backward) -------------------------------------------
def act_grad(y: torch.Tensor, gy: torch.Tensor, act: int, alpha: float = 0.0, want_alpha_terms: bool = False):
    """gz = gy * act'(z) through the output y; with want_alpha_terms also gy * z on z < 0 (PReLU slope gradient terms)."""
    L = _ready()
    y, gy = _f32c(y, "act_grad.y"), _f32c(gy, "act_grad.gy")
    gz = torch.empty_like(y)
    t = torch.empty_like(y) if want_alpha_terms else None
    N.check(L.ragraph_act_grad_f32(y.data_ptr(), gy.data_ptr(), y.numel(), act, float(alpha), gz.data_ptr(), _ptr(t), _stream()),
            "act_grad")
    return (gz, t) if want_alpha_terms else gz


def sigmoid_gate_grad(x: torch.Tensor, z: torch.Tensor, g: torch.Tensor):
    L = _ready()
    x, z, g = _f32c(x, "sigmoid_gate_grad.x"), _f32c(z, "sigmoid_gate_grad.z"), _f32c(g, "sigmoid_gate_grad.g")
    gx, gz = torch.empty_like(x), torch.empty_like(x)
    N.check(L.ragraph_sigmoid_gate_grad_f32(x.data_ptr(), z.data_ptr(), g.data_ptr(), x.numel(), gx.data_ptr(), gz.data_ptr(),
                                            _stream()), "sigmoid_gate_grad")
    return gx, gz


def softmax_grad(p: torch.Tensor, go: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    L = _ready()
    p, go = _f32c(p, "softmax_grad.p"), _f32c(go, "softmax_grad.go")
    p2 = p.reshape(-1, p.shape[-1])
    out = torch.empty_like(p2)
    N.check(L.ragraph_softmax_grad_f32(p2.data_ptr(), go.reshape(p2.shape).data_ptr(), p2.shape[0], p2.shape[1], float(scale),
                                       out.data_ptr(), _stream()), "softmax_grad")
    return out.reshape(p.shape)


def mul_cols(x: torch.Tensor, w: torch.Tensor, act: int = ACT_NONE, alpha: float = 0.0) -> torch.Tensor:
    """act(x[r,:] * w) -- downstreamprompt.forward: plain in the graph flavour (RAGraph_graph/downprompt.py:164-168), ELU
    in the node flavour (RAGraph_node/downprompt.py:118-130)."""
    L = _ready()
    x = _f32c(x, "mul_cols.x")
    w = _f32c(w, "mul_cols.w").reshape(-1)
    if w.numel() != x.shape[-1]:
        raise RagraphNativeError(f"mul_cols: weight of {w.numel()} values for rows of {x.shape[-1]}")
    x2 = x.reshape(-1, x.shape[-1])
    out = torch.empty_like(x2)
    N.check(L.ragraph_mul_cols_act_f32(x2.data_ptr(), w.data_ptr(), x2.shape[0], x2.shape[1], int(act), float(alpha),
                                       out.data_ptr(), _stream()), "mul_cols")
    return out.reshape(x.shape)


def mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a * b elementwise (same shape)."""
    L = _ready()
    a, b = _f32c(a, "mul.a"), _f32c(b, "mul.b")
    if a.shape != b.shape:
        raise RagraphNativeError(f"mul: shapes {tuple(a.shape)} and {tuple(b.shape)} differ")
    out = torch.empty_like(a)
    N.check(L.ragraph_mul_f32(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), _stream()), "mul")
    return out
