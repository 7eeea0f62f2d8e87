#!/usr/bin/env python3
"""Copy what tools/gpu_round_<round>.sh left under gpurun_out/<tag>/ into profiles/<round>_* (the files the docs and bench.py
cite) and rebuild profiles/<round>_pmc_traffic.json from the PMC summaries.   python tools/collect_profiles.py r5"""
import csv, json, os, re, shutil, sys

tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles")
P = tag[:2] if tag[:1] == "r" and tag[1:2].isdigit() else "r3"   # r3 / r4: the round prefix of the profiles


def cp(a, b):
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))


cp("kernel_stats.csv", f"{P}_bench_c2_kernel_stats.csv")
cp("trace_summary.txt", f"{P}_bench_c2_trace_summary.txt")
with open(os.path.join(dst, f"{P}_bench_c2_pmc_traffic.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE (first block) and, separate run, --pmc WRITE_SIZE (second block) of\n"
            "# `python3 bench.py --steps 2 --warmup 1 --no-extras` on MI355X; KB per dispatch, mean and max (tools/pmc_summary.py)\n")
    f.write(open(os.path.join(src, "pmc_fetch.txt")).read())
    f.write(open(os.path.join(src, "pmc_write.txt")).read())
line = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(line, open(os.path.join(dst, f"{P}_bench_line.json"), "w"))
SMALL = (1, 16, 256, 512, 4096)
for B in SMALL:
    cp(f"smallb_B{B}_kernel_stats.csv", f"{P}_smallb_B{B}_kernel_stats.csv")
    cp(f"smallb_B{B}_pmc.txt", f"{P}_smallb_B{B}_pmc_traffic.txt")


def pmc(path):
    out, cur = {}, None
    for l in open(path):
        if not l.startswith(" "):
            cur = l.strip()
        else:
            m = re.match(r"\s+(\w+)\s+mean/dispatch\s+([0-9.]+)\s+\(n=(\d+)\)(?:\s+max/dispatch\s+([0-9.]+))?", l)
            if m:
                out.setdefault(cur, {})[m.group(1)] = float(m.group(2))
                out[cur][m.group(1) + "_n"] = int(m.group(3))
                if m.group(4):
                    out[cur][m.group(1) + "_max"] = float(m.group(4))
    return out


def kernel_us(path, name):
    for r in csv.DictReader(open(path)):
        if name in r["Name"]:
            return float(r["AverageNs"]) / 1e3
    return None


fe, wr = pmc(os.path.join(src, "pmc_fetch.txt")), pmc(os.path.join(src, "pmc_write.txt"))
out = {
    "_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate run, --pmc WRITE_SIZE on MI355X "
            "(tools/gpu_round_<round>.sh, collected by tools/collect_profiles.py; summaries profiles/<round>_bench_c2_pmc_traffic.txt, "
            "profiles/<round>_smallb_B*_pmc_traffic.txt); values are KB per dispatch. HBM-side bytes = (2*FETCH_SIZE + "
            "WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM "
            "section). Infinity-Cache hits are included in FETCH_SIZE.",
}
def find(kind):
    """The filter kernel's PMC name for a kind of launch, whatever its queries-per-wave template argument."""
    for nm in fe:
        m = re.match(r"ragraph::topk_filter_kernel<256, (\d+), (true|false), (true|false)(?:, (?:true|false))*>", nm)
        if m and (m.group(2), m.group(3)) == {"int8": ("false", "true"), "bf16": ("false", "false"), "bound": ("true", "false")}[kind]:
            return nm
    return None


names = {k_: find(k_) for k_ in ("int8", "bf16", "bound")}
rec, total_kb = {}, 0.0
for key, nm in names.items():
    if nm and nm in fe and nm in wr:
        per_call = fe[nm]["FETCH_SIZE_n"] / max(fe[names["bound"]]["FETCH_SIZE_n"], 1) if names["bound"] in fe else 1
        rec[key] = {"FETCH_SIZE_KB_mean": fe[nm]["FETCH_SIZE"], "WRITE_SIZE_KB_mean": wr[nm]["WRITE_SIZE"],
                    "FETCH_SIZE_KB_max": fe[nm].get("FETCH_SIZE_max"), "WRITE_SIZE_KB_max": wr[nm].get("WRITE_SIZE_max"),
                    "dispatches_per_call": per_call}
        total_kb += per_call * (2 * fe[nm]["FETCH_SIZE"] + wr[nm]["WRITE_SIZE"])
dom = rec.get("int8") or rec.get("bf16")
if dom:
    dom_kb = 2 * (dom["FETCH_SIZE_KB_max"] or dom["FETCH_SIZE_KB_mean"]) + (dom["WRITE_SIZE_KB_max"] or dom["WRITE_SIZE_KB_mean"])
    out["topk_filter_kernel B=100000 N=1000000 D=256 k=10"] = {
        "per_kernel": rec,
        "hbm_side_GB": round(dom_kb * 1024 / 1e9, 2),
        "hbm_side_GB_whole_call": round(total_kb * 1024 / 1e9, 2),
        "note": "hbm_side_GB = the launch that takes longest (the last level, on the int8 copy: the largest dispatch of that "
                "kernel); _whole_call = bound pass + every level. Algorithmic minimum of the last level: 3/4 of the int8 copy "
                "once (0.19 GB) + the queries (0.1 GB); the rest are re-reads of the key stream by the 8 XCD groups (each streams "
                "the bank for its ~24 query tiles of 512), L2 misses served by the Infinity Cache / HBM -- a few % of HBM "
                "bandwidth: the kernel is MFMA-bound."}
for B in SMALL:
    f1 = pmc(os.path.join(src, f"smallb_B{B}_pmc.txt"))
    for nm, d in f1.items():
        targs = [t.strip() for t in nm.split("(")[0].split("<", 1)[-1].rstrip(">").split(",")]
        is_bound = len(targs) >= 3 and targs[2] == "true"   # ring <D, QW, BOUND, ...>, direct <D, QREG, BOUND, ...>
        if "topk_filter" in nm and "FETCH_SIZE" in d and not is_bound:
            us = kernel_us(os.path.join(src, f"smallb_B{B}_kernel_stats.csv"), nm.replace("ragraph::", ""))
            gb = (2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024 / 1e9
            out[f"{nm.split('<')[0].replace('ragraph::', '')} B={B} N=1000000 D=256 k=10 [{nm}]"] = {
                "FETCH_SIZE_KB": d["FETCH_SIZE"], "WRITE_SIZE_KB": d.get("WRITE_SIZE"), "hbm_side_GB": round(gb, 3),
                "kernel_us": None if us is None else round(us, 1),
                "kernel_TBps": None if not us else round(gb / us * 1e3, 2)}
# ---- round 5: the GNN forward's counter bytes (bench.py gnn_fwd.bytes_counter), the other configs, the A/B logs -------------
def maybe_cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        cp(a, b)
        return True
    return False


GNN_FORWARDS = 13   # tools/prof_gnn.py under --pmc: 3 warm-ups + 10 repetitions
for mode in ("plain", "structured"):
    path = os.path.join(src, f"gnn_{mode}_pmc.txt")
    if not os.path.exists(path):
        continue
    g = pmc(path)
    per, total = {}, 0.0
    for nm, d in g.items():
        if "FETCH_SIZE" not in d or "csr_row_normalize" in nm or "csr_sym" in nm or "sort" in nm or "scan" in nm or "ingest" in nm:
            continue
        calls = d["FETCH_SIZE_n"] / GNN_FORWARDS
        if calls < 0.9:        # (once-per-graph kernels: ingestion, row normalisation)
            continue
        kb = 2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)
        per[nm.replace("ragraph::", "")] = {"launches_per_forward": round(calls, 2), "FETCH_SIZE_KB": d["FETCH_SIZE"],
                                            "WRITE_SIZE_KB": d.get("WRITE_SIZE"), "hbm_side_MB_per_launch": round(kb * 1024 / 1e6, 1)}
        total += calls * kb * 1024
    log = open(os.path.join(src, f"gnn_{mode}.log")).read()
    m = re.search(r"(gnn_forward n=\d+ F=\d+ D=\d+ hops=\d+) nnz=(\d+).*?: ([0-9.]+) ms per forward", log)
    if m:
        key = m.group(1) + (" structured+reordered" if mode == "structured" else "")
        out[key] = {"hbm_side_bytes": int(total), "per_kernel": per, "ms_per_forward_under_rocprof": float(m.group(3)), "nnz": int(m.group(2)),
                    "how": "tools/prof_gnn.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); per launch 2 FETCH + WRITE, "
                           "summed over the launches of one forward (1 GCN layer + 3 hops)"}
    maybe_cp(f"gnn_{mode}_kernel_stats.csv", f"{P}_gnn_{mode}_kernel_stats.csv")
    maybe_cp(f"gnn_{mode}_pmc.txt", f"{P}_gnn_{mode}_pmc_traffic.txt")
maybe_cp("configs.txt", f"{P}_configs.txt")
maybe_cp("d64_ab.txt", f"{P}_d64_ab.txt")
maybe_cp("pmc_B512_sq.txt", f"{P}_B512_sq_counters.txt")
for B in (256, 512, 4096):
    maybe_cp(f"smallb_noprior_B{B}_kernel_stats.csv", f"{P}_smallb_B{B}_kernel_stats_with_bound_pass.csv")
json.dump(out, open(os.path.join(dst, f"{P}_pmc_traffic.json"), "w"), indent=2)
print(json.dumps(out, indent=1)[:3000])
