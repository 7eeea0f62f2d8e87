"""Host-side pieces of bench.py that need no GPU: the N > 1 watchdog (a rank whose phase stops changing ends the job with
every rank's last heartbeat on stderr and a non-zero exit code) and the byte models of the GNN block."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_watchdog_exits_nonzero_and_names_the_phase(tmp_path):
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "hb = bench.Heartbeat(1, 2, 1.0); hb('exchange phase 1 (#3)'); time.sleep(30)") % ROOT
    env = dict(os.environ, RAGRAPH_HEARTBEAT_DIR=str(tmp_path))
    (tmp_path / "rank0.txt").write_text("1.0 all_gather (100, 4)\n")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 124
    line = [l for l in p.stderr.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert "exchange phase 1 (#3)" in rec["bench_watchdog"] and "rank 1" in rec["bench_watchdog"]
    assert rec["ranks"]["rank0"].startswith("all_gather (100, 4)") and rec["ranks"]["rank1"].startswith("exchange phase 1")


def test_heartbeat_stops_quietly(tmp_path):
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "hb = bench.Heartbeat(0, 1, 0.5); hb('timed steps'); hb.stop(); time.sleep(2); print('alive')") % ROOT
    env = dict(os.environ, RAGRAPH_HEARTBEAT_DIR=str(tmp_path))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "alive" in p.stdout


def test_gnn_byte_model_matches_survey_8d():
    """c2's hop = nnz (4 col + 4 val) + 8 (n + 1) rowptr + 4 n 256 read + 4 n 256 written = 0.214 GB;
    no-reuse replaces the read by 4 nnz 256 (1.23 GB)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_blocks as BB

    b = BB.gnn_bytes(100_000, 128, 256, 1_100_000, 3)
    assert abs(b["hop_ideal"] / 1e9 - 0.214) < 0.002
    assert abs(b["hop_no_reuse"] / 1e9 - 1.23) < 0.02
    assert b["forward_ideal"] == b["encode_ideal"] + 3 * b["hop_ideal"]
