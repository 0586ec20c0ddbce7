"""The N > 1 path of bench.py (one process per GPU, reads sharded by rank, target replicated, barrier + MAX-over-ranks
timing, rank 0 prints one JSON line) exercised with world_size 2 over gloo on CPU: the ranks drive the emulated
library (tests/emu) instead of a GPU.  Every rank's shard is then checked against the oracle."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from sswutil import dna_matrix, oracle_align

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_rank_bench_over_gloo(emu_lib_path, tmp_path):
    env = dict(os.environ, SSW_BENCH_DUMP=str(tmp_path), SSW_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--reads", "5", "--ref-len", "2500", "--read-len", "70", "--cpu-sample", "3", "--lib", emu_lib_path]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # only rank 0 prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["metric"] == "GCUPS" and out["value"] >= 0
    assert out["steps"] == 1 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["roofline"]["bound"] == "valu-issue" and "frac_of_isa_ideal" in out["roofline"] and out["roofline_hbm"]["bound"] == "hbm"
    # an N > 1 line is evidence, not just a rate: every rank's shard checked against the reference, the CPU baseline beside it
    par = out["parity"]
    assert par["ranks_checked"] == 2 and [x["rank"] for x in par["per_rank"]] == [0, 1] and par["mismatching_alignments"] == 0
    assert all(x["sample"] >= 3 and x["mismatching_alignments"] == 0 for x in par["per_rank"])
    if "cpu_baseline" in out:                   # (oracle/_ref present: the build container and the GPU box)
        assert out["cpu_baseline"]["kind"] in ("reference", "port") and out["cpu_baseline"]["value"] > 0
    mat = dna_matrix(2, 2)
    seen = []
    for rank in (0, 1):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        seen.append(z["reads"].tobytes())
        for i, rd in enumerate(z["reads"]):
            d, _ = oracle_align(rd, mat, 5, z["ref"], 3, 1, 0, 0, 0, len(rd) // 2, 2, 0)
            g = z["res"][i, 0]
            assert all(int(g[k]) == d[k] for k in ("score1", "score2", "ref_end1", "read_end1", "ref_end2")), (rank, i)
    assert seen[0] != seen[1]                   # the ranks really worked on different shards


def test_eight_rank_strong_scaling_job_over_gloo(emu_lib_path):
    """bench.py --config 3 --full at the node's size: 8 ranks over gloo, the job's read blocks divided over them (block b on rank b mod 8:
    a fixed job, strong scaling), every block of every rank checked against the reference, one line from rank 0.  Toy shapes on the
    emulator; the same code path as the 1M-read job on 1..8 MI355X."""
    env = dict(os.environ, SSW_BENCH_BACKEND="gloo", SSW_EMU_DEVICES="8", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "3", "--full", "--blocks", "12",
           "--steps", "1", "--warmup", "1", "--reads", "3", "--ref-len", "2500", "--read-len", "70", "--cpu-sample", "2", "--lib", emu_lib_path]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["metric"] == "GCUPS" and out["value"] >= 0
    assert out["config"]["read_blocks"] == 12 and out["config"]["blocks_per_gpu"] == [2, 2, 2, 2, 1, 1, 1, 1] and out["config"]["reads"] == 36
    assert out["config"]["cells"] == 12 * 3 * 70 * 2500
    assert "value_with_h2d" not in out            # (the streamed pass is the one-GPU line's)
    if "parity" in out:                           # (needs oracle/_ref)
        assert out["parity"]["blocks_checked"] == 12 and out["parity"]["mismatching_alignments"] == 0
        assert sorted(x["block"] for x in out["parity"]["per_block"]) == list(range(12))


def test_full_job_on_one_rank_streams_its_blocks(emu_lib_path):
    """the one-GPU line of the same job: all blocks through one context, then once more with the reads on the host and the next block
    uploaded by a feeder thread while the current one is aligned (same records, checked inside bench.py)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--full", "--blocks", "5", "--steps", "1", "--warmup", "1", "--reads", "3",
           "--ref-len", "2500", "--read-len", "70", "--cpu-sample", "2", "--lib", emu_lib_path]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and out["config"]["blocks_per_gpu"] == [5] and out["value_with_h2d"] >= 0
    assert out["roofline"]["bound"] == "valu-issue" and out["roofline"]["launches"] == 5
    if "parity" in out:
        assert out["parity"]["blocks_checked"] == 5 and out["parity"]["mismatching_alignments"] == 0


def test_two_rank_database_search_over_gloo(emu_lib_path):
    """config-5 mode under two ranks: every rank streams its own query block against the replicated DB (no collective on the
    data path), rank 0 prints the one aggregated line"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "5", "--steps", "1", "--warmup", "0",
           "--reads", "24", "--db-targets", "9", "--db-chunk", "4", "--cpu-sample", "2", "--lib", emu_lib_path]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SSW_BENCH_BACKEND="gloo"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["alignments_per_step"] == 24 * 9
    if "parity" in out:                         # (needs oracle/_ref)
        assert out["parity"]["ranks_checked"] == 2 and out["parity"]["mismatching_alignments"] == 0
    assert out["config"]["baseline_config"] == 5 and out["roofline"]["bound"] == "valu-issue" and out["roofline_hbm"]["bound"] == "hbm"


def test_pool_mode_of_bench_on_two_pretend_devices(emu_lib_path):
    """bench.py --pool 2: the in-library per-GPU work queues drive the batch (single process, reads on the host)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--pool", "2", "--steps", "1", "--warmup", "0", "--reads", "12", "--ref-len", "2500",
           "--read-len", "70", "--cpu-sample", "4", "--lib", emu_lib_path]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SSW_EMU_DEVICES="2"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["config"]["pool_workers"] == 2 and [s["device"] for s in out["pool_stats"]] == [0, 1]
    assert sum(s["queries"] for s in out["pool_stats"]) == 12 and out["parity"]["mismatching_alignments"] == 0


def test_gpus_flag_without_a_launcher_starts_the_ranks_itself(emu_lib_path):
    """`python bench.py --gpus 2` as a plain command (no torch.distributed.run around it): bench.py launches the two ranks itself and
    rank 0's single line comes through with n_gpus 2"""
    env = dict(os.environ, SSW_BENCH_BACKEND="gloo", SSW_EMU_DEVICES="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--reads", "4", "--ref-len", "2000",
           "--read-len", "60", "--cpu-sample", "0", "--lib", emu_lib_path]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["reads_per_gpu"] == 4 and "also" not in out


def test_also_block_carries_the_other_configs(emu_lib_path):
    """the metric's line with configs 3, 4 and 5 attached (`also`): here on the emulator with toy shapes, on the GPU at the stated sizes"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--reads", "3", "--ref-len", "2000", "--read-len", "60",
           "--cpu-sample", "0", "--also", "2u8,3,4,5,6", "--lib", emu_lib_path]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and set(out["also"]) >= {"config2u8", "config3", "config4", "config5", "config6"}
    assert out["also"]["config2u8"]["mix"]["word_rules"] == 0          # 1/-3/5/2: every read decided under the 8-bit rules
    for c in ("config2u8", "config3", "config4", "config5", "config6"):
        assert "error" not in out["also"][c], out["also"][c]
        assert out["also"][c]["ms_per_step"] > 0 and out["also"][c]["fill_kernel"]
