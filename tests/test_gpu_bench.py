"""bench.py's output contract on a GPU box: one JSON line (the last line of stdout) with the fields the driver reads, at N=1
and -- through the one-GPU test mode (two ranks on GPU 0, gloo) -- on the multi-rank path that `torch.distributed.run` takes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out: str):
    lines = [l for l in out.strip().splitlines() if l.strip()]
    return json.loads(lines[-1])


def test_bench_line_single_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--nv", "19"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["unit"] == "field-ops/s" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    # one thread, all cores, all cores with the improved bind, and the CPU model, side by side
    assert cb["one_thread"]["cores"] == 1 and cb["one_thread"]["value"] > 0 and cb["all_cores_improved_bind"]["value"] > 0 and cb["cpu_model"]
    assert d["scaling"] == "strong" and "config 3" in d["config"]["workload"] and d["config"]["round_loop"] == "library"
    # the line certifies its own parity: the timed proof and one after the timed region against the CPU leg's proof of the same instance
    par = d["parity"]
    assert par["ok"] is True and par["rounds"] == 19 and par["rounds_equal"] == 19 and par["rounds_equal_after_timed_region"] == 19
    assert "phases_s" in cb and cb["phases_s"]["sums"] > 0 and cb["whole_prove"]["value"] > 0
    assert "traffic_source" in rf


def _two_ranks(extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SC_BENCH_ONE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return _last_json(r.stdout)


def test_bench_line_two_ranks_one_gpu():
    """the driver's N>1 launch line on a one-GPU box: the whole sharded proof inside the library (sc_ml_prove_sharded) over its
    host transport; strong scaling = the same global instance split over the ranks"""
    d = _two_ranks(["--nv", "16"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["nv"] == 16 and d["config"]["nv_per_gpu"] == 15 and d["config"]["round_loop"].startswith("library")
    assert "selftest passed" in d["config"]["round_loop_reason"]
    # no CPU proof at N > 1: the line certifies itself through the verifier and its final oracle query over the sharded tables
    assert d["parity"]["ok"] is True and d["parity"]["verifier_accepts"] and d["parity"]["oracle_query_matches"], d["parity"]
    d = _two_ranks(["--config", "4", "--nv", "17"])
    assert d["config"]["tables"] == 3 and d["config"]["nv_per_gpu"] == 16 and "config 4" in d["config"]["workload"]
    assert d["parity"]["ok"] is True, d["parity"]
    d = _two_ranks(["--nv", "17", "--scaling", "weak"])
    assert d["parity"]["ok"] is True, d["parity"]
