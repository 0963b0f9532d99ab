"""bench.py's output contract on a GPU box: one JSON line (the last line of stdout) with the fields the driver reads, at N=1
and -- through the one-GPU test mode (two ranks on GPU 0) -- on the multi-rank paths: `python bench.py --gpus 2` exactly as the driver
types it (bench.py starts its own ranks: processes over torch.distributed.run, or thread ranks over the peer-to-peer communicator) and
under an external `torch.distributed.run`."""
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
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--nv", "19", "--min-gpu-seconds", "0.5"],
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
    # measurement hygiene: per-round roofline from the live events, the multiplier figures, the sampling floor
    assert d["roofline"]["event_timed_steps"] == 2 and rf["event_timed_every"] == 1  # at least 4 sampled steps, or all of them
    pr = rf["per_round"]
    assert [x["round"] for x in pr] == [1, 2, 3, 4] and all(x["ms"] > 0 and 0 < x["frac"] < 1 for x in pr)  # nv=19: four big rounds (pairs > 2^14)
    assert abs(sum(x["ms"] * x["samples"] for x in pr) / rf["launches"] - rf["avg_launch_ms"]) < 1e-6
    mu = rf["multiplier"]
    assert 0 < mu["frac_executed"] < 1 and mu["executed_products_per_s"] < mu["reference_muls_per_s"] and "modmul_fraction" not in rf
    assert d["config"]["gpu_leg"]["proofs_after_the_clock"] >= 1 and d["config"]["launcher"] == "single process"
    # the whole proof against the roof, and what it pays beside its big-round kernels
    assert abs(rf["whole_proof_frac"] - rf["whole_proof_GBps"] / rf["peak"]) < 1e-12 and 0 < rf["whole_proof_frac"] < rf["frac"]
    fc = rf["fixed_cost"]
    assert 0 < rf["fixed_cost_ms"] < d["ms_per_step"] and abs(fc["big_round_kernels_ms"] + rf["fixed_cost_ms"] - d["ms_per_step"]) < 1e-9
    assert fc["latency_bound_rounds"] == 19 - 4 and fc["finalize_inside_big_rounds_ms"] >= 0 and fc["turnaround_and_latency_bound_rounds_ms"] > 0
    # VERDICT r5 item 4 (SURVEY 8d t_end_to_end): host tables in, proof out, measured after the timed region; every such proof equals the timed one
    ee = d["config"]["end_to_end"]
    assert ee["proofs_equal_timed_region"] is True and ee["bytes"] == 10 * 32 << 19
    for kind in ("pageable", "pinned"):
        e = ee[kind]
        assert e["h2d_ms"] > 0 and e["staged_total_ms"] > 0 and e["copy_then_prove_total_ms"] > 0 and e["total_ms"] == e["staged_total_ms"]


def _two_ranks(extra, launcher=None, timeout=900):
    """launcher None: `python bench.py --gpus 2 ...` with NO launcher around it (the driver's command line); "external": under
    `python -m torch.distributed.run`; "threads" / "processes": the self-launch forced to one form"""
    env = dict(os.environ, SC_BENCH_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--min-gpu-seconds", "0.5"] + extra
    if launcher == "external":
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail + (["--launcher", launcher] if launcher else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    if launcher != "external":  # self-launched: whatever the ranks wrote to stdout (gloo's banner) went to stderr -- ONE line, the record
        assert len(lines) == 1, lines
    return json.loads(lines[-1])  # (under an external launcher the ranks' stdout is the launcher's: the record is the LAST line)


def test_bench_two_gpus_as_the_driver_types_it():
    """`python bench.py --gpus 2` with no launcher: bench.py starts its own ranks; the N > 1 line carries a CPU
    baseline and is bit-exact against the CPU proof of the same (strong-scaling) instance, on every rank"""
    d = _two_ranks(["--nv", "16"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["cpu_baseline"]["cores"] >= min(2, os.cpu_count() or 1)  # (not the OMP_NUM_THREADS=1 a launcher hands its ranks)
    # ... and its threads may run where this test's process may: the launching bench.py's main thread is bound to ONE core by its own
    # OpenMP runtime (OMP_PROC_BIND), a mask its child processes would inherit -- sixteen threads on one core
    assert d["cpu_baseline"]["cpus_allowed"] == len(os.sched_getaffinity(0)), d["cpu_baseline"]["cpus_allowed"]
    par = d["parity"]
    assert par["ok"] is True and par["rounds_equal"] == 16 and par["rounds_equal_after_timed_region"] == 16 and par["ranks_compared"] == 2, par
    c = d["config"]
    assert c["nv"] == 16 and c["nv_per_gpu"] == 15 and c["round_loop"].startswith("library") and "self-launched" in c["launcher"]
    assert c["ranks_seen"] == 2 and c["communicator"] == "host-transport" and c["exchange"]["exchange_us"] > 0 and c["exchange"]["bytes"] == 320
    # every N > 1 line carries the scaling model's prediction for itself and what it assumed
    pr = c["prediction"]
    assert c["predicted_ms_per_step"] == pr["predicted_ms_per_step"] > 0 and c["exchange_assumed_us"] == pr["exchange_assumed_us"] > 0
    assert pr["sharded_rounds"] + pr["replicated_rounds"] == 16 and pr["t1_ms"] > 0 and pr["gather_assumed_us"] > 0
    assert abs(pr["measured_over_predicted"] - d["ms_per_step"] / pr["predicted_ms_per_step"]) < 1e-9
    assert pr["predicted_ms_per_step_with_measured_exchange"] > pr["t1_ms"]
    assert "selftest passed" in c["round_loop_reason"]


def test_bench_two_thread_ranks_over_p2p():
    """the other self-launch: two thread ranks of one process over sc_comm_init_p2p (no torch.distributed, no RCCL)"""
    d = _two_ranks(["--nv", "16"], launcher="threads")
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["communicator"] == "p2p" and c["launcher"].startswith("threads") and c["round_loop"] == "library+p2p"
    assert d["cpu_baseline"]["value"] > 0 and d["parity"]["ok"] is True and d["parity"]["rounds_equal"] == 16 and d["parity"]["ranks_compared"] == 2
    assert c["exchange"]["exchange_us"] > 0


def test_bench_line_two_ranks_external_launcher():
    """under `torch.distributed.run` (the documented launch line): config 4 and weak scaling have no CPU proof of the instance -- the line
    certifies itself through the verifier and its final oracle query over the sharded tables"""
    d = _two_ranks(["--config", "4", "--nv", "17"], launcher="external")
    assert d["config"]["tables"] == 3 and d["config"]["nv_per_gpu"] == 16 and "config 4" in d["config"]["workload"]
    assert d["config"]["launcher"] == "processes (external launcher)" and d["cpu_baseline"]["value"] > 0
    from oracle import cref  # (the launcher hands its ranks OMP_NUM_THREADS=1: the CPU leg still takes the cores this process may use)
    assert d["cpu_baseline"]["cores"] == max(1, min(len(os.sched_getaffinity(0)), cref.cpu_quota_cores() or 1 << 30)), d["cpu_baseline"]["cores"]
    assert d["parity"]["ok"] is True and d["parity"]["rounds_equal"] == 17, d["parity"]  # (config 4 at nv <= 24: the CPU leg proves the instance itself)
    d = _two_ranks(["--nv", "17", "--scaling", "weak", "--no-cpu-baseline"], launcher="external")
    assert d["parity"]["ok"] is True and d["parity"]["verifier_accepts"] and d["parity"]["oracle_query_matches"], d["parity"]
    assert d["cpu_baseline"] is None


def test_bench_rccl_code_path_with_a_world_of_one():
    """what an N > 1 run over RCCL calls -- torch.distributed's "nccl" group (barrier, device all-reduce / all-gather), the library's
    RCCL communicator, its self-test and exchange timing, sc_ml_prove_sharded -- with one rank (RCCL refuses two ranks on one device)"""
    env = dict(os.environ, SC_BENCH_FORCE_SHARDED="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SC_BENCH_ONE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--nv", "19", "--steps", "2", "--warmup", "1", "--min-gpu-seconds", "0.5"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    c = d["config"]
    assert c["communicator"] == "rccl" and c["ranks_seen"] == 1 and c["round_loop"] == "library+rccl" and "forced" in c["launcher"], c
    assert c["exchange"]["exchange_us"] > 0 and "selftest passed" in c["round_loop_reason"]
    assert c["exchange"]["publication"].startswith("direct"), c["exchange"]  # the RCCL rounds run without a publish kernel (probed at sc_comm_init)
    assert d["parity"]["ok"] is True and d["parity"]["rounds_equal"] == 19, d["parity"]
    # ... and with sc_set_policy("rccl_direct", 0) through the publish kernel, to the same bits
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--nv", "19", "--steps", "2", "--warmup", "1", "--min-gpu-seconds", "0.5",
                        "--no-cpu-baseline", "--policy", "rccl_direct=0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d0 = json.loads(r.stdout.strip().splitlines()[-1])
    assert d0["config"]["exchange"]["publication"].startswith("publish kernel") and d0["parity"]["ok"] is True, d0["config"]["exchange"]


def test_bench_refuses_more_gpus_than_visible():
    env = dict(os.environ)
    env.pop("SC_BENCH_ONE_GPU", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 2 and "visible GPUs" in r.stderr and r.stdout.strip() == ""
