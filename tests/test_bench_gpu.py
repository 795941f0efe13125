"""bench.py as the driver runs it: the one-GPU line and -- rehearsed with two ranks on one GPU
over gloo (ARVX_BENCH_ONE_GPU=1) -- the N > 1 line with every collective, several jobs in flight
and the hand-off on its side streams.  Small grids: what is checked is the contract of the JSON
line and that every merge holds the right planes, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def last_json_line(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_one_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "128", "--steps", "9",
                        "--warmup", "2", "--extra-grid", "0", "--no-workloads", "--no-ablation",
                        "--rounds", "3"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json_line(r.stdout)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 9 and d["warmup"] == 2 and d["higher_is_better"] is True
    # the headline is one step after the other; several jobs in flight is a figure of its own
    assert d["config"]["jobs_in_flight"] == 1 and d["slots_agree"] is True
    assert d["rounds"]["n"] == 3 and len(d["rounds"]["ms_per_step"]) == 3
    assert d["rounds"]["min"] <= d["ms_per_step"] <= d["rounds"]["max"]
    t = d["throughput_jobs_in_flight"]
    assert t["jobs"] == 4 and t["slots_agree"] is True and t["value"] > 0
    assert d["e2e"]["runs"][0]["ms_total"] > 0 and d["e2e"]["runs"][0]["bytes_h2d"] > 0
    assert d["e2e"]["runs"][0]["packets_equal_planes"] is True
    assert d["e2e"]["runs"][0]["bytes_d2h"] < d["e2e"]["runs"][0]["planes_form"]["bytes_d2h"]
    assert d["parity_vs_oracle"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["latency_ms_per_step"] > 0
    assert d["roofline"]["kernel_ms"] > 0 and d["cpu_baseline"]["value"] > 0
    assert d["vs_baseline"] is None and d["scaling"] == "weak"


@pytest.mark.parametrize("jobs", [1, 4])
def test_two_rank_rehearsal_every_collective(jobs):
    # jobs = 1: the driver's line (one job after the other, the hand-off on side streams beside the
    # next job); jobs = 4: every slot hands its job over in its own stream
    env = dict(os.environ, ARVX_BENCH_ONE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "9",
                        "--warmup", "2", "--no-mgpu", "--rounds", "2", "--jobs", str(jobs)],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json_line(r.stdout)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["config"]["jobs_in_flight"] == jobs and d["slots_agree"] is True
    assert d["rounds"]["n"] == 2
    assert d["config"]["merged_plane_holds_rank0_planes"] is True
    assert set(d["collectives"]) == {"compressed", "allreduce", "allgather", "packets"}
    for name, c in d["collectives"].items():
        assert "error" not in c, (name, c)
        assert c["merge_ok"] is True, name
        assert c["ms_per_step"] > 0 and c["exchange_bytes_per_rank"] > 0
    assert d["collective_backend"]["ranks"] == 2


def test_bench_starts_its_own_ranks():
    """Plain `python bench.py --gpus 2`, the way the driver starts `--gpus 1` (no launcher, no
    WORLD_SIZE): the process starts its two ranks itself and relays rank 0's line -- it must not
    measure one GPU and call it two (missing 3)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ARVX_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128",
                        "--steps", "6", "--warmup", "2", "--no-mgpu", "--rounds", "1", "--no-cpu",
                        "--no-workloads"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["collective_backend"]["ranks"] == 2
    assert d["config"]["merged_plane_holds_rank0_planes"] is True


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus N under a launcher that started another number of ranks fails loudly -- before any
    GPU work (needs no GPU, but lives with the other bench tests)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
