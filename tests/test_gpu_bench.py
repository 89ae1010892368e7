"""bench.py contract on a GPU box: one JSON line with the keys the driver reads, the roofline and cpu_baseline objects, sane values.
Runs the real script at a reduced image size (the default 700x700 run is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(*args, timeout=900, env=None):
    r = subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gpus_n_relaunches_one_rank_per_gpu():
    """`python bench.py --gpus 8` (how the driver may call it) must start 8 ranks itself: the re-launch command is torchrun with
    --nproc-per-node 8 on 127.0.0.1; a WORLD_SIZE that disagrees with --gpus is an error, not a silent 1-rank run."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--print-launch"], capture_output=True, text=True, timeout=60,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])
    assert "torch.distributed.run" in cmd and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "2"]
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=60, env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_bench_emits_contract_json():
    d = _run("--size", "128", "--steps", "2", "--warmup", "1", "--inflight", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "host_to_host_pairs_per_s", "build_id"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert d["config"]["pairs_per_gpu_per_step"] == 2 and "workload" in d["config"] and d["config"]["name"] == "pair700"
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    assert 0.5 * d["value"] < d["host_to_host_pairs_per_s"] < 1.5 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_GBs"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "k_pm_step<1, 1, 2, 2, 8>" in rf["kernel"] and "k_pm_prop<1, 1, 2, 2, 8>" in rf["kernel"] and rf["launches"] == 41 and rf["traffic"] is None       # PMC passes exist for 700x700 only
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "value_1thread"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert d["single_pair_ms"] > 0 and d["single_pair_ms_min"] <= d["single_pair_ms"] and d["stages_ms"]["total_ms"] > 0 and d["stages_ms"]["nonlocal_ms"] > 0
    assert "resident" in d["value_basis"]
    assert d["rccl_ranks"] in (1, None) and (d["rccl_ranks"] == 1 or d["rccl_error"])                # default --dist-single auto: the RCCL branch ran, or the line says why not
    import hashlib
    so = os.path.join(REPO, "neural-color-transfer_amd", "lib", "libnct.so")
    assert d["build_id_so"] == hashlib.sha256(open(so, "rb").read()).hexdigest()[:16] and len(d["build_id"]) == 16
    rc = d["roofline_color"]                     # colour-solver kernels: event-timed single launches vs their compulsory bytes (the WLS kernels exist at every size)
    assert rc["bound"] == "hbm" and rc["pixels"] == 128 * 128
    for k in ("wls_block_pre", "wls_down", "wls_up", "wls_block_post", "wls_apply", "wls_update"):
        e = rc["kernels"][k]
        assert e["samples"] == 20 and e["avg_launch_us"] > 0 and abs(e["frac"] - e["achieved"] / rc["peak"]) < 1e-9 and e["bytes_per_launch"] == e["bytes_per_pixel"] * 128 * 128
    assert rc["wls_iteration"]["us"] > 0 and rc["wls_iteration"]["survey_8d_bytes"] == 11 * 8 * 128 * 128 * 6


@pytest.mark.gpu
def test_bench_single_rank_runs_the_rccl_branch():
    """every N > 1 test uses gloo, so the RCCL branch (init_process_group("nccl") with a device id, the device-side barrier, the MAX all-reduce of a
    cuda tensor in nct.shard.timed_region) would first execute on the driver's 8-GPU node. `--dist-single on` runs that very branch with ONE rank (and the default
    `auto` does the same on every 1-GPU bench run, with a fallback): rccl_ranks must be 1 and the line otherwise unchanged."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = _run("--dist-single", "on", "--size", "96", "--steps", "2", "--warmup", "1", "--inflight", "2", "--no-cpu-baseline", "--no-roofline", env=env)
    assert d["rccl_ranks"] == 1 and d["rccl_error"] is None and d["n_gpus"] == 1
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    off = _run("--dist-single", "off", "--size", "96", "--steps", "2", "--warmup", "1", "--inflight", "2", "--no-cpu-baseline", "--no-roofline", env=env)
    assert off["rccl_ranks"] is None and off["output_checksum"] == d["output_checksum"]
    assert 0.5 * off["value"] < d["value"] < 2.0 * off["value"]                  # a one-rank barrier costs microseconds: the rate is the same within run-to-run noise


@pytest.mark.gpu
def test_bench_two_ranks_via_self_launch():
    """--gpus 2 on a 1-GPU box: the script re-launches itself under torchrun; both ranks share GPU 0 (test hook), gloo carries the
    barrier and the MAX-reduce. The line must say n_gpus 2 and count both ranks' pairs."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = _run("--gpus", "2", "--size", "96", "--steps", "2", "--warmup", "1", "--inflight", "1", "--dist-backend", "gloo", "--device-override", "0",
             "--no-cpu-baseline", "--no-roofline", env=env)
    assert d["n_gpus"] == 2 and d["config"]["pairs_per_gpu_per_step"] == 1
    assert abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_bench_two_ranks_draw_tickets_from_the_store():
    """mixed256 on 2 ranks (gloo, both on GPU 0): the ranks take pair indices from one counter in the rendezvous store — every pair once, no collective."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = _run("--gpus", "2", "--workload", "mixed256", "--batch", "6", "--steps", "1", "--warmup", "1", "--inflight", "1", "--dist-backend", "gloo",
             "--device-override", "0", "--no-cpu-baseline", "--no-roofline", env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["pairs_per_gpu_per_step"] == 3
    assert abs(d["value"] - 6 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]


def _golden_sum_700():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pair700_oracle.json")))["700"]["sum"]


@pytest.mark.gpu
@pytest.mark.parametrize("wl", ["batch64", "mixed256", "pair256l5"])
def test_bench_other_workloads(wl):
    """BASELINE configs 3 / 5 / 1 as bench workloads, at reduced batch sizes: strong-scaling lines with host-in -> host-out steps."""
    extra = ["--batch", "6"] if wl != "pair256l5" else []
    if wl == "batch64":
        extra = ["--batch", "2"]            # config 3's batch path carrying REAL 700x700 pairs (two of the 64), not a shrunken stand-in
    d = _run("--workload", wl, "--steps", "1", "--warmup", "0", "--inflight", "2", "--no-cpu-baseline", "--no-roofline", *extra, timeout=1200)
    assert d["config"]["name"] == wl and d["value"] > 0
    assert d["scaling"] == ("weak" if wl == "pair256l5" else "strong")
    if wl == "batch64":
        assert "700x700" in d["config"]["workload"] and d["config"]["pairs_per_gpu_per_step"] == 2 and d["output_checksum"] == _golden_sum_700()
