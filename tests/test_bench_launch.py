"""bench.py as the driver starts it: ``python3 bench.py --gpus N ...`` with NO torch.distributed environment.

For N > 1 bench.py starts its ranks itself as child processes (``python -m torch.distributed.run``), relays rank 0's
single JSON line and exits with the children's code (the 1/2/4/8 line comes from this command).
The test boxes have one GPU, so the N = 2 run uses ``--backend gloo`` (two ranks sharing the card, collectives staged
through the host); the RCCL path with the validated hipGraph default runs single-rank (``--force-distributed``).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=900):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=timeout)


def _one_json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_more_ranks_than_gpus_is_refused_before_anything_starts():
    """--gpus N over RCCL needs N devices: said plainly, exit code 2, nothing launched (runs without a GPU too)."""
    import torch

    n = torch.cuda.device_count() + 1
    if n == 1:
        n = 2
    r = _run("--gpus", str(n), "--steps", "2", timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    assert "one rank per GPU" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("geography", ["random", "clustered"])
def test_bench_starts_its_own_ranks(device, geography):
    """(both kinds of world: SURVEY 8d's uniformly random one and the one with a geography, synthetic.GEOGRAPHY)"""
    r = _run("--gpus", "2", "--backend", "gloo", "--agents", "400000", "--steps", "5", "--warmup", "2",
             "--repeats", "2", "--geography", geography)
    assert r.returncode == 0, r.stderr[-4000:]
    j = _one_json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 2 and j["scaling"] == "strong"
    assert j["value"] > 0 and abs(j["value"] - 1e3 / j["ms_per_step"]) < 1e-6 * j["value"]
    assert j["graph"] is False and "gloo" in j["graph_reason"] and j["rccl_ranks"] == 0
    assert j["repeats"]["regions"] == 2 and j["repeats"]["min"] <= j["repeats"]["median"] <= j["repeats"]["max"]
    assert j["host_us_per_step"] > 0 and j["exposed_collective_ms_per_step"] >= 0
    assert {"halo_all_to_all", "partial_all_reduce"} <= set(j["kernel_ms"])
    assert j["exchange"]["halo_agents_rank0"] > 0 and j["exchange"]["partial_sum_floats"] > 0
    assert j["config"]["geography"] == geography and j["exchange"]["venue_classes"]["household"]["venues"] > 100_000
    if geography == "clustered":         # households of neighbours: all but the ones on the one rank boundary are local
        hh = j["exchange"]["venue_classes"]["household"]
        assert hh["local"] >= hh["venues"] - 5 and j["exchange"]["halo_agents_rank0"] < 60_000
    # the two ranks together computed what one GPU computes: same seed, same world, Philox keyed by global agent id
    r1 = _run("--gpus", "1", "--agents", "400000", "--steps", "5", "--warmup", "2", "--repeats", "2",
              "--no-cpu-baseline", "--tune", "off", "--geography", geography)
    assert r1.returncode == 0, r1.stderr[-4000:]
    j1 = _one_json_line(r1.stdout)
    # both runs make warm-up + repeats x steps (+ the sequential bracketed pass of N > 1) steps: compare equal step counts
    assert j1["n_gpus"] == 1 and j1["state_checksum"]["infected"] > 0
    # ... and "after_first_region" is taken after warm-up + K steps in both.  Halo sets are bit-equal across partitions;
    # the partial-sum sets' cum differs by fp32 rounding of the two ranks' terms (<= 1e-6 relative), so a
    # decision can flip only where a probability sits within that of its Philox threshold: a handful of agents at most
    a, b = j["state_checksum"]["infected_after_first_region"], j1["state_checksum"]["infected_after_first_region"]
    assert a > 4000 and abs(a - b) <= max(5.0, 1e-3 * b), (a, b)
    assert "quarantine_social_distancing" in j1 and "high_prevalence" in j1 and "full_step" in j1


@pytest.mark.gpu
def test_rccl_step_is_captured_and_validated_by_default(device):
    """One rank under the nccl backend: the production step (asynchronous collectives) is captured after the warm-up,
    ONE replay is compared with the eager step from the same state, and the timed region replays the graph."""
    r = _run("--gpus", "1", "--force-distributed", "--agents", "400000", "--steps", "6", "--warmup", "3",
             "--repeats", "2", "--graph", "on")
    assert r.returncode == 0, r.stderr[-4000:]
    j = _one_json_line(r.stdout)
    assert j["rccl_ranks"] == 1 and j["backend"] == "nccl"
    assert j["graph"] is True, j["graph_reason"]
    # the default (--graph auto) also times replays against eager steps from one state and keeps the faster form; the
    # state is restored, so the run ends where the others end
    r0 = _run("--gpus", "1", "--force-distributed", "--agents", "400000", "--steps", "6", "--warmup", "3", "--repeats", "2")
    assert r0.returncode == 0, r0.stderr[-4000:]
    j0 = _one_json_line(r0.stdout)
    assert "steps each from one state" in j0["graph_reason"] and ("replaying" in j0["graph_reason"]) == j0["graph"]
    assert j0["state_checksum"] == j["state_checksum"]
    # the same run with the eager production step ends in the same state
    r2 = _run("--gpus", "1", "--force-distributed", "--agents", "400000", "--steps", "6", "--warmup", "3",
              "--repeats", "2", "--graph", "off")
    assert r2.returncode == 0, r2.stderr[-4000:]
    j2 = _one_json_line(r2.stdout)
    assert j2["graph"] is False and j2["state_checksum"] == j["state_checksum"]
    # (no wall-clock comparison of two separate processes on a shared box: tools/host_overhead.py measures the host time
    # per step of both forms inside one process)
    assert j["host_us_per_step"] > 0 and j2["host_us_per_step"] > 0


@pytest.mark.gpu
def test_c5_ranks_draw_and_cut_their_share_on_the_device(device):
    """BASELINE config 5 across ranks, as the driver would start it: ``python bench.py --gpus 2 --preset c5`` (here at
    20 M agents, two gloo ranks sharing the one GPU).  Every rank draws the same seeded world on the device
    (synthetic.iter_world_torch), cuts its share out there (per-venue exchange classes) and compiles it with the
    library's kernels - no rank holds the world in host memory, the set-up takes seconds, not the 7 minutes of the numpy
    route - and the two ranks together end where the single GPU ends on the same world."""
    import time

    t0 = time.time()
    r = _run("--gpus", "2", "--backend", "gloo", "--preset", "c5", "--agents", "20000000", "--generator", "torch",
             "--steps", "4", "--warmup", "2", "--repeats", "1", timeout=600)
    took = time.time() - t0
    assert r.returncode == 0, r.stderr[-4000:]
    j = _one_json_line(r.stdout)
    assert j["n_gpus"] == 2 and took < 300, took
    assert j["host_peak_rss_mb"] < 8000 and j["setup_s"]["total"] < 120, (j["host_peak_rss_mb"], j["setup_s"])
    assert "~big" in " ".join(j["exchange"]["modes"]) and j["exchange"]["venue_classes"]["household"]["partial_sum"] > 0
    r1 = _run("--gpus", "1", "--preset", "c5", "--agents", "20000000", "--generator", "torch", "--steps", "4",
              "--warmup", "2", "--repeats", "1", "--only-headline", "--tune", "off", timeout=600)
    assert r1.returncode == 0, r1.stderr[-4000:]
    j1 = _one_json_line(r1.stdout)
    a, b = j["state_checksum"]["infected_after_first_region"], j1["state_checksum"]["infected_after_first_region"]
    assert b > 200_000 and abs(a - b) <= max(5.0, 1e-4 * b), (a, b)


@pytest.mark.gpu
def test_the_reference_shaped_world_across_two_ranks(device):
    """``--preset june``: the membership structure of the reference's own graphs with its default ELEVEN networks (six on
    the one leisure set; the library holds 16 networks, so not every set that has both venue classes can be cut in two -
    the partitioner says which run whole) on two gloo ranks against one GPU: the same number of infected."""
    common = ("--preset", "june", "--agents", "400000", "--steps", "6", "--warmup", "2", "--repeats", "1", "--infected", "0.05")
    r = _run("--gpus", "2", "--backend", "gloo", *common)
    assert r.returncode == 0, r.stderr[-4000:]
    j = _one_json_line(r.stdout)
    assert j["n_gpus"] == 2 and "11 infection networks" in j["config"]["workload"]
    hh = j["exchange"]["venue_classes"]["household"]
    assert hh["local"] >= hh["venues"] - 5                         # every person in one household of neighbours
    r1 = _run("--gpus", "1", *common, "--only-headline", "--tune", "off")
    assert r1.returncode == 0, r1.stderr[-4000:]
    j1 = _one_json_line(r1.stdout)
    a, b = j["state_checksum"]["infected_after_first_region"], j1["state_checksum"]["infected_after_first_region"]
    assert b > 20_000 and abs(a - b) <= max(3.0, 1e-4 * b), (a, b)
