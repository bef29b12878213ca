"""GPU: the RCCL branch of the multi-rank step, executed for real.

Every other multi-rank test stands the collectives in (device copies) or stages them through the host (gloo); the
production step of distributed.DistributedHotPath - asynchronous all_to_all_single / all_reduce on RCCL's stream,
overlapped with the phases that do not depend on them, work.wait() before the ones that do - runs here under the
`nccl` backend (= RCCL on ROCm) with world_size 1, in a spawned child whose FIRST GPU call is the process-group init:

  * eager production step == SingleGpuHotPath, bit for bit, over several steps (with and without a quarantine policy,
    one and two all-reduce groups);
  * the same step captured ONCE in a hipGraph - kernels and collectives - and replayed per timestep with the device
    clock (engine.StepClock) == the eager run, bit for bit: one host call per step.
(N > 1 over xGMI needs a multi-GPU node, which the test boxes are not: the driver's scaling run is its first execution.)"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, port, out, quarantine, min_group_floats):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # before any other GPU call
    try:
        import bench as B
        from grad_june_amd.benchrun import SingleGpuHotPath
        from grad_june_amd.distributed import DistributedHotPath, choose_modes
        from grad_june_amd.synthetic import make_world

        assert dist.get_backend() == "nccl"
        world = make_world("c3", n_agents=200_000, seed=6, infected_fraction=0.05)
        specs, betas = B.network_specs(world), B.betas_of(world)
        # one rank has neither halo agents nor remote partial sums: force the exchange modes a multi-rank run takes
        modes = {k: ("halo" if k == "household" else "partial") for k in choose_modes(world, 1)}
        kw = {"quarantine_threshold": 4.0} if quarantine else {}
        keys = ("is_infected", "susceptibility", "infection_time")

        def fresh():
            return DistributedHotPath(world, specs, betas, dev, 0, 1, seed=13, modes=modes, production_at_one_rank=True,
                                      min_group_floats=min_group_floats, **kw)

        single = SingleGpuHotPath(world, specs, betas, dev, seed=13, layout="tiled", **kw)
        eager = fresh()
        assert eager.halo is not None and eager.halo.active and not eager.halo.host_staged
        assert len(eager.reduce_groups) == (2 if min_group_floats == 1 else 1) and eager.flat_cum.numel() > 1000
        trail = []
        for _ in range(4):
            single.step()
            eager.step()                          # production form: async collectives, overlapped phases
            torch.cuda.synchronize()
            for k in keys:
                assert torch.equal(eager.state[k], single.state[k]), k
            assert torch.equal(eager.new_infected, single.new_infected)
            trail.append({k: eager.state[k].clone() for k in keys})
        assert single.state["is_infected"].sum().item() > 1.2 * world["state"]["is_infected"].sum()

        replayed = fresh()
        replayed.capture()
        assert replayed.clock.read()[0] == 0.0            # one step before the first replay's now = 1.0
        for i in range(4):
            replayed.step()                       # graph.replay(): clock advance + kernels + collectives
            torch.cuda.synchronize()
            for k in keys:
                assert torch.equal(replayed.state[k], trail[i][k]), (i, k)
        now, step = replayed.clock.read()
        assert now == 4.0 and step == 3
        out[0] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("quarantine,min_group_floats", [(False, 1 << 16), (False, 1), (True, 1 << 16)],
                         ids=["one-all-reduce", "two-all-reduces", "quarantine"])
def test_rccl_production_step_and_graph_replay(device, quarantine, min_group_floats):
    import torch.multiprocessing as mp

    out = mp.get_context("spawn").Array("i", [0])
    mp.spawn(_worker, args=(29400 + os.getpid() % 300, out, quarantine, min_group_floats), nprocs=1, join=True)
    assert out[0] == 1


def test_single_gpu_step_replayed_with_the_device_clock(device):
    """SingleGpuHotPath.capture(): the four launches of gj_step behind a clock-advance node; replays walk through the
    timesteps (now, Philox stream) exactly like eager steps do."""
    import bench as B
    from grad_june_amd.benchrun import SingleGpuHotPath
    from grad_june_amd.synthetic import make_world

    world = make_world("c3", n_agents=150_000, seed=2, infected_fraction=0.05)
    specs, betas = B.network_specs(world), B.betas_of(world)
    eager = SingleGpuHotPath(world, specs, betas, device, seed=4, layout="tiled")
    graph = SingleGpuHotPath(world, specs, betas, device, seed=4, layout="tiled")
    graph.capture()
    for i in range(5):
        eager.step()
        graph.step()
        torch.cuda.synchronize()
        for k in ("is_infected", "susceptibility", "infection_time"):
            assert torch.equal(graph.state[k], eager.state[k]), (i, k)
        assert torch.equal(graph.new_infected, eager.new_infected) and torch.equal(graph.probs, eager.probs)
    assert graph.clock.read() == (5.0, 4)
    # an eager, event-bracketed step in between (bench.py's per-launch timing) keeps the device clock in step
    eager.step()
    graph.step(timed=True)
    eager.step()
    graph.step()
    torch.cuda.synchronize()
    assert graph.clock.read() == (7.0, 6)
    for k in ("is_infected", "susceptibility", "infection_time"):
        assert torch.equal(graph.state[k], eager.state[k]), k
    assert eager.state["is_infected"].sum().item() > 1.2 * world["state"]["is_infected"].sum()
