"""GPU: the multi-GPU step on ONE device.  R "virtual ranks" each compile their part of the world
(halo agents, partial-sum sets) and run the real kernels; the two collectives are stood in by
device-side copies / sums between the ranks' buffers.  The result must equal the single-rank run:
probabilities to 1e-6, decisions and infection counts exactly (Philox is keyed by global agent id)."""
import numpy as np
import pytest
import torch

import bench as B
from grad_june_amd.benchrun import SingleGpuHotPath
from grad_june_amd.distributed import DistributedHotPath, partition_bounds
from grad_june_amd.synthetic import make_world

pytestmark = pytest.mark.gpu


class VirtualRank(DistributedHotPath):
    """DistributedHotPath whose collectives are performed by the test across in-process ranks."""

    def step_until_exchange(self):
        self.p = self.params()
        self.engine.step_phase(self.bufs, self.p, self.io, 0)

    def step_after_halo(self):
        self.engine.step_phase(self.bufs, self.p, self.io, 1)
        self.engine.step_phase(self.bufs, self.p, self.io, 5)

    def step_after_reduce(self):
        self.engine.step_phase(self.bufs, self.p, self.io, 6)
        self.engine.step_phase(self.bufs, self.p, self.io, 3)
        self.t += 1


def reference_world(device):
    """The reference's 769-agent world through the drop-in API (Runner.get_data + seeding), as the neutral
    description the partitioner takes (distributed.world_from_data); 11 networks incl. care_visit."""
    import grad_june_amd as G
    from grad_june_amd.defaults import default_parameters
    from grad_june_amd.distributed import world_from_data

    torch.manual_seed(12)
    params = default_parameters(str(device))
    params["infection_seed"]["log_fraction_initial_cases"] = -1.0
    runner = G.Runner.from_parameters(params)
    runner.set_initial_cases()
    world = world_from_data(runner.data, model=runner.model)
    assert world["n_agents"] == 769 and len(world["networks"]) == 11 and world["state"]["is_infected"].sum() > 20
    return world


@pytest.mark.parametrize("R,source", [(2, "c3"), (4, "c3"), (3, "reference-769"), (3, "c3-quarantine"), (3, "c5-split")])
def test_virtual_ranks_match_single_rank(device, R, source):
    world = (reference_world(device) if source == "reference-769"
             else make_world("c5" if source == "c5-split" else "c3", n_agents=40_000, seed=3, infected_fraction=0.05))
    specs, betas = B.network_specs(world), B.betas_of(world)
    betas = {k: 3.0 * v for k, v in betas.items()} if source == "reference-769" else betas
    kw, modes = {}, None
    if source == "c3-quarantine":
        # an active quarantine policy, with a MASKED set (care homes) forced into halo mode: its halo agents'
        # q * transmission must travel too
        from grad_june_amd.distributed import choose_modes

        kw = {"quarantine_threshold": 4.0}
        modes = dict(choose_modes(world, R), care_home="halo")
    if source == "reference-769":
        # 769 agents: too little halo for the per-venue rule to cut anything in two (every set would run in halo mode);
        # the per-set rule of rounds 1-3 keeps both exchanges in this case
        from grad_june_amd.distributed import choose_modes

        modes = choose_modes(world, R)
    single = SingleGpuHotPath(world, specs, betas, device, seed=7, layout="tiled", **kw)
    ranks = [VirtualRank(world, specs, betas, device, r, R, seed=7, collectives=False, modes=modes, **kw)
             for r in range(R)]
    assert all(rk.exchange_q == (source == "c3-quarantine") for rk in ranks)
    assert {m for rk in ranks for m in rk.rw.modes.values()} == {"halo", "partial"}
    if source == "c5-split":     # power-law venues: every set cut into a halo half and a partial-sum half, networks twinned
        halves = [n for n in ranks[0].rw.edge_sets if n.endswith("~big")]
        assert len(halves) >= 4 and len(ranks[0].rw.edge_sets) == len(world["edge_sets"]) + len(halves)
        assert len(ranks[0].networks) > len(world["networks"]) and all(n + "~big" in ranks[0].networks for n in ("household", "school"))
    b = partition_bounds(world["n_agents"], R)
    for step in range(3):
        single.step()
        for rk in ranks:
            rk.step_until_exchange()
        # halo all-to-all stand-in: every rank's halo slots <- the owners' fresh transmissions
        for key in ("transmission",) + (("q_transmission",) if ranks[0].exchange_q else ()):
            glob = torch.cat([rk.state[key][: rk.rw.n_local] for rk in ranks])
            for rk in ranks:
                idx = torch.from_numpy(rk.rw.halo_global).to(device)
                rk.state[key][rk.rw.n_local_pad:rk.rw.n_local_pad + rk.rw.n_halo] = glob[idx]
        for rk in ranks:
            rk.step_after_halo()
        # all-reduce stand-in over the flat partial-sum buffers
        total = torch.stack([rk.flat_cum for rk in ranks]).sum(0)
        for rk in ranks:
            rk.flat_cum.copy_(total)
        for rk in ranks:
            rk.step_after_reduce()
        torch.cuda.synchronize()
        for k in ("is_infected", "susceptibility", "infection_time"):
            got = torch.cat([rk.state[k] for rk in ranks]).cpu().numpy()
            ref = single.state[k].cpu().numpy()
            assert np.array_equal(got, ref), f"step {step}: {k}"
        got_new = torch.cat([rk.new_infected for rk in ranks]).cpu().numpy()
        assert np.array_equal(got_new > 0.5, single.new_infected.cpu().numpy() > 0.5)
    assert single.state["is_infected"].sum().item() > 0.05 * world["n_agents"]


def _two_rank_worker(rank, R, port, out, min_group_floats):
    """One real process per rank (both on cuda:0), torch.distributed with gloo, collectives staged
    through the host - the multi-process flow bench.py --gpus N takes, minus RCCL."""
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=R)
    try:
        device = torch.device("cuda:0")
        world = make_world("c3", n_agents=30_001, seed=4, infected_fraction=0.05)       # partitions of unequal size
        specs, betas = B.network_specs(world), B.betas_of(world)
        rk = DistributedHotPath(world, specs, betas, device, rank, R, seed=9, min_group_floats=min_group_floats)
        assert rk.halo.host_staged and rk.halo.active
        assert len(rk.reduce_groups) == (2 if min_group_floats == 1 else 1)
        for _ in range(3):
            rk.step()
        torch.cuda.synchronize()
        mine = rk.state["is_infected"].cpu()
        assert mine.numel() == int(rk.rw.bounds[rank + 1] - rk.rw.bounds[rank])
        parts = [None] * R
        dist.all_gather_object(parts, mine)          # every rank enters the collective; partitions may differ in size
        if rank == 0:
            single = SingleGpuHotPath(world, specs, betas, device, seed=9, layout="tiled")
            for _ in range(3):
                single.step()
            torch.cuda.synchronize()
            ref = single.state["is_infected"].cpu()
            assert torch.equal(torch.cat(parts), ref)
            out[0] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("min_group_floats", [1 << 16, 1], ids=["one-all-reduce", "pipelined-all-reduces"])
def test_two_processes_gloo_match_single_rank(device, min_group_floats):
    import os

    import torch.multiprocessing as mp

    R = 2
    out = mp.get_context("spawn").Array("i", [0])
    mp.spawn(_two_rank_worker, args=(R, 29600 + os.getpid() % 300, out, min_group_floats), nprocs=R, join=True)
    assert out[0] == 1


def test_partitioned_on_one_gpu_matches_unpartitioned(device):
    """PartitionedHotPath (the >16 M-agent single-GPU mode) == the unpartitioned run, bit for bit."""
    from grad_june_amd.distributed import PartitionedHotPath

    world = make_world("c5", n_agents=60_000, seed=8, infected_fraction=0.05)
    specs, betas = B.network_specs(world), B.betas_of(world)
    single = SingleGpuHotPath(world, specs, betas, device, seed=5, layout="tiled")
    parted = PartitionedHotPath(world, specs, betas, device, parts=3, seed=5)
    for _ in range(3):
        single.step()
        parted.step()
    torch.cuda.synchronize()
    for k, v in parted.state.items():
        assert torch.equal(v, single.state[k]), k
    assert single.state["is_infected"].sum() > 0.05 * 60_000


def test_partial_sum_reduction_order_stays_within_fp32_rounding(device):
    """What N > 1 over RCCL can and cannot promise.  Halo sets are bitwise independent of the partition (every rank
    holds a venue's complete attendee list and sums it in fixed point).  Partial-sum sets travel as fp32 `cum` and the
    all-reduce adds the ranks' terms in an order of its own: summed in the reversed rank order the per-venue sums agree
    to a few ulp (<= 1e-6 relative, SURVEY 8e's tolerance), the probabilities to 1e-6 absolute, and a decision changes
    only where a probability sits within that of its Philox threshold."""
    from grad_june_amd.distributed import PartitionedHotPath

    world = make_world("c3", n_agents=200_000, seed=9, infected_fraction=0.1)
    specs, betas = B.network_specs(world), B.betas_of(world)
    runs = []
    for order in (None, list(range(7, -1, -1))):
        r = PartitionedHotPath(world, specs, betas, device, parts=8, seed=5)
        r.reduce_order = order
        probs = [torch.empty(rk.rw.n_local, device=device) for rk in r.ranks]
        for rk, p in zip(r.ranks, probs):
            rk.io = rk.engine.io(new_infected=rk.new_infected, not_infected_probs=p)
        r.step()
        torch.cuda.synchronize()
        runs.append((r, torch.cat(probs), torch.cat([rk.new_infected for rk in r.ranks])))
    (a, pa, na), (b, pb, nb) = runs
    assert any(m == "partial" for m in a.ranks[0].rw.modes.values())
    worst = 0.0
    for name, mode in a.ranks[0].rw.modes.items():
        ca, cb = a.ranks[0].engine.plan.cum_of(name), b.ranks[0].engine.plan.cum_of(name)
        if mode != "partial":
            continue
        rel = ((ca - cb).abs() / ca.abs().clamp_min(1e-30)).max().item()
        worst = max(worst, rel)
        assert rel <= 1e-6, (name, rel)
    assert worst > 0.0                                  # the order does reach the last bits: "bit for bit" would be false
    assert float((pa - pb).abs().max()) <= 1e-6
    assert int((na != nb).sum()) <= 2 and float(na.sum()) > 1000


def test_pack_unpack_kernels(device):
    """gj_pack_f32 / gj_unpack_f32 (halo send / receive buffers)."""
    from grad_june_amd import _native as N

    lib = N.load()
    torch.manual_seed(0)
    src = torch.rand(100_000, device=device)
    idx = torch.randperm(100_000, device=device)[:37_001].to(torch.int32)
    out = torch.empty(idx.numel(), device=device)
    N.check(lib.gj_pack_f32(idx.numel(), N.ptr(idx), N.ptr(src), N.ptr(out), N.current_stream()), "gj_pack_f32")
    assert torch.equal(out, src[idx.long()])
    dst = torch.zeros(100_000, device=device)
    N.check(lib.gj_unpack_f32(idx.numel(), N.ptr(idx), N.ptr(out), N.ptr(dst), N.current_stream()), "gj_unpack_f32")
    ref = torch.zeros(100_000, device=device)
    ref[idx.long()] = out
    assert torch.equal(dst, ref)
    assert lib.gj_pack_f32(0, None, None, None, None) == 0 and lib.gj_pack_f32(5, None, None, None, None) == -1


def _api_worker(rank, R, port, out):
    """The API-level run of one rank: DistributedRunner on the reference's 769-agent world, default.yaml."""
    import itertools
    import os

    import torch.distributed as dist

    import grad_june_amd as G
    from grad_june_amd import infection
    from grad_june_amd.defaults import default_parameters
    from grad_june_amd.distributed_api import DistributedRunner

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=R)
    try:
        def params():
            p = default_parameters("cuda:0")
            p["timer"]["total_days"] = 10
            p["infection_seed"]["log_fraction_initial_cases"] = -1.5
            for n in p["networks"]:
                p["networks"][n]["log_beta"] += 0.5
            p["policies"]["quarantine"] = {
                "quarantine": {1: {"start_date": "2022-02-03", "end_date": "2022-02-20", "stage_threshold": 4}}}
            return p

        torch.manual_seed(21)
        runner = DistributedRunner.from_parameters(params())
        assert runner.n_agents_total == 769 and 0 < runner.n_agents < 769
        with torch.no_grad():
            results, local_inf = runner()
        a0, a1 = runner.model.agent_range
        assert local_inf.shape[0] == a1 - a0
        gathered = [None] * R
        dist.all_gather_object(gathered, local_inf.cpu())
        if rank == 0:
            torch.manual_seed(21)
            infection._philox_step = itertools.count(1 << 40)        # the seeding stream of a fresh process
            single = G.Runner.from_parameters(params())
            with torch.no_grad():
                ref, ref_inf = single()
            for key in ("cases_per_timestep", "deaths_per_timestep", "cases_by_age_18", "cases_by_age_65", "cases_by_age_100"):
                assert torch.equal(results[key].cpu(), ref[key].cpu()), key
            assert torch.equal(torch.cat(gathered), ref_inf.cpu())
            assert ref["cases_per_timestep"][-1] > ref["cases_per_timestep"][0] > 0
            out[0] = 1
    finally:
        dist.destroy_process_group()


def test_api_runner_two_processes_match_single_gpu(device):
    """DistributedRunner (model / runner API across ranks; two processes sharing the GPU, gloo) reproduces the
    single-GPU Runner's series and final per-agent state for the same seed: timer, policies incl. a quarantine
    window, 11 networks, symptoms, seeding - everything the reference's run() does."""
    import os

    import torch.multiprocessing as mp

    R = 2
    out = mp.get_context("spawn").Array("i", [0])
    mp.spawn(_api_worker, args=(R, 29900 + os.getpid() % 90, out), nprocs=R, join=True)
    assert out[0] == 1


def _golden_chain_worker(rank, R, port, out):
    """One rank of the reference's recorded 15-step trajectory (june769.npz), everything carried forward."""
    import json
    import os

    import torch.distributed as dist

    import gj_testlib as L
    import grad_june_amd as G
    from grad_june_amd.distributed_api import DistributedGradJune
    from test_host_logic import _cpu

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=R)
    try:
        device = torch.device("cuda:0")
        npz = L.load_npz("june769.npz")
        world = L.world_from(npz)
        params = _cpu(json.loads(str(npz["params_json"])))
        params["system"]["device"] = str(device)
        model = DistributedGradJune.from_parameters(params)
        timer = G.Timer.from_parameters(params)
        rec0 = L.step_record(npz, "step0/")
        d = G.HeteroData()
        ag = d["agent"]
        A = world["n_agents"]
        ag.id = torch.arange(A)
        ag.age, ag.sex = world["age"], world["sex"]
        for s, es in world["edge_sets"].items():
            d[s].id = torch.arange(len(es["people"]))
            d[s].people = es["people"]
            d["agent", "attends_" + s, s].edge_index = torch.vstack((es["agent"], es["venue"]))
        d = d.to(device)
        pre = L.pre_state(rec0)
        ag.infection_parameters = {k: pre[k].to(device) for k in ("max_infectiousness", "shape", "rate", "shift")}
        for k in ("is_infected", "susceptibility", "infection_time"):
            ag[k] = pre[k].to(device)
        ag.transmission = torch.zeros(A, device=device)
        ag.symptoms = {k[8:]: torch.from_numpy(v).to(device) for k, v in rec0.items() if k.startswith("sym_pre/")}
        ag.symptoms["current_stage"] = torch.from_numpy(rec0["pre/current_stage"]).float().to(device)
        local = model.partition(d)
        a0, a1 = model.agent_range
        with torch.no_grad():
            for i in range(int(npz["n_steps"])):
                rec = L.step_record(npz, f"step{i}/")
                next(timer)
                new, _ = model.hot_path(local, timer, exp_noise=torch.from_numpy(rec["exp_noise"]))
                for k in ("susceptibility", "is_infected", "infection_time"):
                    assert np.array_equal(local["agent"][k].cpu().numpy(), rec["post/" + k][a0:a1]), (i, k)
                model.symptoms_updater(local, timer, new, progresses=torch.from_numpy(rec["sym/progresses"][a0:a1]),
                                       dwell=torch.from_numpy(rec["sym/dwell"][a0:a1]))
                for k in ("current_stage", "next_stage", "time_to_next_stage"):
                    assert np.array_equal(local["agent"].symptoms[k].cpu().numpy(), rec["sym_post/" + k][a0:a1]), (i, k)
        total = torch.tensor([float(local["agent"].is_infected.sum())])
        dist.all_reduce(total)
        assert float(total) == float(npz["cases_per_timestep"][-1])
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_the_reference_trajectory(device):
    """The multi-rank path against the REFERENCE itself: the recorded 15-step run of the 769-agent world (11
    networks, policies, symptoms; reference noise injected) is reproduced exactly by two ranks - every state
    array of every agent after every step, and the final case count."""
    import os

    import torch.multiprocessing as mp

    R = 2
    out = mp.get_context("spawn").Array("i", [0] * R)
    mp.spawn(_golden_chain_worker, args=(R, 29700 + os.getpid() % 90, out), nprocs=R, join=True)
    assert list(out) == [1] * R


def _grad_worker(rank, R, port, out, modes):
    """DistributedRunner in differentiable mode on one rank: every log_beta an nn.Parameter, a loss on the
    rank-summed series back-propagated on every rank."""
    import itertools
    import os

    import torch.distributed as dist

    import grad_june_amd as G
    from grad_june_amd import distributed as D
    from grad_june_amd import infection
    from grad_june_amd.defaults import default_parameters
    from grad_june_amd.distributed_api import DistributedRunner

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=R)
    try:
        def params():
            p = default_parameters("cuda:0")
            p["timer"]["total_days"] = 6
            p["infection_seed"]["log_fraction_initial_cases"] = -1.3
            for n in p["networks"]:
                p["networks"][n]["log_beta"] += 0.6
            p["policies"]["quarantine"] = {
                "quarantine": {1: {"start_date": "2022-02-03", "end_date": "2022-02-20", "stage_threshold": 4}}}
            return p

        def run(runner):
            nets = runner.model.infection_networks.networks
            for n in nets.values():
                n.log_beta = torch.nn.Parameter(n.log_beta.detach().clone())
            results, _ = runner()
            w = torch.linspace(0.5, 1.5, results["cases_per_timestep"].numel(), device=results["cases_per_timestep"].device)
            loss = (results["cases_per_timestep"] * w).sum() + 3.0 * results["deaths_per_timestep"].sum() \
                + 0.25 * results["cases_by_age_65"].sum()
            loss.backward()
            return results, {k: (None if n.log_beta.grad is None else float(n.log_beta.grad)) for k, n in nets.items()}

        if modes:        # exercise both exchange modes on this small world (default: by mean venue size)
            real = D.mode_of
            def forced(n_edges, n_venues, world_size, people=None):
                m = modes[0] if n_venues > 50 else modes[1]
                if m == "split" and not ((np.asarray(people) > 8).any() and (np.asarray(people) <= 8).any()):
                    m = "partial"
                return m

            D.mode_of = forced
            D.EXCHANGE_RULE = "set"        # whole sets in the forced mode (default "venue": classes venue by venue)
        torch.manual_seed(33)
        res, grads = run(DistributedRunner.from_parameters(params()))
        if modes:
            D.mode_of = real
            D.EXCHANGE_RULE = "venue"
        gathered = [None] * R
        dist.all_gather_object(gathered, grads)
        assert gathered[0] == gathered[1], "every rank holds the whole gradient"
        if rank == 0:
            torch.manual_seed(33)
            infection._philox_step = itertools.count(1 << 40)
            ref_res, ref = run(G.Runner.from_parameters(params()))
            assert torch.equal(res["cases_per_timestep"].detach().cpu(), ref_res["cases_per_timestep"].detach().cpu())
            assert torch.equal(res["deaths_per_timestep"].detach().cpu(), ref_res["deaths_per_timestep"].detach().cpu())
            nonzero = 0
            for k, g in ref.items():
                if g is None:
                    assert grads[k] is None, k
                    continue
                # one sum order differs (the ranks' dot products are added in fp64): relative 1e-5
                assert grads[k] == pytest.approx(g, rel=2e-5, abs=1e-7), (k, grads[k], g)
                nonzero += g != 0.0
            assert nonzero >= 5
            out[0] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("modes", [None, ("halo", "halo"), ("partial", "partial"), ("split", "split")],
                         ids=["default", "all-halo", "all-partial", "split"])
def test_two_ranks_gradients_match_single_gpu(device, modes):
    """example_scripts/run_model.py:9-11 on two ranks: the gradients of a loss on the case / death series w.r.t.
    every network's log_beta equal the single-GPU run's (same seed; 11 networks, quarantine window, symptoms) and
    are the same on every rank - whatever the exchange modes, "split" sets with their twin networks included."""
    import os

    import torch.multiprocessing as mp

    R = 2
    out = mp.get_context("spawn").Array("i", [0])
    mp.spawn(_grad_worker, args=(R, 29400 + os.getpid() % 90, out, modes), nprocs=R, join=True)
    assert out[0] == 1
