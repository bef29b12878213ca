"""GPU: BASELINE.json's full-size synthetic worlds, checked through size-independent properties
(the oracle would need minutes per step at these sizes):

  * tiled and CSR layouts agree (probabilities <= 2e-6, decisions equal away from Gumbel ties)
  * pass 1 is linear: doubling every transmission doubles every venue sum bit for bit
  * mass conservation: sum_v (sum of venue v) == sum_a degree(a) * transmission[a]   (fp64 check sums)
  * a random sample of venues / agents against fp64 sums taken straight from the COO edge list
  * run-to-run determinism: bitwise identical outputs

C2 (1 M agents, 15 M edges) and C3 (10 M agents, 120 M network-edges: the world bench.py times, same seed) run at
full size; C4's partitioning runs as 8 agent partitions of that C3 world on the one GPU; C5 (power-law venues up to
50 000 attendees) at 1 M agents here and at 20 M and 100 M agents (BASELINE configs[4]'s full size, one GPU) in
test_c5_20m_against_fp64_device_sums.
"""
import os

import numpy as np
import pytest
import torch

import bench as B
from grad_june_amd.benchrun import SingleGpuHotPath
from grad_june_amd.synthetic import edge_set_of, make_world

pytestmark = pytest.mark.gpu

# c5 = power-law venue degrees (Zipf alpha 2, venues up to 50 000 attendees)
# june = the membership structure of the reference's own graphs, its default eleven networks (six on the leisure set,
# care_visit with its asymmetric age > 75 weight), every person in exactly one household (the run form takes the whole set)
CASES = [("c2", None), ("c3", None), ("c5", 1_000_000), ("june", 2_000_000)]

_WORLDS = {}


def cached_world(preset, agents, infected=0.03):
    """The numpy-seeded worlds take ~30 s per 10 M agents to draw: one copy per (preset, size) for the whole module."""
    key = (preset, agents, infected)
    if key not in _WORLDS:
        _WORLDS[key] = make_world(preset, n_agents=agents, seed=1234, infected_fraction=infected)
    w = _WORLDS[key]
    return dict(w, state={k: v.copy() for k, v in w["state"].items()})


def fp64_device_reference(world, r, betas, device):
    """Both passes restated with torch fp64 ops on the device, straight from the COO edge lists: per network the
    venue sums cum_n[v] (base.py:78-79) and per agent ts = susceptibility * sum_n w_n[a] * sum_e cum_n[venue(e)]
    (base.py:80-83, leisure_network.py:74-85) from the transmissions the kernels wrote."""
    A = world["n_agents"]
    x = r.state["transmission"][:A].double()
    susc = r.state["susceptibility"].double()
    cls = torch.from_numpy((world["sex"] * 100 + world["age"]).astype(np.int64)).to(device)
    specs = {s.name: s for s in B.network_specs(world)}
    tdev = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(device)
    cum, acc = {}, torch.zeros(A, dtype=torch.float64, device=device)
    for name in world["networks"]:
        es = world["edge_sets"][edge_set_of(name)]
        agent, venue = tdev(es["agent"]), tdev(es["venue"])
        people = tdev(es["people"]).double()
        pc = torch.clamp(1.0 / (people - 1.0), max=1.0).clamp(min=0.0).float().double()    # fp32 p_contact, as compiled
        xs = x[agent]
        w = None
        if specs[name].table is not None:
            w = torch.from_numpy(np.asarray(specs[name].table, dtype=np.float32).reshape(2, 200)[0]).to(device).double()[cls]
            xs = w[agent] * xs
        sums = torch.zeros(len(es["people"]), dtype=torch.float64, device=device).index_add_(0, venue, xs)
        cum[name] = float(np.float32(betas[name])) * pc * sums
        per_agent = torch.zeros(A, dtype=torch.float64, device=device).index_add_(0, agent, cum[name][venue])
        if w is not None and name == "care_visit":          # leisure_network.py:107-120: susceptibility also x (age > 75)
            w = w * ((cls % 100) > 75).double()
        acc += per_agent if w is None else w * per_agent
    return cum, susc * acc


def run_stages(r, sample=False):
    p = r.params()
    for ph in (0, 1, 2, 3 if sample else 4) if r.layout == "tiled" else (0, 1, 3 if sample else 4):
        r.engine.step_phase(r.bufs, p, r.io, ph)
    torch.cuda.synchronize()


@pytest.mark.parametrize("preset,agents", CASES, ids=[f"{p}-{a or 'full'}" for p, a in CASES])
def test_fullsize_properties(device, preset, agents):
    world = cached_world(preset, agents)
    A = world["n_agents"]
    specs, betas = B.network_specs(world), B.betas_of(world)
    noise = torch.empty(2, A).exponential_(generator=torch.Generator().manual_seed(1)).to(device)
    big = A >= 5_000_000           # the graph compiled on the device (bit-identical, test_device_compiled_plan_is_identical)
    tiled = SingleGpuHotPath(world, specs, betas, device, seed=3, layout="tiled", exp_noise=noise, device_compile=big)
    csr = SingleGpuHotPath(world, specs, betas, device, seed=3, layout="csr", exp_noise=noise)

    # -- layouts agree (probabilities), first without touching the state ----------------------------
    run_stages(tiled)
    run_stages(csr)
    pt, pc = tiled.probs.cpu().numpy(), csr.probs.cpu().numpy()
    assert np.abs(pt - pc).max() <= 2e-6
    assert (pt < 1.0).sum() > 0.5 * A            # the world is actually exposed

    # -- EVERY venue and agent against fp64 sums taken on the device from the COO edge lists -----------
    cum64, ts64 = fp64_device_reference(world, tiled, betas, device)
    per_set = {}
    for name in world["networks"]:
        k = per_set.get(edge_set_of(name), 0)
        per_set[edge_set_of(name)] = k + 1
        for r in (tiled, csr):
            got = r.engine.plan.cum_of(edge_set_of(name))[:, k].double()
            err = (got - cum64[name]).abs() / (cum64[name].abs() + 1e-9)
            assert float(err.max()) <= 3e-5, (name, r.layout, float(err.max()))
    p64 = torch.exp(-torch.clamp(ts64, 1e-6, 100.0)).clamp(0.0, 1.0)
    assert float((tiled.probs.double() - p64).abs().max()) <= 2e-6
    del cum64, ts64, p64

    # -- sampled venues / agents against fp64 sums from the COO edge list (host) -----------------------
    x = tiled.state["transmission"].cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(0)
    for name in world["networks"]:
        es_name = edge_set_of(name)
        if es_name == "leisure":
            continue
        es = world["edge_sets"][es_name]
        V = len(es["people"])
        pick = rng.choice(V, size=min(V, 2000), replace=False)
        sel = np.isin(es["venue"], pick)
        sums = np.bincount(es["venue"][sel], weights=x[es["agent"][sel]], minlength=V)[pick]
        pcv = np.clip(1.0 / np.maximum(es["people"][pick].astype(np.float64) - 1, 1e-300), 0, 1)
        pcv[es["people"][pick] == 1] = 1.0
        ref = np.float32(betas[name]) * pcv * sums
        for r in (tiled, csr):
            got = r.engine.plan.cum_of(es_name)[:, 0].cpu().numpy()[pick]
            assert np.allclose(got, ref, rtol=2e-5, atol=1e-9), (name, r.layout)

    # -- mass conservation per set (fp64 check sums) --------------------------------------------------
    for name in world["networks"]:
        es_name = edge_set_of(name)
        if es_name == "leisure":
            continue
        es = world["edge_sets"][es_name]
        deg = np.bincount(es["agent"], minlength=A)
        rhs = float((deg * x).sum())
        pcv = tiled.engine.plan.keep[tiled.engine.plan.host.set_index[es_name]]["v_pc"].double()
        cum = tiled.engine.plan.cum_of(es_name)[:, 0].double()
        ok = pcv > 0
        lhs = float((cum[ok] / (float(np.float32(betas[name])) * pcv[ok])).sum())
        miss = float((np.bincount(es["venue"], weights=x[es["agent"]], minlength=len(es["people"]))[~ok.cpu().numpy()]).sum())
        assert abs(lhs + miss - rhs) <= 1e-5 * max(1.0, rhs), name

    # -- linearity of pass 1: 2x in, exactly 2x out -------------------------------------------------
    p = tiled.params()
    for r in (tiled, csr):
        _on_the_fixed_point_grid(r, p)
        before = [r.engine.plan.cum_of(s.name).clone() for s in r.engine.plan.host.sets]
        r.state["transmission"].mul_(2.0)
        r.engine.venue_reduce(r.bufs, p)
        torch.cuda.synchronize()
        for s, b in zip(r.engine.plan.host.sets, before):
            now = r.engine.plan.cum_of(s.name)
            if s.name != "leisure":
                assert torch.equal(now, 2.0 * b), (r.layout, s.name)
            else:
                assert torch.allclose(now, 2.0 * b, rtol=1e-5, atol=1e-10)   # table product is rounded before the sum

    # -- determinism and decisions: a full step twice from the same state ----------------------------
    outs = []
    for rep in range(2):
        t2 = SingleGpuHotPath(world, specs, betas, device, seed=3, layout="tiled", exp_noise=noise, device_compile=big)
        t2.step()
        torch.cuda.synchronize()
        outs.append((t2.probs.clone(), t2.new_infected.clone(), t2.state["is_infected"].clone()))
        del t2
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "tiled step is not bitwise reproducible"
    csr.step()
    torch.cuda.synchronize()
    dt, dc = outs[0][1].cpu().numpy() > 0.5, csr.new_infected.cpu().numpy() > 0.5
    bad = dt != dc
    if bad.any():     # only where the Gumbel margin is within float noise
        p_ = torch.from_numpy(pc)
        z0 = (p_.log() - noise[0].cpu().log()) / 0.1
        z1 = ((1 - p_).log() - noise[1].cpu().log()) / 0.1
        assert ((z1 - z0).abs().numpy()[bad] < 1e-3).all()
    assert bad.sum() <= 1e-5 * A
    assert abs(int(dt.sum()) - int(dc.sum())) <= bad.sum()


def test_c3_full_size_eight_partitions_equal_unpartitioned(device):
    """BASELINE.json configs[3]'s partitioning at full size on the one GPU there is: the 10 M-agent C3 world as 8 agent
    partitions (each exactly a rank of the multi-GPU path: halo agents, partial venue sums; the two collectives become
    device copies) against the unpartitioned run - bit for bit, over three steps of in-kernel Philox noise."""
    from grad_june_amd.distributed import PartitionedHotPath

    world = cached_world("c3", None)
    specs, betas = B.network_specs(world), B.betas_of(world)
    single = SingleGpuHotPath(world, specs, betas, device, seed=5, layout="tiled", device_compile=True)
    parted = PartitionedHotPath(world, specs, betas, device, parts=8, seed=5, device_compile=True)
    assert {m for rk in parted.ranks for m in rk.rw.modes.values()} == {"halo", "partial"}
    assert sum(rk.rw.n_halo for rk in parted.ranks) > 1_000_000
    for _ in range(3):
        single.step()
        parted.step()
    torch.cuda.synchronize()
    for k, v in parted.state.items():
        assert torch.equal(v, single.state[k]), k
    got_new = torch.cat([rk.new_infected for rk in parted.ranks])
    assert torch.equal(got_new, single.new_infected)
    assert single.state["is_infected"].sum().item() > 0.04 * world["n_agents"]


def _on_the_fixed_point_grid(r, p):
    """Pass 1 sums in 2^-36 fixed point: it is linear bit for bit in transmissions that lie ON that grid.  A transmission
    below 2^-13 carries bits beyond it (one agent in ~1e8 at these sizes: whether a world holds one is luck), so the
    linearity checks first round the transmissions to multiples of 2^-30 and take pass 1 again."""
    x = r.state["transmission"]
    x.copy_(torch.round(x * 2.0**30) / 2.0**30)
    r.engine.venue_reduce(r.bufs, p)
    torch.cuda.synchronize()


@pytest.mark.parametrize("agents", [20_000_000, 100_000_000], ids=["20m", "100m-the-full-config"])
def test_c5_20m_against_fp64_device_sums(device, agents):
    """BASELINE.json configs[4]'s shape on ONE GPU as a single partition - at 20 M agents and at the configuration's
    full size, 100 M agents / 129 M venues / 900 M set-edges (11 s on an MI355X: the world is drawn and the graph
    compiled on the device) -, power-law venue sizes up to 50 000 attendees (a venue then spans every slice and holds far more edges
    than a slice has agents: the LDS-overflow / multi-block case).  The world is drawn on the device
    (synthetic.make_world_torch: the numpy generator needs minutes at this size) and EVERY venue sum and EVERY agent's
    probability is checked against fp64 sums taken from the same edge lists - in particular every venue with more
    than 20 480 attendees; plus linearity of pass 1 and run-to-run determinism."""
    from grad_june_amd.synthetic import make_world_torch

    world = make_world_torch("c5", agents, seed=1234, device=device, infected_fraction=0.03)
    A = world["n_agents"]
    giant = {k: int((v["people"] > 20480).sum()) for k, v in world["edge_sets"].items()}
    assert all(n >= 10 for n in giant.values()), giant
    print(f"c5 world: {A} agents, {sum(len(v['people']) for v in world['edge_sets'].values())} venues, "
          f"{sum(v['agent'].numel() for v in world['edge_sets'].values())} set-edges, venues above 20 480 attendees: {giant}")
    specs, betas = B.network_specs(world), B.betas_of(world)
    r = SingleGpuHotPath(world, specs, betas, device, seed=3, layout="tiled", device_compile=True)
    run_stages(r)
    cum64, ts64 = fp64_device_reference(world, r, betas, device)
    per_set = {}
    for name in world["networks"]:
        es_name = edge_set_of(name)
        k = per_set.get(es_name, 0)
        per_set[es_name] = k + 1
        got = r.engine.plan.cum_of(es_name)[:, k].double()
        err = (got - cum64[name]).abs() / (cum64[name].abs() + 1e-9)
        assert float(err.max()) <= 3e-5, (name, float(err.max()))
        big = torch.from_numpy(world["edge_sets"][es_name]["people"] > 20480).to(device)
        assert float(err[big].max()) <= 3e-5 and float(cum64[name][big].abs().max()) > 0
    p64 = torch.exp(-torch.clamp(ts64, 1e-6, 100.0)).clamp(0.0, 1.0)
    assert float((r.probs.double() - p64).abs().max()) <= 2e-6
    assert int((r.probs < 1.0).sum()) > 0.5 * A
    del cum64, ts64, p64
    # pass 1 is linear, bit for bit (fixed-point sums)
    _on_the_fixed_point_grid(r, r.params())
    before = [r.engine.plan.cum_of(s.name).clone() for s in r.engine.plan.host.sets]
    r.state["transmission"].mul_(2.0)
    r.engine.venue_reduce(r.bufs, r.params())
    torch.cuda.synchronize()
    for s, b in zip(r.engine.plan.host.sets, before):
        now = r.engine.plan.cum_of(s.name)
        assert torch.equal(now, 2.0 * b) if s.name != "leisure" else torch.allclose(now, 2.0 * b, rtol=1e-5, atol=1e-10)
    del before
    # a full step twice from the same start: identical
    start = {k: r.state[k].clone() for k in ("is_infected", "susceptibility", "infection_time")}
    outs = []
    for rep in range(2):
        for k, v in start.items():
            r.state[k].copy_(v)
        r.t = 0
        r.step()
        torch.cuda.synchronize()
        outs.append((r.probs.clone(), r.new_infected.clone(), r.state["is_infected"].clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    assert outs[0][1].sum().item() > 1000


def test_c5_full_size_seven_partitions_equal_one(device):
    """BASELINE.json configs[4] at its full size - 100 M agents, power-law venues up to 50 000 attendees - stepped as
    SEVEN rank-like partitions on the one GPU against the single partition: every partition exactly a rank of the
    multi-GPU run (per-venue exchange classes: every set comes as a local + halo half and a partial-sum half, the two
    collectives become device copies / fp32 sums in rank order).  The world is drawn on the device
    (synthetic.make_world_torch) and the partitions are cut out there (RankPartitioner's torch path: what
    ``bench.py --gpus 8 --preset c5`` does per rank).  Same discrete state after three Philox steps, up to the handful of
    decisions that may sit within fp32 rounding of a partial sum."""
    import time

    from grad_june_amd.distributed import PartitionedHotPath
    from grad_june_amd.synthetic import make_world_torch

    t0 = time.time()
    world = make_world_torch("c5", 100_000_000, seed=1234, device=device, infected_fraction=0.03)
    specs, betas = B.network_specs(world), B.betas_of(world)
    parted = PartitionedHotPath(world, specs, betas, device, parts=7, seed=5, device_compile=True)
    t_parts = time.time() - t0
    modes = {m for rk in parted.ranks for m in rk.rw.modes.values()}
    assert modes == {"halo", "partial"} and any(n.endswith("~big") for n in parted.ranks[0].rw.edge_sets)
    for _ in range(3):
        parted.step()
    torch.cuda.synchronize()
    got = {k: v.clone() for k, v in parted.state.items()}
    halo = [rk.rw.n_halo for rk in parted.ranks]
    del parted
    torch.cuda.empty_cache()
    single = SingleGpuHotPath(world, specs, betas, device, seed=5, layout="tiled", device_compile=True)
    for _ in range(3):
        single.step()
    torch.cuda.synchronize()
    print(f"c5 at 1e8 agents as 7 partitions: built in {t_parts:.0f} s, halo agents per partition {min(halo)}..{max(halo)}")
    differ = int((got["is_infected"] != single.state["is_infected"]).sum())
    assert differ <= 10, differ
    assert abs(float(got["is_infected"].double().sum()) - float(single.state["is_infected"].double().sum())) <= 10
    assert single.state["is_infected"].sum().item() > 0.031 * world["n_agents"]


def test_clustered_world_partitions_equal_unpartitioned(device):
    """A world with a geography (synthetic.GEOGRAPHY) under the household-major order, 2 M agents as 8 partitions:
    households are rank-local (no halo set-up for them at all), the venues that reach across a partition boundary are
    halo or partial-sum venues one by one - and the run ends in the unpartitioned run's state."""
    from grad_june_amd.distributed import PartitionedHotPath
    from grad_june_amd.synthetic import reorder_agents

    world = reorder_agents(make_world("c3", n_agents=2_000_000, seed=21, infected_fraction=0.03, geography="clustered"),
                           by="household")
    specs, betas = B.network_specs(world), B.betas_of(world)
    single = SingleGpuHotPath(world, specs, betas, device, seed=5, layout="tiled", device_compile=True)
    parted = PartitionedHotPath(world, specs, betas, device, parts=8, seed=5, device_compile=True)
    rw0 = parted.ranks[0].rw
    assert rw0.modes["household"] == "halo" and "household~big" not in rw0.edge_sets
    assert max(rk.rw.n_halo for rk in parted.ranks) < 0.25 * rw0.n_local          # the random world: ~1.1x the owned agents
    assert len(rw0.edge_sets["company~big"]["people"]) < len(world["edge_sets"]["company"]["people"])
    for _ in range(3):
        single.step()
        parted.step()
    torch.cuda.synchronize()
    differ = int((parted.state["is_infected"] != single.state["is_infected"]).sum())
    assert differ <= 2, differ
    assert single.state["is_infected"].sum().item() > 0.035 * world["n_agents"]


def test_c2_full_size_against_the_oracle(device):
    """BASELINE.json configs[1] at full size (1 M agents, household/school/company, 15 M edges): one
    whole step on the GPU against the CPU oracle on identical injected noise - per-agent probabilities
    within 1e-5, infection decisions identical away from Gumbel ties, equal infection counts."""
    import gj_oracle as O

    world = cached_world("c2", None, infected=0.02)
    A = world["n_agents"]
    specs, betas = B.network_specs(world), B.betas_of(world)
    noise = O.draw_exp_noise(A, generator=torch.Generator().manual_seed(7))
    w = {"n_agents": A, "age": torch.from_numpy(world["age"]), "sex": torch.from_numpy(world["sex"]),
         "edge_sets": {k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in world["edge_sets"].items()}}
    st = {k: torch.from_numpy(v.copy()) for k, v in world["state"].items()}
    torch.set_num_threads(16)
    ref = O.hot_path_step(w, st, now=1.0, delta_time=1.0, day_type=0, active=world["networks"], betas=betas,
                          quarantine_thresholds=None, exp_noise=noise)
    for layout in ("tiled", "csr"):
        r = SingleGpuHotPath(world, specs, betas, device, seed=0, layout=layout, exp_noise=noise.to(device))
        r.step()
        torch.cuda.synchronize()
        p = r.probs.cpu().numpy()
        assert np.abs(p - ref["not_infected_probs"].numpy()).max() <= 1e-5, layout
        dec, dref = r.new_infected.cpu().numpy() > 0.5, ref["new_infected"].numpy() > 0.5
        bad = dec != dref
        if bad.any():
            pr = ref["not_infected_probs"]
            margin = (((1 - pr).log() - noise[1].log()) / 0.1 - (pr.log() - noise[0].log()) / 0.1).abs().numpy()
            assert (margin[bad] < 1e-3).all(), layout
        assert bad.sum() <= 3, layout
        assert abs(int(dec.sum()) - int(dref.sum())) <= bad.sum()
        assert dref.sum() > 1000
        assert np.allclose(r.state["transmission"].cpu().numpy(), ref["transmission"].numpy(), rtol=2e-5, atol=1e-9)


def test_step_is_capturable_in_a_hip_graph(device):
    """gj_step is four plain launches on the caller's stream (no host synchronisation, no allocation), so it
    can be captured in a hipGraph and replayed: a replay equals the eager step bit for bit."""
    world = make_world("c3", n_agents=300_000, seed=7, infected_fraction=0.05)
    specs, betas = B.network_specs(world), B.betas_of(world)
    r = SingleGpuHotPath(world, specs, betas, device, seed=5, layout="tiled")
    keys = ("is_infected", "susceptibility", "infection_time")
    start = {k: r.state[k].clone() for k in keys}
    r.step()                                              # eager step 0
    torch.cuda.synchronize()
    eager = {k: r.state[k].clone() for k in keys}
    eager_new = r.new_infected.clone()
    assert eager_new.sum().item() > 0
    for k in keys:
        r.state[k].copy_(start[k])
    r.t = 0
    p = r.params()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                         # warm-up on the capture stream
        r.engine.step(r.bufs, p, r.io)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        r.engine.step(r.bufs, p, r.io)
    for _ in range(2):                                    # replay twice from the same start state
        for k in keys:
            r.state[k].copy_(start[k])
        r.new_infected.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(r.state[k], eager[k]), k
        assert torch.equal(r.new_infected, eager_new)


def test_device_compiled_plan_is_identical(device):
    """The contact graph compiled on the GPU (gj_compile_* kernels, row f4) equals the numpy-compiled one array for
    array, and a step on it gives bitwise the same state."""
    from grad_june_amd.plan import _host, compile_plan

    world = make_world("c3", n_agents=500_000, seed=11, infected_fraction=0.05)
    specs, betas = B.network_specs(world), B.betas_of(world)
    host = compile_plan(world["n_agents"], world["edge_sets"], age=world["age"], sex=world["sex"], layout="tiled")
    dev = compile_plan(world["n_agents"], world["edge_sets"], age=world["age"], sex=world["sex"], layout="tiled",
                       device=device)
    assert np.array_equal(dev.work, host.work)
    for a, b in zip(dev.sets, host.sets):
        assert a.tiled.e_lv.device.type == "cuda"
        for k in ("blk_v0", "blk_e0", "e_cls", "tile_sptr", "tile_jpos", "chunk_ptr", "chunk_desc", "multi_slots"):
            x, y = getattr(a.tiled, k), getattr(b.tiled, k)
            assert (x is None) == (y is None) and (x is None or np.array_equal(_host(x), y)), (a.name, k)
        for k in ("e_lv", "a_la"):
            assert np.array_equal(_host(getattr(a.tiled, k)).view(np.uint16), getattr(b.tiled, k)), (a.name, k)
    r0 = SingleGpuHotPath(world, specs, betas, device, seed=2, layout="tiled")
    r1 = SingleGpuHotPath(world, specs, betas, device, seed=2, layout="tiled", device_compile=True)
    for _ in range(2):
        r0.step()
        r1.step()
    torch.cuda.synchronize()
    for k in ("is_infected", "susceptibility", "infection_time"):
        assert torch.equal(r0.state[k], r1.state[k]), k
    assert torch.equal(r0.probs, r1.probs)


def test_the_arrangement_bench_times(device):
    """The EXACT world and arrangement of the driver's bench line: C3, seed 1234, 1 % infected, agents renumbered
    household-major (bench.py's default `--reorder`), graph compiled on the device, Philox noise keyed by seed 1234.
      (a) 25 steps (the driver's --warmup 5 --steps 20) end with is_infected.sum() == 2 958 814 - the checksum the
          builder's profiled run and the driver's BENCH_r02 run both printed: the kernels that are timed compute THIS;
      (b) the CSR layout stepped from the same states with the same Philox stream takes the same decisions except
          where a probability sits within float rounding of its threshold;
      (c) every venue sum and every agent's probability of that arrangement against fp64 sums taken on the device from
          the renumbered COO lists (diagonal household tiles, wide descriptors and all)."""
    from grad_june_amd.synthetic import reorder_agents

    world = reorder_agents(make_world("c3", seed=1234, infected_fraction=0.01), by="household")
    A = world["n_agents"]
    specs, betas = B.network_specs(world), B.betas_of(world)
    tiled = SingleGpuHotPath(world, specs, betas, device, seed=1234, layout="tiled", device_compile=True)
    assert any(hs.tiled.desc_wide for hs in tiled.engine.plan.host.sets)      # the arrangement has small-tile sets
    # (c) first: stateless stages on the initial state
    run_stages(tiled)
    cum64, ts64 = fp64_device_reference(world, tiled, betas, device)
    per_set = {}
    for name in world["networks"]:
        k = per_set.get(edge_set_of(name), 0)
        per_set[edge_set_of(name)] = k + 1
        got = tiled.engine.plan.cum_of(edge_set_of(name))[:, k].double()
        err = (got - cum64[name]).abs() / (cum64[name].abs() + 1e-9)
        assert float(err.max()) <= 3e-5, (name, float(err.max()))
    p64 = torch.exp(-torch.clamp(ts64, 1e-6, 100.0)).clamp(0.0, 1.0)
    assert float((tiled.probs.double() - p64).abs().max()) <= 2e-6
    del cum64, ts64, p64
    # (b) + (a): 25 production steps; the CSR runner is teacher-forced with the tiled runner's state before each of the
    # first steps (a flipped tie would otherwise send the two epidemics apart)
    csr = SingleGpuHotPath(world, specs, betas, device, seed=1234, layout="csr")
    keys = ("is_infected", "susceptibility", "infection_time", "current_stage")
    flips = 0
    for step in range(25):
        if step < 4:
            for k in keys:
                csr.state[k].copy_(tiled.state[k])
            csr.t = tiled.t
            csr.step()
        tiled.step()
        if step < 4:
            torch.cuda.synchronize()
            differ = tiled.new_infected != csr.new_infected
            n = int(differ.sum())
            flips += n
            assert n <= 8, (step, n)
            assert float((tiled.probs - csr.probs).abs().max()) <= 2e-6
            if n:        # only where the probability is within rounding of the agent's threshold
                from gj_philox_ref import infection_uniform

                idx = torch.nonzero(differ).flatten().cpu().numpy()
                theta = infection_uniform(1234, step, idx.astype(np.int64))
                assert np.abs(tiled.probs.cpu().numpy()[idx] - theta).max() <= 2e-6
            assert abs(float(tiled.new_infected.sum()) - float(csr.new_infected.sum())) <= 8
    torch.cuda.synchronize()
    assert float(tiled.state["is_infected"].double().sum()) == 2958814.0
    print(f"decisions that differ between the layouts over 4 teacher-forced steps: {flips}")
