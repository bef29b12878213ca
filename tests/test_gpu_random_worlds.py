"""GPU parity on RANDOM small worlds: the fused production step (through the C ABI, tiled and CSR layouts, random tile
geometries) against the CPU oracle on the same seeded inputs and the same injected noise.

The golden fixtures pin the reference's values on the reference's own worlds; this file covers what those worlds do not
have - edge sets without edges, venues nobody attends, `people` that differs from the degree (p_contact is computed from
`people`, base.py:63-70), duplicated (agent, venue) pairs, one venue holding a third of a set, agents without any edge,
every subset of the eleven networks, both day types, quarantine on and off - in combinations nobody wrote down.

Tolerances as in test_gpu_golden_parity.py: transmissions rtol 2e-5, probabilities atol 1e-5, decisions identical except
where the oracle's Gumbel margin is below 1e-3, post-state equal wherever the decisions are."""
import numpy as np
import pytest
import torch

import gj_testlib as L

pytestmark = pytest.mark.gpu

SIZES = (37, 64, 200, 1000, 3000, 6000)


def random_world(rng):
    A = int(rng.choice(SIZES)) + int(rng.integers(0, 30))
    names = [s for s in L.EDGE_SETS if rng.random() < 0.7] or ["household"]
    sets = {}
    for s in names:
        V = int(rng.integers(1, max(2, A // int(rng.choice([1, 3, 20, 200])))))
        E = 0 if rng.random() < 0.1 else int(rng.integers(1, 4 * A))
        venue = rng.integers(0, V, E)
        if E and rng.random() < 0.3:
            venue[: E // 3] = int(rng.integers(0, V))             # one venue with a third of the set
        agent = rng.integers(0, max(1, int(A * rng.choice([1.0, 0.5]))), E)    # 0.5: half the agents without an edge here
        people = np.bincount(venue, minlength=V)
        if rng.random() < 0.5:                                    # `people` is data of its own, not the degree
            people = people + rng.integers(0, 3, V)
        sets[s] = {"agent": torch.from_numpy(agent.astype(np.int64)), "venue": torch.from_numpy(venue.astype(np.int64)),
                   "people": torch.from_numpy(people.astype(np.int64))}
    return {"n_agents": A, "age": torch.from_numpy(rng.integers(0, 100, A).astype(np.int64)),
            "sex": torch.from_numpy(rng.integers(0, 2, A).astype(np.int64)), "edge_sets": sets}


def random_state(rng, A, now):
    inf = (rng.random(A) < rng.choice([0.0, 0.02, 0.3, 0.9])).astype(np.float32)
    inf[rng.random(A) < 0.01] = 2.0                               # `is_infected` is additive (model.py:103-110)
    f32 = lambda x: torch.from_numpy(np.asarray(x, dtype=np.float32))
    return {"max_infectiousness": f32(rng.lognormal(0.0, 0.5, A)), "shape": f32(rng.normal(1.56, 0.08, A)),
            "rate": f32(rng.normal(0.53, 0.03, A)), "shift": f32(rng.normal(-2.12, 0.1, A)),
            "infection_time": f32((now - 15.0 * rng.random(A)) * (inf > 0)), "is_infected": f32(inf),
            "susceptibility": f32(np.where(inf > 0, 0.0, rng.choice([1.0, 0.4], A))),
            "current_stage": f32(rng.integers(1, 7, A))}


def random_layout(rng, A):
    if rng.random() < 0.2:
        return "csr", {}
    kw = {}
    if rng.random() < 0.7:
        kw["sv_max"] = int(rng.choice([16, 64, 256]))
        kw["eb_target"] = int(rng.choice([64, 512, 4096]))
    if rng.random() < 0.7:
        sa = int(rng.choice([64, 128, 512]))
        kw["slices"] = (-(-A // sa), sa)
    r = rng.random()
    if r < 0.25:
        kw["direct"] = False
    if rng.random() < 0.3:
        kw["desc_wide"] = bool(rng.random() < 0.5)
    elif rng.random() < 0.2:
        kw["desc_explicit"] = True
    if rng.random() < 0.2:
        kw["split_epilogue"] = True
    return "tiled", kw


@pytest.mark.parametrize("seed", range(150))
def test_random_world_against_the_oracle(device, seed):
    import gj_oracle as O
    from grad_june_amd.engine import AgentBuffers

    rng = np.random.default_rng(1000 + seed)
    world = random_world(rng)
    A = world["n_agents"]
    now, dt, day_type = float(rng.choice([1.0, 7.5, 30.0])), float(rng.choice([1.0, 0.5, 1.0 / 3.0])), int(rng.integers(0, 2))
    tables = {n: torch.from_numpy(rng.random((2, 2, 100)).astype(np.float32)) for n in L.LEISURE + ("care_visit",)
              if rng.random() < 0.8}
    specs = L.network_specs(world, tables)
    if not specs:
        pytest.skip("the draw has no network on any of its sets")
    active = [s.name for s in specs if rng.random() < 0.8] or [specs[0].name]
    hot = rng.random() < 0.3                                      # a third of the draws: an epidemic that takes off
    betas = {n: float(rng.uniform(3.0, 40.0) if hot else rng.uniform(0.1, 3.0)) for n in active}
    thr = None if rng.random() < 0.5 else [float(rng.choice([2.0, 3.0, 4.0]))]
    state = random_state(rng, A, now)
    noise = O.draw_exp_noise(A, generator=torch.Generator().manual_seed(seed))
    ref = O.hot_path_step(world, {k: v.clone() for k, v in state.items()}, now=now, delta_time=dt, day_type=day_type,
                          active=active, betas=betas, leisure_tables=tables, quarantine_thresholds=thr, exp_noise=noise)

    layout, kw = random_layout(rng, A)
    engine = L.make_engine(world, tables, device, layout=layout, **kw)
    st = L.device_state(state, device)
    bufs = AgentBuffers(engine.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"], rate=st["rate"],
                        shift=st["shift"], infection_time=st["infection_time"], is_infected=st["is_infected"],
                        susceptibility=st["susceptibility"], transmission=st["transmission"],
                        current_stage=st["current_stage"])
    p = engine.params(now=now, delta_time=dt, day_type=day_type, active=active, betas=betas, has_quarantine=thr is not None,
                      q_threshold=L.q_threshold(thr))
    probs, new = torch.empty(A, device=device), torch.empty(A, device=device)
    engine.step(bufs, p, engine.io(not_infected_probs=probs, new_infected=new, exp_noise=noise.to(device).contiguous()))
    torch.cuda.synchronize()
    what = f"seed {seed}: {A} agents, sets {list(world['edge_sets'])}, active {active}, q {thr}, {layout} {kw}"
    assert np.allclose(st["transmission"].cpu().numpy(), ref["transmission"].numpy(), rtol=2e-5, atol=1e-9), what
    pr = ref["not_infected_probs"].numpy()
    assert np.abs(probs.cpu().numpy() - pr).max() <= 1e-5, what
    dec, dref = new.cpu().numpy() > 0.5, ref["new_infected"].numpy() > 0.5
    bad = dec != dref
    if bad.any():
        t = torch.from_numpy(pr)
        margin = (((1 - t).log() - noise[1].log()) / 0.1 - (t.log() - noise[0].log()) / 0.1).abs().numpy()
        assert (margin[bad] < 1e-3).all(), what
    ok = ~bad
    for k in ("susceptibility", "is_infected", "infection_time"):
        assert np.allclose(st[k].cpu().numpy()[ok], ref[k].numpy()[ok], rtol=1e-6, atol=1e-6), (what, k)


# ---- the same kind of world cut into partitions (= the ranks of the multi-GPU path, collectives stood in on the device) ----
JUNE_NETWORKS = ["school", "university", "company", "care_home", "pub", "gym", "grocery", "visit", "care_visit", "cinema",
                 "household"]


def random_bench_world(rng):
    """A random world in the form bench.py / distributed.py take (numpy arrays, a `networks` list, a `state`)."""
    from grad_june_amd.synthetic import edge_set_of

    w = random_world(rng)
    A = w["n_agents"]
    sets = {k: {kk: vv.numpy() for kk, vv in v.items()} for k, v in w["edge_sets"].items()}
    nets = [n for n in JUNE_NETWORKS if edge_set_of(n) in sets and rng.random() < 0.8] or \
           [n for n in JUNE_NETWORKS if edge_set_of(n) in sets][:1]
    st = random_state(rng, A, 1.0)
    return {"preset": "random", "n_agents": A, "age": w["age"].numpy(), "sex": w["sex"].numpy(), "networks": nets,
            "edge_sets": sets, "state": {k: v.numpy() for k, v in st.items()}}


@pytest.mark.parametrize("seed", range(60))
def test_random_world_partitions_equal_the_unpartitioned_run(device, seed, monkeypatch):
    """2-8 partitions of a random world (per-venue exchange classes; in half of the draws with the split thresholds
    lowered so that small sets are cut into a halo half and a partial-sum half too) against the single partition: the
    same state after three steps.  Halo and local venues are bit-equal across partitions, partial-sum venues agree to
    fp32 rounding of the partitions' terms - a decision can flip only within that of its Philox threshold."""
    import bench as B
    from grad_june_amd import distributed as D
    from grad_june_amd.benchrun import SingleGpuHotPath

    rng = np.random.default_rng(5000 + seed)
    world = random_bench_world(rng)
    if rng.random() < 0.5:
        monkeypatch.setattr(D, "MIN_SPLIT_HALO_FLOATS", 8)
        monkeypatch.setattr(D, "MIN_SPLIT_HALO_SHARE", 0.01)
        monkeypatch.setattr(D, "MIN_SPLIT_VENUE_SHARE", 0.01)
    if rng.random() < 0.25:
        monkeypatch.setattr(D, "EXCHANGE_RULE", "set")
    R = int(rng.choice([2, 3, 5, 8]))
    specs = B.network_specs(world)
    betas = {n: float(rng.uniform(0.5, 20.0)) for n in world["networks"]}
    single = SingleGpuHotPath(world, specs, betas, device, seed=seed, layout="tiled")
    parted = D.PartitionedHotPath(world, specs, betas, device, parts=R, seed=seed)
    for _ in range(3):
        single.step()
        parted.step()
    torch.cuda.synchronize()
    what = f"seed {seed}: {world['n_agents']} agents, {R} partitions, modes {parted.ranks[0].rw.modes}"
    ps = parted.state
    differ = int((ps["is_infected"] != single.state["is_infected"]).sum())
    assert differ <= 2, (what, differ)
    if differ == 0:
        for k in ("susceptibility", "infection_time"):
            assert torch.equal(ps[k], single.state[k]), (what, k)


# ---- gradients (row f3) on random worlds: the HIP backward against autograd through the oracle -------------------------
def _hetero(G, world, state, device):
    d = G.HeteroData()
    A = world["n_agents"]
    ag = d["agent"]
    ag.id = torch.arange(A)
    ag.age, ag.sex = world["age"], world["sex"]
    for s, es in world["edge_sets"].items():
        d[s].id = torch.arange(len(es["people"]))
        d[s].people = es["people"]
        d["agent", "attends_" + s, s].edge_index = torch.vstack((es["agent"], es["venue"]))
    d = d.to(device)
    st = {k: v.to(device) for k, v in state.items()}
    ag.infection_parameters = {k: st[k] for k in ("max_infectiousness", "shape", "rate", "shift")}
    for k in ("is_infected", "susceptibility", "infection_time"):
        ag[k] = st[k].clone()
    ag.transmission = torch.zeros(A, device=device)
    ag.symptoms = {"current_stage": st["current_stage"].clone(), "next_stage": st["current_stage"].clone(),
                   "time_to_next_stage": torch.full((A,), 1e9, device=device)}
    return d


@pytest.mark.parametrize("seed", range(30))
def test_random_world_gradients_against_oracle_autograd(device, seed):
    """d (cases after the last step, and summed over the steps) / d log_beta of every network through three chained
    timesteps of `GradJune.hot_path` in grad mode (hand-written HIP adjoints) against torch autograd through the oracle
    on the same injected noise - on worlds with empty sets, unattended venues, `is_infected` = 2, any subset of the
    networks and a quarantine policy in half of the draws.  The oracle's autograd is pinned to the reference's recorded
    gradients within 2e-4 (tests/test_gradients.py); the bound here is 1e-3 of the largest gradient of the draw (+ 1e-6)."""
    import gj_oracle as O
    import grad_june_amd as G
    from grad_june_amd.defaults import default_parameters
    from grad_june_amd.synthetic import edge_set_of

    rng = np.random.default_rng(9000 + seed)
    world = random_world(rng)
    A = world["n_agents"]
    names = [n for n in JUNE_NETWORKS if edge_set_of(n) in world["edge_sets"] and rng.random() < 0.8]
    if not names:
        pytest.skip("the draw has no network on any of its sets")
    params = default_parameters(str(device))
    params["networks"] = {n: {"log_beta": float(rng.uniform(-0.3, 1.2))} for n in names}
    params["policies"] = {"interaction": {}}
    thr = None
    if rng.random() < 0.5:
        thr = float(rng.choice([3.0, 4.0]))
        params["policies"]["quarantine"] = {
            "quarantine": {1: {"start_date": "2022-01-01", "end_date": "2022-12-31", "stage_threshold": thr}}}
    model = G.GradJune.from_parameters(params)
    acts = (tuple(names),)
    timer = G.Timer(initial_day="2022-02-01", total_days=10, weekday_step_duration=(24,), weekend_step_duration=(24,),
                    weekday_activities=acts, weekend_activities=acts)
    state = random_state(rng, A, 0.0)
    data = _hetero(G, world, state, device)
    for n in names:
        net = model.infection_networks.networks[n]
        net.log_beta = torch.nn.Parameter(net.log_beta.detach().clone())
    tables = {n: model.infection_networks.networks[n].leisure_probabilities.detach().cpu()
              for n in names if edge_set_of(n) == "leisure"}
    mult = {n: torch.ones((), requires_grad=True) for n in names}
    st = {k: v.clone() for k, v in state.items()}
    hip_series, ref_series = [], []
    for i in range(3):
        next(timer)
        noise = O.draw_exp_noise(A, generator=torch.Generator().manual_seed(100 * seed + i))
        betas = {n: float(model.infection_networks[n].beta_value(model.policies, timer)) for n in names}
        model.hot_path(data, timer, exp_noise=noise)
        hip_series.append(data["agent"].is_infected.sum())
        out = O.hot_path_step(world, st, now=timer.now, delta_time=timer.duration,
                              day_type=0 if timer.day_type == "weekday" else 1, active=names,
                              betas={n: torch.tensor(np.float32(betas[n])) * mult[n] for n in names},
                              leisure_tables=tables, quarantine_thresholds=None if thr is None else [thr], exp_noise=noise)
        for k in ("susceptibility", "is_infected", "infection_time"):
            st[k] = out[k]
        ref_series.append(out["is_infected"].sum())
        if not np.array_equal(data["agent"].is_infected.detach().cpu().numpy(), out["is_infected"].detach().numpy()):
            pytest.skip(f"step {i}: a decision at a Gumbel tie differs - the two graphs are not the same function")
    ps = [model.infection_networks.networks[n].log_beta for n in names]
    for tag, hip, ref in (("last", hip_series[-1], ref_series[-1]),
                          ("series", torch.stack(hip_series).sum(), torch.stack(ref_series).sum())):
        if not hip.requires_grad:                        # nobody infectious meets anybody susceptible: no graph
            assert not ref.requires_grad or all(g is None or float(g) == 0.0 for g in torch.autograd.grad(
                ref, list(mult.values()), retain_graph=True, allow_unused=True)), tag
            continue
        got = [0.0 if g is None else float(g) for g in torch.autograd.grad(hip, ps, retain_graph=True, allow_unused=True)]
        want = [0.0 if g is None else float(g) * np.log(10.0)
                for g in torch.autograd.grad(ref, [mult[n] for n in names], retain_graph=True, allow_unused=True)]
        scale = max(1e-6, max(abs(w) for w in want))
        for n, a, b in zip(names, got, want):
            # (+ 1e-6 absolute: a draw whose gradients are all ~1e-6 lives on agents at the probability floor, where
            # 1 - p itself carries a relative fp32 error of several per cent in either implementation)
            assert abs(a - b) <= 1e-3 * scale + 1e-6, (seed, tag, n, a, b, scale)


# ---- the graph compile (row f4) on random worlds: the library's kernels against the numpy specification ----------------
@pytest.mark.parametrize("seed", range(60))
def test_random_world_device_compile_equals_the_numpy_compile(device, seed):
    """`compile_plan(device=...)` (csrc/gj_compile.hip behind the C ABI) against `compile_plan` with numpy on the same random
    world and the same random geometry: the work list and every array of every set bit for bit - tiles, descriptors (narrow,
    wide, explicit slots, multi-slot rows), ELL rows - incl. sets without edges and venues without attendees."""
    from grad_june_amd.plan import _host, compile_plan
    from test_gpu_compile_native import assert_same_tiled

    rng = np.random.default_rng(7000 + seed)
    world = random_bench_world(rng)
    A = world["n_agents"]
    _, kw = random_layout(rng, A)
    kw.pop("split_epilogue", None)
    host = compile_plan(A, world["edge_sets"], age=world["age"], sex=world["sex"], layout="tiled", **kw)
    dev = compile_plan(A, {k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in world["edge_sets"].items()},
                       age=world["age"], sex=world["sex"], layout="tiled", device=device, **kw)
    what = f"seed {seed}: {A} agents, {kw}"
    assert np.array_equal(dev.work, host.work) and dev.n_slices == host.n_slices, what
    for a, b in zip(dev.sets, host.sets):
        assert (a.name, a.n_venues, a.n_edges) == (b.name, b.n_venues, b.n_edges), what
        assert_same_tiled(a.tiled, b.tiled, what + " " + a.name)
        assert a.tiled.ell_k == b.tiled.ell_k, (what, a.name)
        if b.tiled.ell is not None:
            assert np.array_equal(_host(a.tiled.ell).view(np.uint16), b.tiled.ell), (what, a.name)


# ---- medium-sized benchmark worlds in random geometries against the oracle ----------------------------------------------
MEDIUM = [("c3", "random", 60_000), ("c3", "clustered", 150_000), ("c5", "random", 120_000), ("c5", "clustered", 300_000),
          ("june", "clustered", 100_000), ("june", "clustered", 250_000), ("c2", "random", 200_000), ("c2", "clustered", 80_000)]


@pytest.mark.parametrize("preset,geography,n_agents", MEDIUM)
@pytest.mark.parametrize("variant", [0, 1])
def test_medium_world_in_a_random_geometry_against_the_oracle(device, preset, geography, n_agents, variant):
    """The benchmark presets (random and clustered geographies; power-law venues; the eleven networks of the `june`
    preset) at 60 k - 300 k agents - large enough for wide descriptors, rows of explicit slots, wave-wide venue runs, the
    run form of the households and several slices of up to 19 840 agents - in a random tile geometry, one fused step
    against the oracle on injected noise, with a quarantine policy in every second case."""
    import bench as B
    import gj_oracle as O
    from grad_june_amd.benchrun import SingleGpuHotPath
    from grad_june_amd.synthetic import make_world, reorder_agents

    import zlib

    rng = np.random.default_rng(zlib.crc32(f"{preset} {geography} {n_agents} {variant}".encode()))
    world = make_world(preset, n_agents=n_agents, seed=int(rng.integers(1 << 30)), infected_fraction=0.05, geography=geography)
    if variant:
        world = reorder_agents(world, by="household")
    specs = B.network_specs(world)
    betas = {n: 4.0 * v for n, v in B.betas_of(world).items()}
    kw = {}
    if rng.random() < 0.6:
        kw["sv_max"], kw["eb_target"] = int(rng.choice([256, 2048, 16384])), int(rng.choice([2048, 32768, 131072]))
    if rng.random() < 0.6:
        sa = int(rng.choice([1024, 4928, 19840]))
        kw["slices"] = (-(-n_agents // sa), sa)
    if rng.random() < 0.3:
        kw["direct"] = False
    thr = 4.0 if variant else None
    A = world["n_agents"]
    noise = O.draw_exp_noise(A, generator=torch.Generator().manual_seed(n_agents + variant))
    w = {"n_agents": A, "age": torch.from_numpy(world["age"]), "sex": torch.from_numpy(world["sex"]),
         "edge_sets": {k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in world["edge_sets"].items()}}
    st = {k: torch.from_numpy(v.copy()) for k, v in world["state"].items()}
    tables = {s.name: torch.from_numpy(s.table) for s in specs if s.table is not None}
    ref = O.hot_path_step(w, st, now=1.0, delta_time=1.0, day_type=0, active=world["networks"], betas=betas,
                          leisure_tables=tables, quarantine_thresholds=None if thr is None else [thr], exp_noise=noise)
    r = SingleGpuHotPath(world, specs, betas, device, seed=0, layout="tiled", exp_noise=noise.to(device),
                         quarantine_threshold=thr, **kw)
    r.step()
    torch.cuda.synchronize()
    what = f"{preset} {geography} {n_agents} variant {variant}: {kw}"
    pr = ref["not_infected_probs"].numpy()
    assert np.abs(r.probs.cpu().numpy() - pr).max() <= 1e-5, what
    dec, dref = r.new_infected.cpu().numpy() > 0.5, ref["new_infected"].numpy() > 0.5
    bad = dec != dref
    if bad.any():
        t = torch.from_numpy(pr)
        margin = (((1 - t).log() - noise[1].log()) / 0.1 - (t.log() - noise[0].log()) / 0.1).abs().numpy()
        assert (margin[bad] < 1e-3).all() and bad.sum() <= 3, what
    assert dref.sum() > 0.002 * A, (what, int(dref.sum()))
    assert np.allclose(r.state["transmission"].cpu().numpy(), ref["transmission"].numpy(), rtol=2e-5, atol=1e-9), what
