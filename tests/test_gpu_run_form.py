"""GPU: the "run form" of the edge set that orders the agents (tiling.split_primary_runs; gj_tiled_set.run_*).

Under the household-major agent order one edge per agent - its primary edge, to its smallest venue - leaves the tiled
arrays: phase B reads its value from the transmission array itself, phase D reads the venue's cum through a per-slice
window.  Checked here:
  * the compile kernels (gj_compile_runs_*) against the numpy specification, array for array;
  * the reference's recorded trajectories of its 769-agent world, renumbered household-major (results mapped back):
    venue sums, per-agent sums, probabilities, decisions under the recorded noise, post-state - incl. the quarantine
    windows (households read raw transmissions) - in several tile geometries;
  * run form == plain tiled arrays on a 400 k-agent benchmark world: venue sums bit for bit (both add fixed-point
    terms), probabilities to rounding, Philox decisions equal except at ties; a world where EVERY household edge is
    primary (one household per person, the reference's kind of world: the set's tiled arrays are empty);
  * the backward pass through a run-form set == the plain one.
"""
import numpy as np
import pytest
import torch

import bench as B
import gj_testlib as L
from grad_june_amd import tiling as TL
from grad_june_amd.plan import _host

pytestmark = pytest.mark.gpu


def household_major(agent, venue, A):
    first = np.full(A, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first, agent, venue)
    order = np.argsort(first, kind="stable")
    new_of = np.empty(A, dtype=np.int64)
    new_of[order] = np.arange(A)
    return order, new_of


@pytest.mark.parametrize("A,V,mem,sa,halo,svmax,eb", [(5000, 1800, 2, 512, 0, 300, 2000), (5000, 2500, 1, 512, 0, 8192, 32768),
                                                     (3000, 900, 3, 256, 700, 64, 500), (40000, 30000, 2, 1024, 0, 4096, 8192),
                                                     (700, 5000, 1, 64, 0, 16, 64)])
def test_native_run_form_equals_the_numpy_specification(device, A, V, mem, sa, halo, svmax, eb):
    from grad_june_amd.tiling_native import build_tiled_native, finish_run_form_native, split_primary_runs_native

    rng = np.random.default_rng(A + V + mem)
    agent = np.repeat(np.arange(A), mem)
    venue = rng.integers(0, V, len(agent))
    order, new_of = household_major(agent, venue, A)
    perm = rng.permutation(len(agent))
    agent, venue = new_of[agent][perm], venue[perm]
    n_own = -(-A // sa)
    if halo:
        ha = n_own * sa + rng.integers(0, halo, 2 * halo)
        agent, venue = np.concatenate([agent, ha]), np.concatenate([venue, rng.integers(0, V, 2 * halo)])
        perm = rng.permutation(len(agent))
        agent, venue = agent[perm], venue[perm]
    n_ext = n_own * sa + halo if halo else A
    S = -(-n_ext // sa)
    pc = rng.random(V).astype(np.float32)
    ref = TL.split_primary_runs(agent, venue, A, V, sa)
    got, rest = split_primary_runs_native(torch.from_numpy(agent).to(device), torch.from_numpy(venue).to(device), A, V, sa,
                                          device)
    assert ref is not None and got is not None and got.n_primary == ref.n_primary == A
    assert np.array_equal(_host(got.vmin), ref.vmin) and got.max_window == ref.max_window
    assert np.array_equal(_host(got.win_lo), ref.win_lo) and np.array_equal(_host(got.win_n), ref.win_n)
    assert np.array_equal(rest["agent"].cpu().numpy(), agent[ref.keep]) and np.array_equal(rest["venue"].cpu().numpy(), venue[ref.keep])
    t_ref = TL.build_tiled("hh", agent[ref.keep], venue[ref.keep], V, pc, S, sa, sv_max=svmax, eb_target=eb)
    t_got = build_tiled_native("hh", rest["agent"], rest["venue"], V, pc, S, sa, sv_max=svmax, eb_target=eb, device=device,
                               n_ext_agents=n_ext)
    assert np.array_equal(_host(t_got.blk_v0), t_ref.blk_v0) and t_got.n_edges == t_ref.n_edges
    ref = TL.finish_run_form(ref, t_ref.blk_v0, A, sa)
    got = finish_run_form_native(got, t_got.blk_v0, A, sa, device)
    assert np.array_equal(_host(got.blk_r0), ref.blk_r0)
    assert np.array_equal(_host(got.pv_blk).view(np.uint16), ref.pv_blk)
    assert np.array_equal(_host(got.pv_win).view(np.uint16), ref.pv_win)
    # not in that order: no run form, from either build
    shuffled = rng.permutation(n_ext)[agent]
    assert TL.split_primary_runs(shuffled, venue, A, V, sa) is None
    assert split_primary_runs_native(torch.from_numpy(shuffled).to(device), torch.from_numpy(venue).to(device), A, V, sa,
                                     device)[0] is None


GEOMETRIES = [dict(), dict(sv_max=64, eb_target=512, slices="small"), dict(sv_max=16, eb_target=64, slices="small", desc_wide=True),
              dict(split_epilogue=True), dict(slices="small", device_compile=True)]
GEOMETRY_IDS = ["default", "small-tiles", "tiny-tiles-wide-desc", "split-epilogue", "small-slices-device-compile"]


@pytest.mark.parametrize("geometry", GEOMETRIES, ids=GEOMETRY_IDS)
@pytest.mark.parametrize("name", ["june769.npz", "june769_hot.npz"])
def test_reference_trajectories_in_household_major_order(device, name, geometry):
    """Every recorded step of the reference's 769-agent world (11 networks, policies on) with the agents renumbered
    household-major and the household set in the run form; per-agent results are compared in the reference's order."""
    from grad_june_amd.engine import AgentBuffers, InfectionEngine
    from grad_june_amd.plan import DevicePlan, compile_plan

    npz = L.load_npz(name)
    world = L.world_from(npz)
    tables = L.tables_from(npz)
    A = world["n_agents"]
    hh = world["edge_sets"]["household"]
    order, new_of = household_major(hh["agent"].numpy(), hh["venue"].numpy(), A)
    es = {k: {"agent": new_of[v["agent"].numpy()], "venue": v["venue"].numpy(), "people": v["people"].numpy()}
          for k, v in world["edge_sets"].items()}
    kw = dict(geometry)
    split = kw.pop("split_epilogue", False)
    dev_compile = kw.pop("device_compile", False)
    if kw.get("slices") == "small":
        kw["slices"] = (-(-A // 64), 64)
    host = compile_plan(A, es, age=world["age"].numpy()[order], sex=world["sex"].numpy()[order], layout="tiled",
                        runs=("household",), device=device if dev_compile else None, **kw)
    t = {s.name: s.tiled for s in host.sets}["household"]
    assert t.runs is not None and t.runs.n_primary == int((np.bincount(hh["agent"].numpy(), minlength=A) > 0).sum())
    assert t.n_edges == len(hh["agent"]) - t.runs.n_primary          # (one household per person: 0 edges stay tiled)
    engine = InfectionEngine(DevicePlan(host, L.network_specs(world, tables), device, split_epilogue=split))
    to_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(torch.float32).to(device)
    n_checked = 0
    for i in range(int(npz["n_steps"])):
        rec = L.step_record(npz, f"step{i}/")
        sc = L.step_scalars(rec)
        has_q = sc["quarantine_thresholds"] is not None
        st = {k: to_dev(v.numpy()[order]) for k, v in L.pre_state(rec).items()}
        st["transmission"] = torch.zeros(A, device=device)
        p = engine.params(now=sc["now"], delta_time=sc["delta_time"], day_type=sc["day_type"], active=sc["active"],
                          betas=sc["betas"], has_quarantine=has_q, q_threshold=L.q_threshold(sc["quarantine_thresholds"]))
        bufs = AgentBuffers(engine.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"], rate=st["rate"],
                            shift=st["shift"], infection_time=st["infection_time"], is_infected=st["is_infected"],
                            susceptibility=st["susceptibility"], transmission=st["transmission"],
                            current_stage=st["current_stage"])
        noise = to_dev(rec["exp_noise"][:, order])
        probs, new, ts = (torch.empty(A, device=device) for _ in range(3))
        engine.step(bufs, p, engine.io(not_infected_probs=probs, new_infected=new, exp_noise=noise, trans_susc=ts))
        torch.cuda.synchronize()
        back = lambda x: x.cpu().numpy()[new_of]                      # reference order
        per_set = {}
        for net in sc["active"]:
            spec = engine.plan.networks[net]
            k = per_set.get(spec.edge_set, 0)
            per_set[spec.edge_set] = k + 1
            ref = rec["cum/" + net]
            got = engine.plan.cum_of(spec.edge_set)[:, k].cpu().numpy()
            assert np.allclose(got, ref, rtol=2e-5, atol=1e-12 + 1e-6 * float(np.abs(ref).max() if ref.size else 0)), (i, net)
        ts_ref = np.zeros(A, dtype=np.float32)
        for net in sc["active"]:
            ts_ref += rec["ts/" + net]
        assert np.allclose(back(ts), ts_ref, rtol=2e-5, atol=1e-7), i
        assert np.abs(back(probs) - rec["not_infected_probs"]).max() <= 1e-5, i
        assert np.array_equal(back(new) > 0.5, rec["new_infected"] > 0.5), i
        for k in ("susceptibility", "is_infected", "infection_time"):
            assert np.allclose(back(st[k]), rec["post/" + k], rtol=1e-6, atol=1e-6), (i, k)
        n_checked += 1
    assert n_checked >= 10


def _runner(world, device, runs, **kw):
    from grad_june_amd.benchrun import SingleGpuHotPath

    specs, betas = B.network_specs(world), B.betas_of(world)
    return SingleGpuHotPath(world, specs, betas, device, seed=5, runs=runs, **kw)


@pytest.mark.parametrize("one_household_each", [False, True], ids=["c3-400k", "one-household-per-person"])
@pytest.mark.parametrize("device_compile", [False, True], ids=["numpy-compile", "device-compile"])
def test_run_form_equals_the_plain_tiled_set(device, one_household_each, device_compile):
    from grad_june_amd.synthetic import make_world, reorder_agents

    world = make_world("c3", n_agents=400_000, seed=21, infected_fraction=0.1)
    if one_household_each:      # the reference's kind of world: every person lives in exactly one household
        hh = world["edge_sets"]["household"]
        first = np.unique(hh["agent"], return_index=True)[1]
        keep = np.zeros(len(hh["agent"]), dtype=bool)
        keep[first] = True
        hh = {"agent": hh["agent"][keep], "venue": hh["venue"][keep]}
        hh["people"] = np.bincount(hh["venue"], minlength=len(world["edge_sets"]["household"]["people"]))
        world["edge_sets"]["household"] = hh
    world = reorder_agents(world, by="household")
    A = world["n_agents"]
    plain = _runner(world, device, False, device_compile=device_compile)
    runs = _runner(world, device, ("household",) if one_household_each else None, device_compile=device_compile)
    t = {s.name: s.tiled for s in runs.engine.plan.host.sets}["household"]
    assert t.runs is not None and t.runs.n_primary == A            # (auto: 160 k households are too many for the direct form)
    assert (t.n_edges == 0) == one_household_each
    assert all(s.tiled.runs is None for s in runs.engine.plan.host.sets if s.name != "household")
    for r in (plain, runs):
        r.ts = torch.empty(A, device=device)
        r.io = r.engine.io(not_infected_probs=r.probs, new_infected=r.new_infected, trans_susc=r.ts)
    flips = 0
    for step in range(6):
        for k in ("is_infected", "susceptibility", "infection_time"):
            runs.state[k].copy_(plain.state[k])                     # teacher-forced: a flipped tie must not compound
        plain.step()
        runs.step()
        torch.cuda.synchronize()
        for hs in plain.engine.plan.host.sets:                      # pass 1: both forms add fixed-point terms - bitwise
            assert torch.equal(plain.engine.plan.cum_of(hs.name), runs.engine.plan.cum_of(hs.name)), (step, hs.name)
        assert torch.allclose(plain.ts, runs.ts, rtol=2e-6, atol=1e-9), step
        assert float((plain.probs - runs.probs).abs().max()) <= 1e-6, step
        differ = plain.new_infected != runs.new_infected
        flips += int(differ.sum())
        assert int(differ.sum()) <= 3, step
    assert float(plain.state["is_infected"].sum()) > 1.5 * 0.1 * A
    print("decisions that differ over 6 teacher-forced steps:", flips)


def test_backward_through_a_run_form_set(device):
    """d cases / d log_beta through three differentiable steps: run form == plain tiled arrays (the transposed passes
    read the cotangents of the primary edges from the per-agent array, like the forward reads the transmissions)."""
    from types import SimpleNamespace

    from grad_june_amd.autograd import HotPathStep
    from grad_june_amd.synthetic import make_world, reorder_agents

    world = reorder_agents(make_world("c3", n_agents=300_000, seed=8, infected_fraction=0.05), by="household")
    networks, betas = world["networks"], B.betas_of(world)
    grads = {}
    for label, runs in (("plain", False), ("runs", None)):
        r = _runner(world, device, runs, device_compile=True)
        has_runs = any(s.tiled.runs is not None for s in r.engine.plan.host.sets)
        assert has_runs == (label == "runs")
        logb = {n: torch.nn.Parameter(torch.tensor(B.DEFAULT_LOG_BETA[n], device=device)) for n in networks}
        nets = [SimpleNamespace(name=n, log_beta=logb[n]) for n in networks]
        fixed = {k: r.state[k] for k in ("max_infectiousness", "shape", "rate", "shift")}
        s, i, t = (r.state[k].clone() for k in ("susceptibility", "is_infected", "infection_time"))
        for k in range(3):
            params = r.engine.params(now=1.0 + k, delta_time=1.0, day_type=0, active=networks, betas=betas, seed=5, step=k)
            env = {"engine": r.engine, "params": params, "fixed": fixed, "stage": None, "exp_noise": None, "nets": nets,
                   "betas": betas}
            s, i, t, _ = HotPathStep.apply(env, s, i, t, *[n.log_beta for n in nets])
        i.sum().backward()
        grads[label] = np.array([float(v.grad) for v in logb.values()])
    assert np.abs(grads["plain"]).min() > 0
    assert np.allclose(grads["runs"], grads["plain"], rtol=2e-4), (grads["runs"], grads["plain"])


@pytest.mark.parametrize("device_compile", [False, True], ids=["numpy-compile", "device-compile"])
def test_run_form_on_the_ranks_of_a_partition(device, device_compile):
    """What `bench.py --gpus N` builds: the household-major world cut into ranks.  On a rank the household set runs in
    halo mode - owned agents' primary edges in the run form, the edges of halo agents (indices behind the owned slices)
    in the tiled arrays - and the partition reproduces the single-GPU run's discrete state exactly."""
    from grad_june_amd.benchrun import SingleGpuHotPath
    from grad_june_amd.distributed import PartitionedHotPath
    from grad_june_amd.synthetic import make_world, reorder_agents

    world = reorder_agents(make_world("c3", n_agents=400_000, seed=31, infected_fraction=0.08), by="household")
    specs, betas = B.network_specs(world), B.betas_of(world)
    single = SingleGpuHotPath(world, specs, betas, device, seed=5, device_compile=device_compile)
    parted = PartitionedHotPath(world, specs, betas, device, parts=4, seed=5, device_compile=device_compile)
    for rk in parted.ranks:
        t = {s.name: s.tiled for s in rk.engine.plan.host.sets}["household"]
        assert rk.rw.modes["household"] == "halo" and rk.rw.n_halo > 1000
        assert t.runs is not None and t.runs.n_primary == rk.rw.n_local and t.n_edges > 0
    for _ in range(4):
        single.step()
        parted.step()
    torch.cuda.synchronize()
    for k, v in parted.state.items():
        assert torch.equal(v, single.state[k]), k
    assert float(single.state["is_infected"].sum()) > 1.3 * 0.08 * world["n_agents"]
