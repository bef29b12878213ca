"""GPU: BASELINE.json's full-size synthetic worlds, checked through size-independent properties
(the oracle would need minutes per step at these sizes):

  * tiled and CSR layouts agree (probabilities <= 2e-6, decisions equal away from Gumbel ties)
  * pass 1 is linear: doubling every transmission doubles every venue sum bit for bit
  * mass conservation: sum_v (sum of venue v) == sum_a degree(a) * transmission[a]   (fp64 check sums)
  * a random sample of venues / agents against fp64 sums taken straight from the COO edge list
  * run-to-run determinism: bitwise identical outputs

C2 (1 M agents, 15 M edges) always runs; C3 (10 M agents, 120 M network-edges) when GJ_FULL_C3=1.
"""
import os

import numpy as np
import pytest
import torch

import bench as B
from grad_june_amd.benchrun import SingleGpuHotPath
from grad_june_amd.synthetic import edge_set_of, make_world

pytestmark = pytest.mark.gpu

# c5 = power-law venue degrees (Zipf alpha 2, venues up to 50 000 attendees), scaled to one test box
CASES = ([("c2", None)] + ([("c3", None)] if os.environ.get("GJ_FULL_C3") == "1" else [("c3", 2_000_000)])
         + [("c5", 1_000_000)])


def run_stages(r, sample=False):
    p = r.params()
    for ph in (0, 1, 2, 3 if sample else 4) if r.layout == "tiled" else (0, 1, 3 if sample else 4):
        r.engine.step_phase(r.bufs, p, r.io, ph)
    torch.cuda.synchronize()


@pytest.mark.parametrize("preset,agents", CASES, ids=[f"{p}-{a or 'full'}" for p, a in CASES])
def test_fullsize_properties(device, preset, agents):
    world = make_world(preset, n_agents=agents, seed=1234, infected_fraction=0.03)
    A = world["n_agents"]
    specs, betas = B.network_specs(world), B.betas_of(world)
    noise = torch.empty(2, A).exponential_(generator=torch.Generator().manual_seed(1)).to(device)
    tiled = SingleGpuHotPath(world, specs, betas, device, seed=3, layout="tiled", exp_noise=noise)
    csr = SingleGpuHotPath(world, specs, betas, device, seed=3, layout="csr", exp_noise=noise)

    # -- layouts agree (probabilities), first without touching the state ----------------------------
    run_stages(tiled)
    run_stages(csr)
    pt, pc = tiled.probs.cpu().numpy(), csr.probs.cpu().numpy()
    assert np.abs(pt - pc).max() <= 2e-6
    assert (pt < 1.0).sum() > 0.5 * A            # the world is actually exposed

    # -- sampled venues / agents against fp64 sums from the COO edge list ---------------------------
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
        t2 = SingleGpuHotPath(world, specs, betas, device, seed=3, layout="tiled", exp_noise=noise)
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


def test_c2_full_size_against_the_oracle(device):
    """BASELINE.json configs[1] at full size (1 M agents, household/school/company, 15 M edges): one
    whole step on the GPU against the CPU oracle on identical injected noise - per-agent probabilities
    within 1e-5, infection decisions identical away from Gumbel ties, equal infection counts."""
    import gj_oracle as O

    world = make_world("c2", seed=1234, infected_fraction=0.02)
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
    """The contact graph compiled on the GPU (tiling_device, row f4) equals the numpy-compiled one array for
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
        for k in ("blk_v0", "blk_e0", "e_cls", "tile_sptr", "tile_jpos", "chunk_ptr", "chunk_desc"):
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
