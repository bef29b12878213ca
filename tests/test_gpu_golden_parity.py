"""GPU parity: the HIP path (through the C ABI) against the golden vectors recorded from the
reference, stage by stage and as the fused production step.

Tolerances (north_star: per-agent infection probabilities within 1e-5 fp32, equal infection
counts on identical noise):
  transmission           rtol 2e-5  (device lgammaf/powf/expf vs the host's libm)
  cum_n[v], ts           rtol 2e-5  (+ summation order for venues summed by more than one lane)
  not_infected_probs     atol 1e-5
  new_infected           EXACTLY the recorded decisions when fed the recorded probabilities;
                         in the fused step a decision may differ only where the oracle's Gumbel
                         margin |z1-z0| is below 1e-3 (none occurs in the fixtures)
"""
import numpy as np
import pytest
import torch

import gj_testlib as L

pytestmark = pytest.mark.gpu

RTOL = 2e-5

# every parity case runs on both device layouts: "csr" (deterministic CSR kernels) and "tiled"
# (LDS-tiled fast path; small tiles/blocks forced so that multi-slice / multi-block paths run)
# and with pass 2 in its two forms: through the per-edge workspace (phases C + D, direct=False) and "direct" (venue
# values staged through LDS, tiling.build_ell; the golden worlds are small enough for every set to qualify - a tiny
# table forces several venue groups per set)
LAYOUTS = [("csr", {}), ("tiled", dict(direct=False)),
           ("tiled", dict(sv_max=64, eb_target=512, slices="small", desc_wide=False, direct=False)),
           ("tiled", dict(sv_max=64, eb_target=512, slices="small", desc_wide=True, direct=False)),
           ("tiled", dict(sv_max=16, eb_target=64, slices="small", desc_wide=True, direct=False)),
           ("tiled", dict(split_epilogue=True, direct=False)),
           ("tiled", {}), ("tiled", dict(sv_max=64, eb_target=512, slices="small", direct_table_floats=24)),
           ("tiled", dict(split_epilogue=True, direct_table_floats=40)),
           ("tiled", dict(slices="small", direct=("school", "leisure", "household"))),
           ("tiled", dict(sv_max=16, eb_target=64, slices="small", desc_explicit=True, direct=False)),
           ("tiled", dict(sv_max=64, eb_target=512, slices="small", desc_explicit=True)),
           ("tiled", dict(direct=False, tile_pad=16)), ("tiled", dict(sv_max=256, eb_target=4096, tile_pad=16)),
           ("tiled", dict(presum=True)),
           ("tiled", dict(sv_max=64, eb_target=512, slices="small", direct_table_floats=24, presum=True))]
LAYOUT_IDS = ["csr", "tiled", "tiled-small-tiles", "tiled-small-tiles-wide-desc", "tiled-tiny-tiles-wide-desc",
              "tiled-split-epilogue", "tiled-direct", "tiled-direct-small-tiles-venue-groups",
              "tiled-direct-split-epilogue-venue-groups", "tiled-direct-some-sets", "tiled-tiny-tiles-explicit-slots",
              "tiled-direct-small-tiles-explicit-slots", "tiled-padded-to-sectors", "tiled-direct-blocks-padded-to-sectors", "tiled-direct-pass1-direct",
              "tiled-direct-small-tiles-venue-groups-pass1-direct"]


def engine_for(world, tables, device, layout):
    name, kw = layout
    kw = dict(kw)
    if kw.get("slices") == "small":
        sa = 64
        kw["slices"] = (-(-world["n_agents"] // sa), sa)
    if isinstance(kw.get("direct"), tuple):      # name only the sets this world has and that qualify
        from grad_june_amd.plan import compile_plan

        es = {k: {kk: vv.numpy() for kk, vv in v.items()} for k, v in world["edge_sets"].items()}
        auto = compile_plan(world["n_agents"], es, age=world["age"].numpy(), sex=world["sex"].numpy(), layout="tiled",
                            slices=kw.get("slices"))
        kw["direct"] = tuple(s.name for s in auto.sets if s.tiled.ell_k and s.name in kw["direct"])
    return L.make_engine(world, tables, device, layout=name, **kw)


def _close(a, b, rtol=RTOL, atol=1e-9, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    ok = np.allclose(a, b, rtol=rtol, atol=atol)
    assert ok, f"{what}: max abs diff {np.abs(a - b).max():.3e}, max rel {np.max(np.abs(a - b) / (np.abs(b) + 1e-30)):.3e}"


def _margins(p, noise):
    p = torch.from_numpy(p)
    z0 = (p.log() - torch.from_numpy(noise[0]).log()) / 0.1
    z1 = ((1 - p).log() - torch.from_numpy(noise[1]).log()) / 0.1
    return (z1 - z0).abs().numpy()


def run_case(npz, prefix, engine, device, tables):
    from grad_june_amd.engine import AgentBuffers

    rec = L.step_record(npz, prefix)
    sc = L.step_scalars(rec)
    st = L.device_state(L.pre_state(rec), device)
    A = engine.plan.host.n_agents
    has_q = sc["quarantine_thresholds"] is not None
    p = engine.params(now=sc["now"], delta_time=sc["delta_time"], day_type=sc["day_type"],
                      active=sc["active"], betas=sc["betas"], has_quarantine=has_q,
                      q_threshold=L.q_threshold(sc["quarantine_thresholds"]))

    def buffers(s):
        return AgentBuffers(engine.plan, max_infectiousness=s["max_infectiousness"], shape=s["shape"],
                            rate=s["rate"], shift=s["shift"], infection_time=s["infection_time"],
                            is_infected=s["is_infected"], susceptibility=s["susceptibility"],
                            transmission=s["transmission"], current_stage=s["current_stage"])

    # ---- stage by stage -------------------------------------------------------------------
    bufs = buffers(st)
    engine.transmission_update(bufs, p)
    _close(st["transmission"], rec["transmission"], what=prefix + "transmission (a1)")
    if has_q and "qmask" in rec:
        _close(bufs.tensors["q_transmission"], rec["qmask"] * rec["transmission"], what=prefix + "q*transmission (a2)")

    engine.venue_reduce(bufs, p)
    per_set = {}
    for name in sc["active"]:
        spec = engine.plan.networks[name]
        k = per_set.get(spec.edge_set, 0)
        per_set[spec.edge_set] = k + 1
        cum = engine.plan.cum_of(spec.edge_set)[:, k]
        ref = rec["cum/" + name]
        _close(cum, ref, atol=1e-12 + 1e-6 * float(np.abs(ref).max() if ref.size else 0), what=f"{prefix}cum/{name} (a5)")

    probs = torch.empty(A, device=device)
    ts = torch.empty(A, device=device)
    if engine.plan.c.tiled:       # gj_step_io.agent_sums: the sums before the susceptibility factor (kept by a backward pass)
        sums = torch.full((A,), float("nan"), device=device)
        engine.agent_gather(bufs, p, engine.io(not_infected_probs=probs, trans_susc=ts, agent_sums=sums), sample=False)
        assert torch.equal(ts, st["susceptibility"] * sums), prefix + "trans_susc == susceptibility * agent_sums"
    else:                         # the CSR kernels multiply per term, as the reference does: no such output
        with pytest.raises(Exception, match="plan"):
            engine.agent_gather(bufs, p, engine.io(trans_susc=ts, agent_sums=torch.empty(A, device=device)), sample=False)
        engine.agent_gather(bufs, p, engine.io(not_infected_probs=probs, trans_susc=ts), sample=False)
    ts_ref = np.zeros(A, dtype=np.float32)
    for name in sc["active"]:
        ts_ref += rec["ts/" + name]
    _close(ts, ts_ref, atol=1e-7, what=prefix + "sum_n ts_n (a6)")
    assert np.abs(probs.cpu().numpy() - rec["not_infected_probs"]).max() <= 1e-5, prefix + "not_infected_probs (a7)"

    # ---- a8 + a9 teacher-forced on the recorded probabilities: decisions must be exact -------
    s2 = {k: v.clone() for k, v in st.items()}
    new_inf = torch.empty(A, device=device)
    noise = torch.from_numpy(rec["exp_noise"]).to(device).contiguous()
    engine.sample_infect(torch.from_numpy(rec["not_infected_probs"]).to(device), now=sc["now"],
                         susceptibility=s2["susceptibility"], is_infected=s2["is_infected"],
                         infection_time=s2["infection_time"], exp_noise=noise, new_infected=new_inf)
    got = new_inf.cpu().numpy()
    assert np.array_equal(got > 0.5, rec["new_infected"] > 0.5), prefix + "decisions (a8)"
    assert np.abs(got - rec["new_infected"]).max() <= 1e-6
    for k in ("susceptibility", "is_infected", "infection_time"):
        _close(s2[k], rec["post/" + k], rtol=1e-6, atol=1e-6, what=prefix + "post/" + k + " (a9)")

    # ---- the fused production step from the recorded pre-state -----------------------------
    s3 = L.device_state(L.pre_state(rec), device)
    new3 = torch.empty(A, device=device)
    p3 = torch.empty(A, device=device)
    engine.step(buffers(s3), p, engine.io(not_infected_probs=p3, new_infected=new3, exp_noise=noise))
    torch.cuda.synchronize()
    assert np.abs(p3.cpu().numpy() - rec["not_infected_probs"]).max() <= 1e-5
    dec = new3.cpu().numpy() > 0.5
    bad = dec != (rec["new_infected"] > 0.5)
    if bad.any():
        m = _margins(rec["not_infected_probs"], rec["exp_noise"])
        assert (m[bad] < 1e-3).all(), prefix + "fused decision differs away from a tie"
    else:
        assert dec.sum() == (rec["new_infected"] > 0.5).sum()
        for k in ("susceptibility", "is_infected", "infection_time"):
            _close(s3[k], rec["post/" + k], rtol=1e-6, atol=1e-6, what=prefix + "fused post/" + k)
    return int(bad.sum())


@pytest.mark.parametrize("layout", LAYOUTS, ids=LAYOUT_IDS)
def test_kat6(device, layout):
    """The reference's exact known-answer test (test_base.py:39-44) through the HIP path."""
    from grad_june_amd.engine import AgentBuffers

    npz = L.load_npz("kat6.npz")
    world = L.world_from(npz)
    eng = engine_for(world, None, device, layout)
    A = 6
    z = lambda: torch.zeros(A, device=device)
    trans = torch.from_numpy(npz["transmission"]).to(device)
    susc = torch.from_numpy(npz["susceptibility"]).to(device)
    bufs = AgentBuffers(eng.plan, infection_time=z(), is_infected=z(), susceptibility=susc, transmission=trans)
    p = eng.params(now=0.0, delta_time=float(npz["dt"]), day_type=0, active=["school"],
                   betas={"school": float(npz["beta/school"])})
    probs = torch.empty(A, device=device)
    eng.venue_reduce(bufs, p)
    eng.agent_gather(bufs, p, eng.io(not_infected_probs=probs), sample=False)
    expected = np.exp(-npz["expected_exponent"])
    assert np.allclose(probs.cpu().numpy(), expected)
    assert np.abs(probs.cpu().numpy() - npz["not_infected_probs"]).max() <= 1e-6


@pytest.mark.parametrize("layout", LAYOUTS, ids=LAYOUT_IDS)
def test_c100_policy_variants(device, layout):
    npz = L.load_npz("c100.npz")
    world = L.world_from(npz)
    eng = engine_for(world, None, device, layout)
    flips = 0
    for v in str(npz["variants"]).split(","):
        flips += run_case(npz, v + "/", eng, device, None)
    assert flips == 0


@pytest.mark.parametrize("layout", [l for l in LAYOUTS if l[0] == "tiled"][:8],
                         ids=[i for l, i in zip(LAYOUTS, LAYOUT_IDS) if l[0] == "tiled"][:8])
def test_agent_sums_do_not_depend_on_the_susceptibility(device, layout):
    """gj_step_io.agent_sums is the per-agent sum BEFORE the susceptibility factor: the same bits whatever the
    susceptibilities are (fractional, zero), and trans_susc is exactly susceptibility * agent_sums - in the fused step
    as in the stand-alone pass 2.  (What the differentiable step keeps for its backward, autograd.py.)"""
    from grad_june_amd.engine import AgentBuffers

    npz = L.load_npz("june769_hot.npz")
    world, tables = L.world_from(npz), L.tables_from(npz)
    eng = engine_for(world, tables, device, layout)
    rec = L.step_record(npz, "step2/")
    sc = L.step_scalars(rec)
    A = eng.plan.host.n_agents
    has_q = sc["quarantine_thresholds"] is not None
    p = eng.params(now=sc["now"], delta_time=sc["delta_time"], day_type=sc["day_type"], active=sc["active"],
                   betas=sc["betas"], has_quarantine=has_q, q_threshold=L.q_threshold(sc["quarantine_thresholds"]))
    gen = torch.Generator().manual_seed(5)
    out = []
    for susc in (None, torch.rand(A, generator=gen), torch.zeros(A)):
        st = L.device_state(L.pre_state(rec), device)
        if susc is not None:
            st["susceptibility"] = susc.to(device)
        s0 = st["susceptibility"].clone()
        bufs = AgentBuffers(eng.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"], rate=st["rate"],
                            shift=st["shift"], infection_time=st["infection_time"], is_infected=st["is_infected"],
                            susceptibility=st["susceptibility"], transmission=st["transmission"],
                            current_stage=st["current_stage"])
        sums, ts = torch.empty(A, device=device), torch.empty(A, device=device)
        noise = torch.from_numpy(rec["exp_noise"]).to(device).contiguous()
        eng.step(bufs, p, eng.io(trans_susc=ts, agent_sums=sums, exp_noise=noise, new_infected=torch.empty(A, device=device)))
        assert torch.equal(ts, s0 * sums)
        out.append(sums)
    assert float(out[0].abs().max()) > 0
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])


@pytest.mark.parametrize("layout", LAYOUTS, ids=LAYOUT_IDS)
@pytest.mark.parametrize("name", ["june769.npz", "june769_hot.npz", "synth10k.npz"])
def test_trajectory_teacher_forced(device, name, layout):
    """Every recorded step of the reference trajectories, each from its recorded pre-state."""
    npz = L.load_npz(name)
    world = L.world_from(npz)
    tables = L.tables_from(npz)
    eng = engine_for(world, tables, device, layout)
    flips = 0
    for i in range(int(npz["n_steps"])):
        flips += run_case(npz, f"step{i}/", eng, device, tables)
    assert flips == 0


@pytest.mark.parametrize("layout", LAYOUTS, ids=LAYOUT_IDS)
@pytest.mark.parametrize("name", ["june769.npz", "june769_hot.npz"])
def test_trajectory_chained_counts(device, name, layout):
    """15 chained hot-path steps on the GPU (own state carried forward; the recorded symptom
    stage supplies the quarantine mask, symptoms being outside the path): infection counts per
    timestep must equal the reference's cases_per_timestep under the same injected noise."""
    from grad_june_amd.engine import AgentBuffers

    npz = L.load_npz(name)
    world = L.world_from(npz)
    tables = L.tables_from(npz)
    eng = engine_for(world, tables, device, layout)
    A = world["n_agents"]
    rec0 = L.step_record(npz, "step0/")
    st = L.device_state(L.pre_state(rec0), device)
    cases = [float(st["is_infected"].sum().item())]
    for i in range(int(npz["n_steps"])):
        rec = L.step_record(npz, f"step{i}/")
        sc = L.step_scalars(rec)
        st["current_stage"] = torch.from_numpy(rec["pre/current_stage"]).to(torch.float32).to(device)
        has_q = sc["quarantine_thresholds"] is not None
        p = eng.params(now=sc["now"], delta_time=sc["delta_time"], day_type=sc["day_type"], active=sc["active"],
                       betas=sc["betas"], has_quarantine=has_q, q_threshold=L.q_threshold(sc["quarantine_thresholds"]))
        bufs = AgentBuffers(eng.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"], rate=st["rate"],
                            shift=st["shift"], infection_time=st["infection_time"], is_infected=st["is_infected"],
                            susceptibility=st["susceptibility"], transmission=st["transmission"],
                            current_stage=st["current_stage"])
        noise = torch.from_numpy(rec["exp_noise"]).to(device).contiguous()
        eng.step(bufs, p, eng.io(exp_noise=noise))
        cases.append(float(st["is_infected"].sum().item()))
        assert np.array_equal(st["is_infected"].cpu().numpy(), rec["post/is_infected"]), f"step {i}"
    assert np.allclose(cases, npz["cases_per_timestep"])


@pytest.mark.parametrize("layout", LAYOUTS, ids=LAYOUT_IDS)
def test_no_active_network(device, layout):
    """Every venue closed (close_venue_policies.py:11-22 can empty the list): trans_susc stays 0, the
    reference's floor clamp(…, 1e-6) sets p = exp(-1e-6 * dt) for everyone (base.py:136-140)."""
    from grad_june_amd.engine import AgentBuffers

    npz = L.load_npz("c100.npz")
    world = L.world_from(npz)
    eng = engine_for(world, None, device, layout)
    rec = L.step_record(npz, "plain_t3/")
    st = L.device_state(L.pre_state(rec), device)
    p = eng.params(now=3.0, delta_time=1.0, day_type=0, active=[], betas={})
    bufs = AgentBuffers(eng.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"], rate=st["rate"],
                        shift=st["shift"], infection_time=st["infection_time"], is_infected=st["is_infected"],
                        susceptibility=st["susceptibility"], transmission=st["transmission"])
    probs, new = torch.empty(100, device=device), torch.empty(100, device=device)
    before = st["is_infected"].clone()
    noise = torch.from_numpy(rec["exp_noise"]).to(device).contiguous()
    eng.step(bufs, p, eng.io(not_infected_probs=probs, new_infected=new, exp_noise=noise))
    torch.cuda.synchronize()
    assert torch.allclose(probs, torch.full_like(probs, float(np.exp(np.float32(-1e-6)))))
    expect = O_sample(probs.cpu(), noise.cpu())
    assert torch.equal(new.cpu() > 0.5, expect > 0.5)
    assert torch.equal(st["is_infected"], before + new)


@pytest.mark.parametrize("poison", ["nan", "inf", "huge"])
@pytest.mark.parametrize("layout", LAYOUTS, ids=LAYOUT_IDS)
def test_non_finite_transmissions_propagate(device, layout, poison):
    """A NaN infectiousness poisons, a huge finite one SATURATES - as in the reference, whose scatter_add and torch.clamp
    (base.py:78-83, 136-140) carry a NaN to every co-attendee of the poisoned agent's venues and clamp a huge finite sum
    to 100, i.e. p = exp(-100 dt): infected with certainty, while an already infected co-attendee (susceptibility 0:
    0 * 3e30 == 0) keeps the clean run's probability.  The tiled layout sums in fixed point and flags what lies beyond
    its window (gj_tiled.h fx_flag: the element reads back +-1e30 or NaN); both epilogues clamp NaN-preservingly.
    Everyone else gets exactly the clean run's probability."""
    import gj_oracle as O
    from grad_june_amd.engine import AgentBuffers

    npz = L.load_npz("c100.npz")
    world = L.world_from(npz)
    rec = L.step_record(npz, "plain_t3/")
    sc = L.step_scalars(rec)
    pre = L.pre_state(rec)
    who = int(torch.nonzero(pre["is_infected"] > 0)[0])
    bad = {"nan": float("nan"), "inf": float("inf"), "huge": 3e30}[poison]
    dirty = {k: v.clone() for k, v in pre.items()}
    dirty["max_infectiousness"][who] = bad
    noise = torch.from_numpy(rec["exp_noise"])
    ref_clean = O.hot_path_step(world, pre, exp_noise=noise, **sc)
    ref = O.hot_path_step(world, dirty, exp_noise=noise, **sc)
    hit = ~torch.isfinite(ref["not_infected_probs"]) | (ref["not_infected_probs"] != ref_clean["not_infected_probs"])
    assert 2 <= int(hit.sum()) < 100                      # the agent's co-attendees, not the whole world
    eng = engine_for(world, None, device, layout)
    st = L.device_state(dirty, device)
    p = eng.params(now=sc["now"], delta_time=sc["delta_time"], day_type=sc["day_type"], active=sc["active"],
                   betas=sc["betas"])
    bufs = AgentBuffers(eng.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"], rate=st["rate"],
                        shift=st["shift"], infection_time=st["infection_time"], is_infected=st["is_infected"],
                        susceptibility=st["susceptibility"], transmission=st["transmission"])
    probs = torch.empty(100, device=device)
    eng.step(bufs, p, eng.io(not_infected_probs=probs, new_infected=torch.empty(100, device=device),
                             exp_noise=noise.to(device).contiguous()))
    torch.cuda.synchronize()
    got = probs.cpu()
    # everyone who shares a venue with the poisoned agent
    touched = torch.zeros(100, dtype=torch.bool)
    for es in world["edge_sets"].values():
        mine = es["venue"][es["agent"] == who]
        touched[es["agent"][torch.isin(es["venue"], mine)]] = True
    assert bool(hit[~touched].sum() == 0) and 2 <= int(touched.sum()) < 100
    clean = ~touched
    assert np.abs(got[clean].numpy() - ref_clean["not_infected_probs"][clean].numpy()).max() <= 1e-5
    want_nan = torch.isnan(ref["not_infected_probs"])
    rest = touched & ~hit
    r = got[rest]
    if poison == "nan":
        assert bool(want_nan[hit].all()) and bool(torch.isnan(got[hit]).all())
    elif poison == "huge" and not layout[1].get("presum"):
        # finite and beyond every window: THE REFERENCE'S VALUES - exp(-100) for the susceptible co-attendees, who are all
        # infected under the recorded noise, and the clean probability for the already infected ones
        assert not bool(torch.isnan(got).any()) and not bool(want_nan.any())
        assert np.abs(got.numpy() - ref["not_infected_probs"].numpy()).max() <= 1e-5
        assert bool((got[hit] <= 1e-30).all()) and bool((ref["new_infected"][hit] > 0.5).all())
        assert torch.equal(st["is_infected"].cpu(), ref["is_infected"])
        assert bool(((r - ref_clean["not_infected_probs"][rest]).abs() <= 1e-5).all())
        return
    else:
        # +inf: the reference saturates where the sum stays +inf (ts -> 100, p = exp(-100)) and gives NaN where inf meets a
        # zero factor (susceptibility 0); the fixed-point path treats +inf as saturation throughout (exp(-100) / the clean
        # probability).  (The opt-in pass-1-direct experiment keeps round 3's rule: NaN for whatever it cannot sum.)
        g = got[hit]
        zero_factor = (g - ref_clean["not_infected_probs"][hit]).abs() <= 1e-5       # susceptibility 0: the clean probability
        assert bool((torch.isnan(g) | (g <= 1e-30) | (want_nan[hit] & zero_factor)).all())
        assert bool((g <= 1e-30)[~want_nan[hit]].all()) or layout[1].get("presum")
    # co-attendees the reference leaves untouched (susceptibility 0) read the clean probability - or NaN where a true
    # infinity met the zero
    assert bool((torch.isnan(r) | ((r - ref_clean["not_infected_probs"][rest]).abs() <= 1e-5)).all())


SAT_LAYOUTS = [LAYOUTS[i] for i in (0, 1, 2, 6, 7, 10)]
SAT_IDS = [LAYOUT_IDS[i] for i in (0, 1, 2, 6, 7, 10)]


@pytest.mark.parametrize("layout", SAT_LAYOUTS, ids=SAT_IDS)
def test_saturation_on_the_reference_world_with_leisure_and_quarantine(device, layout):
    """The reference's own 769-agent world - eleven networks, leisure tables (class weights of 0 for some ages), an active
    quarantine policy in the recorded step - with a huge finite infectiousness (3e30) on three infected agents: every
    co-attendee's probability equals the ORACLE's on the same state (no NaN anywhere): exp(-100) where the reference's
    finite sums are clamped, the clean value where a zero factor (susceptibility 0, a quarantined agent's mask, a leisure
    weight of 0) removes the term - which is why a saturated sum reads back a finite 1e30 and not an infinity."""
    import gj_oracle as O
    from grad_june_amd.engine import AgentBuffers

    npz = L.load_npz("june769_hot.npz")
    world, tables = L.world_from(npz), L.tables_from(npz)
    A = world["n_agents"]
    step = next(i for i in range(int(npz["n_steps"])) if int(npz[f"step{i}/has_quarantine"])
                and not np.isnan(npz[f"step{i}/q_thresholds"]).all())
    rec = L.step_record(npz, f"step{step}/")
    sc = L.step_scalars(rec)
    assert sc["quarantine_thresholds"] is not None and {"pub", "household"} <= set(sc["active"])
    pre = L.pre_state(rec)
    dirty = {k: v.clone() for k, v in pre.items()}
    who = torch.nonzero(pre["is_infected"] > 0).reshape(-1)[:3]
    assert len(who) == 3
    dirty["max_infectiousness"][who] = 3e30
    noise = torch.from_numpy(rec["exp_noise"])
    ref = O.hot_path_step(world, dirty, exp_noise=noise, leisure_tables=tables, **sc)
    clean = O.hot_path_step(world, pre, exp_noise=noise, leisure_tables=tables, **sc)
    assert not bool(torch.isnan(ref["not_infected_probs"]).any())
    hit = ref["not_infected_probs"] != clean["not_infected_probs"]
    assert int(hit.sum()) >= 10 and bool((ref["not_infected_probs"][hit] <= 1e-30).any())
    eng = engine_for(world, tables, device, layout)
    st = L.device_state(dirty, device)
    has_q = True
    p = eng.params(now=sc["now"], delta_time=sc["delta_time"], day_type=sc["day_type"], active=sc["active"],
                   betas=sc["betas"], has_quarantine=has_q, q_threshold=L.q_threshold(sc["quarantine_thresholds"]))
    bufs = AgentBuffers(eng.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"], rate=st["rate"],
                        shift=st["shift"], infection_time=st["infection_time"], is_infected=st["is_infected"],
                        susceptibility=st["susceptibility"], transmission=st["transmission"],
                        current_stage=st["current_stage"])
    probs, new = torch.empty(A, device=device), torch.empty(A, device=device)
    eng.step(bufs, p, eng.io(not_infected_probs=probs, new_infected=new, exp_noise=noise.to(device).contiguous()))
    torch.cuda.synchronize()
    got = probs.cpu()
    assert not bool(torch.isnan(got).any())
    assert np.abs(got.numpy() - ref["not_infected_probs"].numpy()).max() <= 1e-5
    assert torch.equal(new.cpu() > 0.5, ref["new_infected"] > 0.5)
    assert torch.equal(st["is_infected"].cpu(), ref["is_infected"])


@pytest.mark.parametrize("term, saturates", [(1000.0, False), (10000.0, True)])
def test_large_venue_sums_saturate_and_never_wrap(device, term, saturates):
    """20 000 attendees with a transmission of 1 000 / 10 000 each in ONE venue: 2e7 is summed exactly; 2e8 is beyond
    what the 2^-36 fixed point holds in 64 bits (1.3e8) and used to wrap silently.  The window of a TERM is chosen per
    set from its largest venue (gj_tiled_set.max_venue_edges: 2^26 / 32 768 = 2 048 here), so terms x attendees cannot
    leave the 64 bits: terms of 10 000 saturate the venue, and its attendees end at the reference's clamp, exp(-100)."""
    import gj_oracle as O
    from grad_june_amd.engine import AgentBuffers

    n = 20_000
    world = {"n_agents": n, "age": torch.zeros(n, dtype=torch.int64), "sex": torch.zeros(n, dtype=torch.int64),
             "edge_sets": {"school": {"agent": torch.arange(n), "venue": torch.zeros(n, dtype=torch.int64),
                                      "people": torch.tensor([n])}}}
    eng = L.make_engine(world, None, device, layout="tiled")
    assert eng.plan.tiled_c.sets[0].max_venue_edges == n
    st = {k: torch.zeros(n, device=device) for k in ("max_infectiousness", "shape", "rate", "shift", "infection_time",
                                                     "is_infected", "current_stage")}
    st["susceptibility"] = torch.ones(n, device=device)
    st["susceptibility"][::7] = 0.0
    st["transmission"] = torch.full((n,), term, device=device)
    beta = 0.5
    p = eng.params(now=1.0, delta_time=1.0, day_type=0, active=["school"], betas={"school": beta})
    bufs = AgentBuffers(eng.plan, **{k: st[k] for k in ("max_infectiousness", "shape", "rate", "shift", "infection_time",
                                                        "is_infected", "susceptibility", "transmission", "current_stage")})
    eng.venue_reduce(bufs, p)
    cum = float(eng.plan.cum_of("school")[0, 0])
    exact = beta * (1.0 / (n - 1)) * term * n
    probs = torch.empty(n, device=device)
    eng.agent_gather(bufs, p, eng.io(not_infected_probs=probs), sample=False)
    ref_ts = torch.clamp(torch.tensor(exact) * st["susceptibility"].cpu(), 1e-6, 100.0)
    ref_p = torch.exp(-ref_ts).clamp(0.0, 1.0)
    if saturates:
        assert np.isfinite(cum) and cum >= exact                     # +1e30 x beta x p_contact: finite, never a wrapped sum
    else:
        assert abs(cum - exact) <= 1e-6 * exact
    assert np.abs(probs.cpu().numpy() - ref_p.numpy()).max() <= 1e-6


def O_sample(p, noise):
    import gj_oracle as O

    return O.sample_infected(p, noise)


@pytest.mark.parametrize("layout", LAYOUTS[:2], ids=LAYOUT_IDS[:2])
def test_argument_errors_are_reported_not_launched(device, layout):
    """The ABI's error contract on a live plan: negative codes for bad arguments, nothing is written."""
    import ctypes as C

    from grad_june_amd import _native as N
    from grad_june_amd.engine import AgentBuffers

    lib = N.load()
    npz = L.load_npz("c100.npz")
    world = L.world_from(npz)
    eng = engine_for(world, None, device, layout)
    rec = L.step_record(npz, "plain_t3/")
    st = L.device_state(L.pre_state(rec), device)
    bufs = AgentBuffers(eng.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"], rate=st["rate"],
                        shift=st["shift"], infection_time=st["infection_time"], is_infected=st["is_infected"],
                        susceptibility=st["susceptibility"], transmission=st["transmission"])
    probs, new = torch.full((100,), -7.0, device=device), torch.full((100,), -7.0, device=device)
    io = eng.io(not_infected_probs=probs, new_infected=new)
    names = list(eng.plan.networks)
    betas = {n: 1.0 for n in names}

    def call(p, state=bufs.c, io_=io):
        return lib.gj_step(C.byref(eng.plan.c), C.byref(state) if state is not None else None, C.byref(p),
                           C.byref(io_), N.current_stream())

    good = eng.params(now=3.0, delta_time=1.0, day_type=0, active=names, betas=betas)
    p = eng.params(now=3.0, delta_time=1.0, day_type=0, active=names, betas=betas)
    p.n_nets = N.GJ_MAX_NETS + 1
    assert call(p) == -2                                            # GJ_E_RANGE
    p = eng.params(now=3.0, delta_time=1.0, day_type=2, active=names, betas=betas)
    assert call(p) == -2
    p = eng.params(now=3.0, delta_time=1.0, day_type=0, active=names, betas=betas)
    p.nets[0].set = 99
    assert call(p) == -2
    p = eng.params(now=3.0, delta_time=1.0, day_type=0, active=names, betas=betas)
    p.nets[0].mask_kind = 17
    assert call(p) == -2
    if len(names) >= 3:                                             # networks of one set must be adjacent
        p = eng.params(now=3.0, delta_time=1.0, day_type=0, active=names[:3], betas=betas)
        p.nets[2].set = p.nets[0].set
        assert call(p) == -3                                        # GJ_E_PLAN
    p = eng.params(now=3.0, delta_time=1.0, day_type=0, active=names, betas=betas)
    p.nets[0].mask_kind = N.MASK_QL                                 # a leisure mask without a table
    p.nets[0].table = -1
    assert call(p) == -3
    assert call(good, state=None) == -1                             # GJ_E_NULL
    broken = N.AgentState()
    C.memmove(C.byref(broken), C.byref(bufs.c), C.sizeof(N.AgentState))
    broken.susceptibility = None
    assert call(good, state=broken) == -1
    p = eng.params(now=3.0, delta_time=1.0, day_type=0, active=names, betas=betas, has_quarantine=True, q_threshold=4.0)
    assert call(p) == -1                                            # quarantine without stage / q_transmission
    torch.cuda.synchronize()
    assert (probs == -7.0).all() and (new == -7.0).all()            # no launch wrote anything
    assert N.load().gj_error_string(-3).decode() != ""
    assert call(good) == 0
    torch.cuda.synchronize()
    assert (probs > 0).all() and (probs <= 1).all()
