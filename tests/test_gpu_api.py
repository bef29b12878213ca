"""GPU: the reference-shaped Python API (InfectionNetworks / GradJune / Runner ...) running on the
HIP kernels.  Each test restates a test of the reference's own suite (named in its docstring,
paths relative to /root/reference/test/unit/) against this package."""
import datetime

import numpy as np
import pytest
import torch

import gj_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture()
def G(device):
    import grad_june_amd as g

    return g


def conftest_world(G, device, seed=999):
    """The reference's 100-agent fixture graph (conftest.py:36-89): 4 schools x 25, 4 companies x 25,
    25 households x 4, every 10th agent infected at t=0."""
    torch.manual_seed(seed)
    from grad_june_amd.defaults import default_parameters
    from grad_june_amd.infection import infect_people_at_indices

    n = 100
    d = G.HeteroData()
    ag = d["agent"]
    ag.id = torch.arange(n)
    ag.age = torch.randint(0, 100, (n,))
    ag.sex = torch.randint(0, 2, (n,))
    v = G.TransmissionSampler.from_parameters(default_parameters("cpu"))(n)
    ag.infection_parameters = {"max_infectiousness": v[0], "shape": v[1], "rate": v[2], "shift": v[3]}
    ag.transmission = torch.zeros(n)
    ag.susceptibility = torch.ones(n)
    ag.is_infected = torch.zeros(n)
    ag.infection_time = torch.zeros(n)
    ag.symptoms = {"current_stage": torch.ones(n, dtype=torch.long), "next_stage": torch.ones(n, dtype=torch.long),
                   "time_to_next_stage": torch.zeros(n)}
    for name, nv, per in (("school", 4, 25), ("company", 4, 25), ("household", 25, 4)):
        d[name].id = torch.arange(nv)
        d[name].people = per * torch.ones(nv)
        d["agent", "attends_" + name, name].edge_index = torch.vstack(
            (torch.arange(n), torch.tensor(np.repeat(np.arange(nv), per))))
    d = G.ToUndirected()(d).to(device)
    return infect_people_at_indices(d, list(range(0, 100, 10)), device=device)


def day_timer(G, acts, weekend=None, day="2022-02-01"):
    return G.Timer(initial_day=day, total_days=10, weekday_step_duration=(24,), weekend_step_duration=(24,),
                   weekday_activities=(tuple(acts),), weekend_activities=(tuple(weekend or acts),))


def nets_of(G, device, **log_betas):
    from grad_june_amd import infection_networks as inw

    cls = {"school": inw.SchoolNetwork, "company": inw.CompanyNetwork, "household": inw.HouseholdNetwork}
    return G.InfectionNetworks(device=device, **{k: cls[k](log_beta=v, device=device) for k, v in log_betas.items()})


def test_infection_passing_kat(G, device):
    """infection_networks/test_base.py:16-44 (exact known answer)."""
    d = G.HeteroData()
    d["agent"].id = torch.arange(6)
    d["agent"].transmission = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    d["agent"].susceptibility = torch.tensor([1, 2, 3, 0.5, 0.7, 1.0])
    d["school"].id = torch.arange(2)
    d["school"].people = torch.tensor([2, 2])
    d["agent", "attends_school", "school"].edge_index = torch.vstack((torch.arange(6), torch.tensor([0, 0, 0, 1, 1, 1])))
    d = G.ToUndirected()(d).to(device)
    nets = nets_of(G, device, school=np.log10(2.0))
    p = nets(data=d, timer=day_timer(G, ["school"]), policies=G.Policies())
    assert np.allclose(p.cpu().numpy(), np.exp(-np.array([1.2, 2.4, 3.6, 1.5, 2.1, 3])))
    ts = nets["school"](data=d, timer=day_timer(G, ["school"]), policies=G.Policies())     # single-network term
    assert np.allclose(ts.cpu().numpy(), [1.2, 2.4, 3.6, 1.5, 2.1, 3.0])


def test_leisure_network_masks(G, device):
    """infection_networks/test_leisure_network.py:41-77."""
    from grad_june_amd.infection_networks import LeisureNetwork

    d = G.HeteroData()
    d["agent"].id = torch.arange(5)
    d["agent"].age = torch.tensor([1, 60, 20, 30, 50])
    d["agent"].sex = torch.tensor([0, 1, 1, 0, 0])
    d["agent"].susceptibility = 0.5 * torch.ones(5)
    d["agent"].transmission = 2.0 * torch.ones(5)
    d["leisure"].id = torch.arange(3)
    d["agent", "attends_leisure", "leisure"].edge_index = torch.tensor([[0, 1, 2], [0, 0, 0]])
    d = G.ToUndirected()(d).to(device)
    probs = {"weekday": {"male": {"0-50": 0.5, "50-100": 0.2}, "female": {"0-100": 0.5}},
             "weekend": {"male": {"0-100": 1.0}, "female": {"0-100": 1.0}}}
    ln = LeisureNetwork(log_beta=0.0, device=device, leisure_probabilities=probs)
    ln.initialize_leisure_probabilities(d)
    assert (ln.weekday_probabilities.cpu() == torch.tensor([0.5, 0.5, 0.5, 0.5, 0.2])).all()
    assert (ln.weekend_probabilities.cpu() == torch.ones(5)).all()
    assert (ln._get_edge_index(d) == d["agent", "attends_leisure", "leisure"].edge_index).all()
    assert (ln._get_reverse_edge_index(d) == d["leisure", "rev_attends_leisure", "agent"].edge_index).all()
    t = G.Timer(initial_day="2022-05-20")          # a Friday
    pol = G.Policies()
    assert (ln._get_susceptibilities(d, pol, t).cpu() == 0.5 * torch.tensor([0.5, 0.5, 0.5, 0.5, 0.2])).all()
    next(t); next(t)                                # Saturday
    assert (ln._get_transmissions(d, pol, t).cpu() == 2.0 * torch.ones(5)).all()


def test_close_venue_integration(G, device):
    """policies/test_close_venue_policies.py:46-69."""
    from grad_june_amd.policies import CloseVenue

    d = conftest_world(G, device)
    d["agent"]["transmission"] = d["agent"]["transmission"] + 1.0
    nets = nets_of(G, device, company=3.0)
    t = day_timer(G, ["company"])
    ret = nets(data=d, timer=t, policies=G.Policies.from_policy_list([]))
    assert np.isclose(ret.sum().item(), 10.0)
    pol = G.Policies.from_policy_list([CloseVenue(names=("company",), start_date="2022-02-01", end_date="2022-02-05")])
    assert np.isclose(nets(data=d, timer=t, policies=pol).sum().item(), 100)


def test_quarantine_integration(G, device):
    """policies/test_quarantine_policies.py:40-72."""
    from grad_june_amd.policies import Quarantine

    d = conftest_world(G, device)
    d["agent"]["transmission"] = d["agent"]["transmission"] + 1.0
    d["agent"]["symptoms"]["current_stage"] = 5 * torch.ones(100, device=device)
    nets = nets_of(G, device, company=3.0, household=3.0)
    t = day_timer(G, ["company"], weekend=["company", "household"])
    pol = G.Policies.from_policy_list([Quarantine(stage_threshold=3, start_date="2022-02-01", end_date="2022-03-15")])
    assert np.isclose(nets(data=d, timer=t, policies=pol).sum().item(), 100)
    while not t.is_weekend:
        next(t)
    assert np.isclose(nets(data=d, timer=t, policies=pol).sum().item(), 10.0)


def test_social_distancing_integration(G, device):
    """policies/test_interaction_policies.py:92-123: exact ratio of exposures under two factors."""
    from grad_june_amd.policies import SocialDistancing

    d = conftest_world(G, device)
    d["agent"]["transmission"] = d["agent"]["transmission"] + 1.0
    nets = nets_of(G, device, school=2.0, company=0.0)
    t = day_timer(G, ["company"])
    out = []
    for f in ({"school": 0.3, "company": 0.5}, {"school": 0.6, "company": 0.2}):
        pol = G.Policies.from_policy_list([SocialDistancing(start_date="2022-02-01", end_date="2022-02-05", beta_factors=f)])
        ts = -torch.log(nets(data=d, timer=t, policies=pol)) * t.duration
        out.append(ts.cpu())
    keep = out[0] > 5e-6
    assert np.allclose((out[1][keep] / out[0][keep]).numpy(), 0.2 / 0.5)


def test_model_step_and_new_case_stage(G, device):
    """test_model.py:25-33 and :169-185: a step at t=3 infects people; new cases are 'exposed' (2)."""
    d = conftest_world(G, device)
    model = G.GradJune(infection_networks=nets_of(G, device, company=0.5, household=0.5, school=0.5),
                       policies=G.Policies.from_policy_list([]), device=device)
    t = G.Timer(initial_day="2022-02-01", total_days=10, weekday_step_duration=(8, 8, 8), weekend_step_duration=(12, 12),
                weekday_activities=(("company", "school", "household"), ("household",), ("household",)),
                weekend_activities=(("household",), ("household",)))
    while t.now < 3:
        next(t)
    before = d["agent"].is_infected.clone()
    with torch.no_grad():
        for _ in range(4):          # a few draws at the same time step, as test_model.py:84-85 does: P(no new case) ~ 8 % per draw
            res = model(timer=t, data=d)
    assert res["agent"]["is_infected"].sum() > 10 and res["agent"]["susceptibility"].sum() < 90
    new = (res["agent"].is_infected - before) > 0.5
    assert (res["agent"].symptoms["current_stage"][new & (before < 0.5)] == 2).all()
    assert (res["agent"].infection_time[new & (before < 0.5)] == t.now).all()


def test_model_matches_oracle_with_injected_noise(G, device):
    """GradJune.hot_path through the API == CPU oracle on the same noise (probabilities 1e-5, decisions equal)."""
    d = conftest_world(G, device)
    model = G.GradJune(infection_networks=nets_of(G, device, company=0.5, household=0.3, school=0.4),
                       policies=G.Policies.from_policy_list([]), device=device)
    t = day_timer(G, ["company", "school", "household"])
    for _ in range(3):
        next(t)
    ag = d["agent"]
    world = {"n_agents": 100, "age": ag.age.cpu(), "sex": ag.sex.cpu(), "edge_sets": {
        s: {"agent": d["attends_" + s].edge_index[0].cpu(), "venue": d["attends_" + s].edge_index[1].cpu(),
            "people": d[s].people.cpu()} for s in ("school", "company", "household")}}
    st = {k: ag["infection_parameters"][k].cpu() for k in ("max_infectiousness", "shape", "rate", "shift")}
    st.update({k: ag[k].cpu().clone() for k in ("infection_time", "is_infected", "susceptibility")})
    st["current_stage"] = ag.symptoms["current_stage"].cpu()
    noise = O.draw_exp_noise(100)
    betas = {n.name: n.beta_value(model.policies, t) for n in model.infection_networks.networks.values()}
    ref = O.hot_path_step(world, st, now=t.now, delta_time=t.duration, day_type=0, active=list(betas), betas=betas,
                          quarantine_thresholds=[], exp_noise=noise)
    with torch.no_grad():
        new, probs = model.hot_path(d, t, exp_noise=noise, want_probs=True)
    assert np.abs(probs.cpu().numpy() - ref["not_infected_probs"].numpy()).max() <= 1e-5
    assert np.array_equal(new.cpu().numpy() > 0.5, ref["new_infected"].numpy() > 0.5)
    assert np.array_equal(ag.is_infected.cpu().numpy(), ref["is_infected"].numpy())
    assert np.allclose(ag.transmission.cpu().numpy(), ref["transmission"].numpy(), rtol=2e-5, atol=1e-9)


def test_transmission_updater_and_sampler(G, device):
    """test_transmission.py:9-33."""
    from grad_june_amd.defaults import default_parameters

    torch.manual_seed(999)
    s = G.TransmissionSampler.from_parameters(default_parameters("cpu"))(20000)
    assert np.allclose(s.mean(1).numpy(), [1.1331, 1.56, 0.53, -2.12], rtol=2e-2)
    d = conftest_world(G, device)
    t = day_timer(G, ["household"])
    for _ in range(5):
        next(t)
    tr = G.TransmissionUpdater()(data=d, timer=t).cpu()
    assert (tr[::10] > 0).all() and tr.sum() == tr[::10].sum()
    d["agent"].is_infected = torch.zeros(100, device=device)
    assert G.TransmissionUpdater()(data=d, timer=t).sum() == 0


def test_transmission_profile_negative_base(G, device):
    """transmission.py:45-49 with t < shift: ``torch.pow`` of a negative base is finite for an INTEGER exponent
    (shape - 1) and the sign factor makes the product 0; for a non-integer exponent it is NaN and 0 * NaN = NaN
    (the v_log/v_exp form of pow returned NaN for every negative base).  Forward and the adjoint."""
    d = conftest_world(G, device)
    n = 100
    ag = d["agent"]
    shape = torch.tensor([2.0, 3.0, 4.0, 1.0, 1.56], device=device).repeat(n // 5)
    ag.infection_parameters = {"max_infectiousness": torch.full((n,), 1.3, device=device), "shape": shape,
                               "rate": torch.full((n,), 0.53, device=device),
                               "shift": torch.full((n,), 7.5, device=device)}       # positive: t - shift < 0 at t = 5
    ag.is_infected = torch.ones(n, device=device)
    ag.infection_time = torch.zeros(n, device=device)
    t = day_timer(G, ["household"])
    for _ in range(5):
        next(t)
    ip = {k: v.cpu() for k, v in ag.infection_parameters.items()}
    for now_shift in (0.0, 6.0):                    # t - shift = -2.5 (negative base), then +3.5 (the ordinary branch)
        ag.infection_time = torch.full((n,), -now_shift, device=device)
        ref = O.transmission_update(ip["max_infectiousness"], ip["shape"], ip["rate"], ip["shift"],
                                    ag.infection_time.cpu(), ag.is_infected.cpu(), t.now)
        got = G.TransmissionUpdater()(data=d, timer=t).cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(ref))
        ok = ~torch.isnan(ref)
        assert np.allclose(got[ok].numpy(), ref[ok].numpy(), rtol=2e-5, atol=1e-12)
        if now_shift == 0.0:
            sh = ip["shape"]
            assert (got[sh != 1.56] == 0).all() and torch.isnan(got[sh == 1.56]).all()


def test_is_infected_sampler_statistics(G, device):
    """infection_networks/test_is_infected_sampler.py:7-24 (mean of 2000 draws ~ 1-p, rtol 0.1)."""
    sampler = G.IsInfectedSampler()
    p = torch.tensor([0.2, 0.5, 0.7, 0.3], device=device).repeat(50_000)    # 50 000 draws of each in one launch
    x = sampler(p).cpu().view(50_000, 4)
    assert set(np.unique(x.numpy()).tolist()) <= {0.0, 1.0}
    assert np.allclose(x.mean(0).numpy(), [0.8, 0.5, 0.3, 0.7], rtol=0.03)
    p = p[:2000]
    noise = O.draw_exp_noise(2000)
    assert torch.equal(sampler(p, exp_noise=noise).cpu() > 0.5, O.sample_infected(p.cpu(), noise) > 0.5)


def test_is_infected_sampler_straight_through_gradient(G, device):
    """IsInfectedSampler in grad mode = F.gumbel_softmax(tau=0.1, hard=True) (infection.py:13-18): same hard
    decisions and the same straight-through gradient as the oracle's autograd for the same noise."""
    sampler = G.IsInfectedSampler()
    g = torch.Generator().manual_seed(8)
    n = 5000
    p0 = torch.rand(n, generator=g) * 0.98 + 0.01
    noise = O.draw_exp_noise(n)
    w = torch.rand(n, generator=g)
    p_dev = p0.clone().to(device).requires_grad_()
    out = sampler(p_dev, exp_noise=noise)
    assert out.requires_grad and set(np.unique(out.detach().cpu().numpy().round(6)).tolist()) <= {0.0, 1.0}
    got, = torch.autograd.grad((out * w.to(device)).sum(), p_dev)
    p_ref = p0.clone().requires_grad_()
    ref_out = O.sample_infected(p_ref, noise)
    assert torch.equal(out.detach().cpu() > 0.5, ref_out.detach() > 0.5)
    ref, = torch.autograd.grad((ref_out * w).sum(), p_ref)
    assert torch.allclose(got.cpu(), ref, rtol=1e-3, atol=1e-5)
    assert sampler(p_dev).requires_grad                                    # own noise: still on the graph


def test_standalone_probabilities_are_differentiable(G, device):
    """The stand-alone InfectionNetworks.forward (base.py:118-141) in grad mode: gradients w.r.t. every log_beta,
    the transmissions and the susceptibilities equal autograd through the oracle's op-for-op restatement; with a
    quarantine policy active too; and one network's trans_susc (InfectionNetwork.forward, base.py:61-84)."""
    from grad_june_amd.policies import Quarantine

    for quarantine in (False, True):
        d = conftest_world(G, device)
        n = 100
        g = torch.Generator().manual_seed(4)
        trans0 = torch.rand(n, generator=g) * (torch.rand(n, generator=g) < 0.4)
        susc0 = torch.rand(n, generator=g)
        stage = torch.randint(1, 7, (n,), generator=g)
        d["agent"].symptoms["current_stage"] = stage.to(device)
        w = torch.rand(n, generator=g)
        log_betas = {"school": 0.3, "company": 0.5, "household": 0.2}
        names = ["school", "company", "household"]
        policies = G.Policies.from_policy_list(
            [Quarantine(stage_threshold=4, start_date="2022-01-01", end_date="2022-12-01")] if quarantine else [])
        nets = nets_of(G, device, **{k: torch.nn.Parameter(torch.tensor(v)) for k, v in log_betas.items()})
        tr = trans0.clone().to(device).requires_grad_()
        su = susc0.clone().to(device).requires_grad_()
        d["agent"].transmission, d["agent"].susceptibility = tr, su
        timer = day_timer(G, names)
        p = nets(data=d, timer=timer, policies=policies)
        assert p.requires_grad
        loss = (p * w.to(device)).sum()
        got = torch.autograd.grad(loss, [tr, su] + [nets[k].log_beta for k in names])
        # oracle: the same computation with autograd on the CPU
        lb = {k: torch.tensor(v, requires_grad=True) for k, v in log_betas.items()}
        tr_c, su_c = trans0.clone().requires_grad_(), susc0.clone().requires_grad_()
        qmask = O.quarantine_mask(stage.float(), [4.0]) if quarantine else 1.0
        ts = []
        for k in names:
            ei = d["attends_" + k].edge_index.cpu()
            ts.append(O.infection_network(kind="household" if k == "household" else "plain", beta=10.0 ** lb[k],
                                          people=d[k].people.cpu(), agent_index=ei[0], venue_index=ei[1],
                                          transmission=tr_c, susceptibility=su_c, qmask=qmask))
        p_ref = O.not_infected_probabilities(ts, n, 1.0)
        assert torch.allclose(p.detach().cpu(), p_ref.detach(), rtol=2e-5, atol=1e-9)
        ref = torch.autograd.grad((p_ref * w).sum(), [tr_c, su_c] + [lb[k] for k in names], retain_graph=True)
        for a, b, what in zip(got, ref, ["transmission", "susceptibility"] + names):
            assert torch.allclose(a.cpu(), b, rtol=2e-3, atol=1e-6), (quarantine, what, a, b)
        # a single network's term
        t1 = nets["company"](data=d, timer=timer, policies=policies)
        g1 = torch.autograd.grad(t1.sum(), [tr, nets["company"].log_beta])
        r1 = torch.autograd.grad(ts[1].sum(), [tr_c, lb["company"]], retain_graph=True)
        for a, b in zip(g1, r1):
            assert torch.allclose(a.cpu(), b, rtol=2e-3, atol=1e-6)


def test_run_model_script_flow_with_gradients(G, device):
    """example_scripts/run_model.py:5-11 and test_model.py:34-53: household log_beta as nn.Parameter,
    run, cases_per_timestep.sum().backward() -> a finite, non-zero gradient."""
    torch.manual_seed(5)
    runner = G.Runner.from_parameters(params_on(device, days=8))
    for key in ("household", "company"):
        net = runner.model.infection_networks.networks[key]
        net.log_beta = torch.nn.Parameter(net.log_beta)
    results, is_infected = runner()
    assert results["cases_per_timestep"].requires_grad and results["cases_by_age_65"].requires_grad
    cases = results["cases_per_timestep"].sum()
    cases.backward()
    for key in ("household", "company"):
        g = runner.model.infection_networks.networks[key].log_beta.grad
        assert g is not None and torch.isfinite(g) and g != 0
    # the forward values are those of the kernel reductions
    assert results["cases_per_timestep"][-1].item() == pytest.approx(is_infected.sum().item())


def test_deaths_gradient(G, device):
    """test_runner.py:82-90: in a differentiable run the deaths series stays on the autograd graph
    (one record per day + the initial one) and equals data["results"]["deaths_per_timestep"]."""
    torch.manual_seed(6)
    runner = G.Runner.from_parameters(params_on(device, days=15))
    for net in runner.model.infection_networks.networks.values():
        net.log_beta = torch.nn.Parameter(net.log_beta)
    results, is_infected = runner()
    assert results["cases_per_timestep"].requires_grad
    daily_deaths = runner.data["results"]["deaths_per_timestep"]
    assert (results["deaths_per_timestep"] == daily_deaths).all()
    assert daily_deaths.shape[0] == runner.input_parameters["timer"]["total_days"] + 1
    assert daily_deaths.requires_grad
    # the graph reaches the parameters through the symptoms updates: backward runs and is finite
    (daily_deaths.sum() + results["cases_per_timestep"][-1]).backward()
    g = runner.model.infection_networks.networks["household"].log_beta.grad
    assert g is not None and torch.isfinite(g)
    # forward values are those of the fused reduction kernel (gj_step_stats), which records every run
    stats = runner._series[: daily_deaths.shape[0]].to(torch.float32)
    assert torch.equal(stats[:, 0], results["cases_per_timestep"].detach())
    assert torch.equal(stats[:, -1], daily_deaths.detach())


def params_on(device, days=15):
    from grad_june_amd.defaults import default_parameters

    p = default_parameters(str(device))
    p["timer"]["total_days"] = days
    return p


def test_runner_records_and_csv(G, device, tmp_path):
    """test_runner.py:25-90: 16 records for 15 days, result keys, CSV round trip."""
    import pandas as pd

    p = params_on(device)
    p["save_path"] = str(tmp_path / "out")
    torch.manual_seed(769)
    runner = G.Runner.from_parameters(p)
    assert runner.n_agents == 769
    with torch.no_grad():
        runner.set_initial_cases()
        assert np.isclose(runner.data["agent"].is_infected.sum().item(), 0.10 * 769, rtol=3e-1)
        results, is_inf = runner()
    assert len(results["dates"]) == 16 and results["dates"][0] == datetime.datetime(2022, 2, 1)
    for key in ("cases_per_timestep", "daily_cases_per_timestep", "deaths_per_timestep", "cases_by_age_18",
                "cases_by_age_65", "cases_by_age_100"):
        assert results[key].shape[0] == 16, key
    c = results["cases_per_timestep"].cpu().numpy()
    assert (np.diff(c) >= 0).all() and c[-1] >= c[0] > 0
    assert is_inf.shape[0] == 769
    runner.save_results(results, is_inf)
    df = pd.read_csv(tmp_path / "out" / "results.csv", index_col=0)
    assert len(df) == 16 and np.allclose(df["cases_per_timestep"].values, c)
    assert len(pd.read_csv(tmp_path / "out" / "results_is_infected.csv")) == 769
    # restore_initial_data: everyone back to susceptible
    runner.restore_initial_data()
    assert runner.data["agent"].symptoms["current_stage"].sum().item() == 769


def test_runner_series_match_plain_reductions(G, device):
    """Row f2: the fused per-step reductions (gj_step_stats) equal the reference's formulas
    (runner.py:167, 198-224) evaluated with plain tensor ops on the final state."""
    torch.manual_seed(11)
    p = params_on(device, days=6)
    for k in p["networks"]:
        p["networks"][k]["log_beta"] += 1.0
    runner = G.Runner.from_parameters(p)
    with torch.no_grad():
        results, is_inf = runner()
    data = runner.data
    assert results["cases_per_timestep"][-1].item() == pytest.approx(is_inf.sum().item())
    assert torch.allclose(results["cases_by_age_18"][-1], runner.get_cases_by_age(data)[0])
    assert torch.allclose(results["cases_by_age_65"][-1], runner.get_cases_by_age(data)[1])
    assert torch.allclose(results["cases_by_age_100"][-1], runner.get_cases_by_age(data)[2])
    stage = data["agent"].symptoms["current_stage"]
    assert results["deaths_per_timestep"][-1].item() == (stage == 7).sum().item()
    assert results["cases_per_timestep"][0].item() > 0 and results["daily_cases_per_timestep"][0] == results["cases_per_timestep"][0]
    # open age intervals: ages exactly on a bin edge are in no bin (reference quirk, runner.py:220-222)
    edge = (data["agent"].age == 18) | (data["agent"].age == 65) | (data["agent"].age == 0)
    by_age_total = sum(results[f"cases_by_age_{k:02d}"][-1].item() for k in (18, 65, 100))
    assert by_age_total == pytest.approx((is_inf * (~edge)).sum().item())


def test_config1_plumbing_30_timesteps(G, device):
    """BASELINE.json configs[0]: the shipped 769-agent world (the London blob is absent), default
    parameters, 30 timesteps through Runner - every step on the HIP path."""
    torch.manual_seed(1)
    runner = G.Runner.from_parameters(params_on(device, days=30))
    with torch.no_grad():
        results, _ = runner()
    assert len(results["dates"]) == 31
    assert runner.model.n_steps == 30
    assert torch.isfinite(results["cases_per_timestep"]).all()


@pytest.mark.parametrize("example", ["abi_demo", "compile_demo"])
def test_c_abi_from_plain_c(device, tmp_path, example):
    """examples/abi_demo.c (sampler + result reductions) and examples/compile_demo.c (the graph compile: COO edge_index
    -> tiled layout + ELL rows, with the layout's invariants checked in C): the library called from C with
    hipMalloc'ed buffers - no Python objects involved."""
    import os
    import subprocess

    from grad_june_amd import _native as N

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "gradabm-june_amd", "grad_june_amd", "lib")
    exe = str(tmp_path / example)
    subprocess.run(["gcc", os.path.join(root, "examples", example + ".c"), "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-I", os.path.join(root, "include"), "-L", lib_dir, "-lgradjune_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.rstrip().endswith("ok") and (example != "abi_demo" or f"ABI version {N.GJ_ABI_VERSION}" in out)


def test_locality_order_gives_the_same_probabilities(G, device):
    """system.locality_order renumbers the agents household-major at load time: the infection probabilities
    of every agent are those of the original numbering ("results reported in original
    agent order"), and runner() returns its per-agent result in the file's order."""
    torch.manual_seed(3)
    plain = G.Runner.from_parameters(params_on(device, days=3))
    p2 = params_on(device, days=3)
    p2["system"]["locality_order"] = "household"
    torch.manual_seed(3)
    local = G.Runner.from_parameters(p2)
    original = local.data["agent"].original_index.to(device)
    assert not torch.equal(original, torch.arange(len(original), device=device))
    assert torch.equal(local.data["agent"].age, plain.data["agent"].age[original])
    # same epidemic state in both numberings
    n = plain.n_agents
    g = torch.Generator().manual_seed(1)
    trans = torch.rand(n, generator=g).to(device) * (torch.rand(n, generator=g).to(device) < 0.2)
    susc = (torch.rand(n, generator=g).to(device) < 0.7).float()
    for r, idx in ((plain, None), (local, original)):
        ag = r.data["agent"]
        ag.transmission = trans if idx is None else trans[idx].contiguous()
        ag.susceptibility = susc if idx is None else susc[idx].contiguous()
    next(plain.timer), next(local.timer)
    pa = plain.model.infection_networks(data=plain.data, timer=plain.timer, policies=plain.model.policies)
    pb = local.model.infection_networks(data=local.data, timer=local.timer, policies=local.model.policies)
    assert torch.allclose(pb, pa[original], rtol=2e-6, atol=1e-9)
    with torch.no_grad():
        _, is_infected = local()
    assert is_infected.shape[0] == n
    assert torch.equal(is_infected[original], local.data["agent"].is_infected)
    # the household-major world with its household set in the RUN FORM (what a world too large for the direct form takes
    # by itself; forced here): every person of the reference's world lives in one household, so the set's tiled arrays
    # are empty - and the probabilities are still those of the original numbering
    from grad_june_amd import world as W

    try:
        W.RUNS = "household"
        torch.manual_seed(3)
        runs = G.Runner.from_parameters(p2)
        ag = runs.data["agent"]
        ag.transmission, ag.susceptibility = trans[original].contiguous(), susc[original].contiguous()
        next(runs.timer)
        pc = runs.model.infection_networks(data=runs.data, timer=runs.timer, policies=runs.model.policies)
        eng = W.engine_for(runs.data, [n_.spec() for n_ in runs.model.infection_networks.networks.values()], device)
        t = {s_.name: s_.tiled for s_ in eng.plan.host.sets}["household"]
        assert t.runs is not None and t.n_edges == 0 and t.runs.n_primary > 700
    finally:
        W.RUNS = "auto"
    assert torch.allclose(pc, pa[original], rtol=2e-6, atol=1e-9)


def test_api_with_device_side_compile(G, device):
    """world.DEVICE_COMPILE: the Runner's plan compiled on the GPU by the library's compile kernels (the default when the
    world's edge lists are on the device) gives the same run as the host-compiled one."""
    from grad_june_amd import world as W

    def run():
        import itertools

        from grad_june_amd import infection

        torch.manual_seed(9)
        infection._philox_step = itertools.count(1 << 40)
        runner = G.Runner.from_parameters(params_on(device, days=6))
        with torch.no_grad():
            res, inf = runner()
        return res["cases_per_timestep"].cpu(), inf.cpu()

    assert W.DEVICE_COMPILE == "auto"
    try:
        W.DEVICE_COMPILE = "0"
        ref = run()
        W.DEVICE_COMPILE = "1"
        got = run()
        W.DEVICE_COMPILE = "auto"
        auto = run()
    finally:
        W.DEVICE_COMPILE = "auto"
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and ref[0][-1] > 0
    assert torch.equal(auto[0], ref[0]) and torch.equal(auto[1], ref[1])


def test_api_geometry_tuning_changes_speed_not_results(G, device):
    """world.TUNE: a mid-size world compiled through the API mirrors (edge lists on the device) is compiled under a few
    tile geometries and the fastest kept; whichever it is, a step gives bitwise the same state as with the defaults."""
    import bench as B
    from grad_june_amd import world as W
    from grad_june_amd.engine import AgentBuffers
    from grad_june_amd.synthetic import make_world

    world = make_world("c2", n_agents=1_000_000, seed=5, infected_fraction=0.05)      # BASELINE configs[1]

    def data_of():
        d = G.HeteroData()
        ag = d["agent"]
        ag.id = torch.arange(world["n_agents"])
        ag.age, ag.sex = torch.from_numpy(world["age"]), torch.from_numpy(world["sex"])
        for s, es in world["edge_sets"].items():
            d[s].id = torch.arange(len(es["people"]))
            d[s].people = torch.from_numpy(es["people"])
            d["agent", "attends_" + s, s].edge_index = torch.vstack((torch.from_numpy(es["agent"]), torch.from_numpy(es["venue"])))
        return d.to(device)

    specs, betas = B.network_specs(world), B.betas_of(world)

    def step_with(tune):
        W.TUNE = tune
        try:
            data = data_of()
            eng = W.engine_for(data, specs, device)
        finally:
            W.TUNE = "auto"
        st = {k: torch.from_numpy(v).to(device) for k, v in world["state"].items()}
        n = world["n_agents"]
        new = torch.empty(n, device=device)
        bufs = AgentBuffers(eng.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"], rate=st["rate"],
                            shift=st["shift"], infection_time=st["infection_time"], is_infected=st["is_infected"],
                            susceptibility=st["susceptibility"], transmission=torch.zeros(n, device=device))
        p = eng.params(now=1.0, delta_time=1.0, day_type=0, active=[s.name for s in specs], betas=betas, seed=3, step=0)
        eng.step(bufs, p, eng.io(new_infected=new))
        torch.cuda.synchronize()
        geo = tuple((s.name, s.tiled.n_blocks) for s in eng.plan.host.sets)
        return st, new, geo, W._time_passes(eng, specs)

    st0, new0, geo0, ms0 = step_with("0")
    st1, new1, geo1, ms1 = step_with("auto")
    for k in ("is_infected", "susceptibility", "infection_time"):
        assert torch.equal(st0[k], st1[k]), k
    assert torch.equal(new0, new1) and float(new0.sum()) > 0
    assert ms1 <= ms0 * 1.10            # the tuner keeps the defaults unless another geometry measured faster
    print(f"defaults {ms0:.3f} ms {geo0}; tuned {ms1:.3f} ms {geo1}")


def test_tile_geometry_cannot_change_a_bit(G, device):
    """Every candidate geometry of the tuner (world.TUNE_CANDIDATES, bench's GEOMETRY_CANDIDATES without its
    direct=False variant) gives bitwise the same per-venue sums, per-agent trans_susc and probabilities: phase B merges
    the runs of a venue as fixed-point integers (round 2 merged them in fp32, where a run's position relative to the
    8-slot groups - i.e. the geometry - reached the last bits), phase D adds fixed-point terms, the direct form sums in
    COO order.  So a wall-clock race between geometries cannot make two runs of one script differ."""
    import bench as B
    from grad_june_amd import world as W
    from grad_june_amd.benchrun import GEOMETRY_CANDIDATES, SingleGpuHotPath
    from grad_june_amd.synthetic import make_world

    world = make_world("c3", n_agents=300_000, seed=11, infected_fraction=0.2)
    specs, betas = B.network_specs(world), B.betas_of(world)
    cands = [dict(c) for c in W.TUNE_CANDIDATES] + [dict(c) for c in GEOMETRY_CANDIDATES if "direct" not in c]
    cands.append({"eb_target": 4096, "sv_max": 64, "slice_agents": 1024})       # tiny tiles, runs cut everywhere
    ref = None
    seen = set()
    for cand in cands:
        cand = dict(cand)
        sa = cand.pop("slice_agents", None)
        if sa is not None:
            cand["slices"] = (-(-world["n_agents"] // sa), sa)
        r = SingleGpuHotPath(world, specs, betas, device, seed=3, device_compile=True, **cand)
        ts = torch.empty(world["n_agents"], device=device)
        r.io = r.engine.io(not_infected_probs=r.probs, new_infected=r.new_infected, trans_susc=ts)
        r.step()
        torch.cuda.synchronize()
        got = {"ts": ts.clone(), "probs": r.probs.clone(), "new": r.new_infected.clone()}
        for hs in r.engine.plan.host.sets:
            got["cum/" + hs.name] = r.engine.plan.cum_of(hs.name).clone()
        seen.add(tuple(hs.tiled.n_blocks for hs in r.engine.plan.host.sets) + (r.engine.plan.host.n_slices,))
        if ref is None:
            ref = got
            assert float(ts.max()) > 0 and float(got["new"].sum()) > 100
        else:
            for k in ref:
                assert torch.equal(got[k], ref[k]), (cand, k)
    assert len(seen) >= 4          # the candidates really are different geometries
