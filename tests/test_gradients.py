"""Row f3: d(cases)/d(log_beta) through several timesteps.

CPU: the oracle (autograd through its torch ops) against the gradients recorded from the reference.
GPU: the HIP path's hand-written adjoint (`GradJune` in grad mode) against the same records."""
import numpy as np
import pytest
import torch

import gj_oracle as O
import gj_testlib as L

CASES = ["g1", "g2"]


def load_case(case):
    npz = L.load_npz("grads_symptoms.npz" if case == "g3" else "grads.npz")
    pre = case + "/"
    sub = {k[len(pre):]: v for k, v in npz.items() if k.startswith(pre)}
    world = L.world_from(sub)
    tables = {k[6:]: torch.from_numpy(v) for k, v in sub.items() if k.startswith("table/")}
    names = str(sub["networks"]).split(",")
    return sub, world, tables, names


def step_info(sub, i):
    p = f"step{i}/"
    active = str(sub[p + "active"]).split(",") if str(sub[p + "active"]) else []
    thr = None
    if int(sub[p + "has_quarantine"]):
        thr = [None if np.isnan(t) else float(t) for t in sub[p + "q_thresholds"]]
    return dict(now=float(sub[p + "now"]), dt=float(sub[p + "dt"]), day_type=int(sub[p + "day_type"]), active=active,
                betas={n: float(sub[p + "beta/" + n]) for n in active}, thr=thr,
                noise=torch.from_numpy(sub[p + "exp_noise"]), stage=torch.from_numpy(sub[p + "current_stage"]),
                is_infected=sub[p + "is_infected"])


@pytest.mark.parametrize("case", CASES)
def test_oracle_autograd_matches_reference(case):
    sub, world, tables, names = load_case(case)
    # log_beta per network recovered from the recorded beta of a step without distancing factors is not
    # needed: d/dlog_beta = ln(10) * beta * d/dbeta, so differentiate w.r.t. a unit multiplier of beta
    mult = {n: torch.ones((), requires_grad=True) for n in names}
    st = {k[7:]: torch.from_numpy(v) for k, v in sub.items() if k.startswith("state0/")}
    series = []
    for i in range(int(sub["n_steps"])):
        s = step_info(sub, i)
        st["current_stage"] = s["stage"]
        betas = {n: torch.tensor(np.float32(s["betas"][n])) * mult[n] for n in s["active"]}
        out = O.hot_path_step(world, st, now=s["now"], delta_time=s["dt"], day_type=s["day_type"], active=s["active"],
                              betas=betas, leisure_tables=tables, quarantine_thresholds=s["thr"], exp_noise=s["noise"])
        for k in ("susceptibility", "is_infected", "infection_time"):
            st[k] = out[k]
        assert np.array_equal(out["is_infected"].detach().numpy(), s["is_infected"]), i
        series.append(out["is_infected"].sum())
    for tag, loss in (("last", series[-1]), ("series", torch.stack(series).sum())):
        grads = torch.autograd.grad(loss, [mult[n] for n in names], retain_graph=True, allow_unused=True)
        for n, g in zip(names, grads):
            got = 0.0 if g is None else float(g) * np.log(10.0)      # d/dlog_beta = ln10 * d/dmult
            ref = float(sub[f"grad_{tag}/{n}"])
            assert got == pytest.approx(ref, rel=2e-4, abs=1e-6), (tag, n)


@pytest.mark.parametrize("case", CASES)
def test_handwritten_adjoint_matches_reference(case):
    """The adjoint recursion the HIP backward implements, restated densely on the CPU: reverse sweep
    over the recorded steps, gradients of both losses against the reference's."""
    sub, world, tables, names = load_case(case)
    T = int(sub["n_steps"])
    st = {k[7:]: torch.from_numpy(v) for k, v in sub.items() if k.startswith("state0/")}
    states, infos = [], []
    for i in range(T):                                   # forward sweep (stores the pre-state of every step)
        s = step_info(sub, i)
        st["current_stage"] = s["stage"]
        states.append(dict(st))
        infos.append(s)
        out = O.hot_path_step(world, st, now=s["now"], delta_time=s["dt"], day_type=s["day_type"], active=s["active"],
                              betas=s["betas"], leisure_tables=tables, quarantine_thresholds=s["thr"], exp_noise=s["noise"])
        for k in ("susceptibility", "is_infected", "infection_time"):
            st[k] = out[k]
    A = world["n_agents"]
    for tag in ("last", "series"):
        total = {n: 0.0 for n in names}
        gs, gi, gt = torch.zeros(A), torch.zeros(A), torch.zeros(A)
        for i in reversed(range(T)):
            if tag == "series" or i == T - 1:
                gi = gi + 1.0                            # d loss / d is_infected after step i
            s = infos[i]
            gs, gi, gt, glb, _ = O.adjoint_step(world, states[i], now=s["now"], delta_time=s["dt"], day_type=s["day_type"],
                                                active=s["active"], betas=s["betas"], leisure_tables=tables,
                                                quarantine_thresholds=s["thr"], exp_noise=s["noise"],
                                                g_susc=gs, g_inf=gi, g_time=gt)
            for n, v in glb.items():
                total[n] += v
        for n in names:
            assert total[n] == pytest.approx(float(sub[f"grad_{tag}/{n}"]), rel=5e-4, abs=1e-5), (tag, n)


# ------------------------------------------------------------------------------------------------------
# g3: gradients that reach log_beta THROUGH the symptoms state machine (deaths-style losses,
# reference runner.py:198-215 / test_runner.py:82-90)
# ------------------------------------------------------------------------------------------------------
def symptom_losses(sub, sym, occupancy):
    """The losses recorded in g3 from the final symptoms state and the per-step occupancy sums."""
    w_cur, w_nxt = torch.from_numpy(sub["w_cur"]), torch.from_numpy(sub["w_nxt"])
    dev = sym["current_stage"].device
    losses = {"stage_lin": (w_cur.to(dev) * sym["current_stage"]).sum() + (w_nxt.to(dev) * sym["next_stage"]).sum()}
    for tag in str(sub["loss_tags"]).split(","):
        if tag.startswith("occupancy"):
            losses[tag] = torch.stack(occupancy[int(tag[9:])]).sum()
    return losses


def occupancy_of(cur, k):
    return ((cur == k) * cur / k).sum()          # store_differentiable_deaths' form for stage k (runner.py:204-215)


def test_oracle_autograd_through_symptoms_matches_reference():
    sub, world, tables, names = load_case("g3")
    mult = {n: torch.ones((), requires_grad=True) for n in names}
    st = {k[7:]: torch.from_numpy(v) for k, v in sub.items() if k.startswith("state0/") and "/sym/" not in k}
    sym = {k: torch.from_numpy(sub["state0/sym/" + k]) for k in ("current_stage", "next_stage", "time_to_next_stage")}
    n_stages = sub["sym_table"].shape[0]
    occupancy = {k: [] for k in range(2, n_stages)}
    for i in range(int(sub["n_steps"])):
        s = step_info(sub, i)
        st["current_stage"] = sym["current_stage"]
        assert np.array_equal(sym["current_stage"].detach().numpy(), sub[f"step{i}/sym/pre/current_stage"]), i
        betas = {n: torch.tensor(np.float32(s["betas"][n])) * mult[n] for n in s["active"]}
        out = O.hot_path_step(world, st, now=s["now"], delta_time=s["dt"], day_type=s["day_type"], active=s["active"],
                              betas=betas, leisure_tables=tables, quarantine_thresholds=s["thr"], exp_noise=s["noise"])
        for k in ("susceptibility", "is_infected", "infection_time"):
            st[k] = out[k]
        assert np.array_equal(out["is_infected"].detach().numpy(), s["is_infected"]), i
        cur, nxt, ttn = O.symptoms_update(world["age"], sym["current_stage"], sym["next_stage"], sym["time_to_next_stage"],
                                          out["new_infected"], s["now"], n_stages,
                                          torch.from_numpy(sub[f"step{i}/sym/progresses"]),
                                          torch.from_numpy(sub[f"step{i}/sym/dwell"]))
        sym = {"current_stage": cur, "next_stage": nxt, "time_to_next_stage": ttn}
        for k in sym:
            assert np.array_equal(sym[k].detach().numpy(), sub[f"step{i}/sym/post/{k}"]), (i, k)
        for k in occupancy:
            occupancy[k].append(occupancy_of(cur, k))
    for tag, loss in symptom_losses(sub, sym, occupancy).items():
        assert float(loss.detach()) == pytest.approx(float(sub["loss_" + tag]), rel=1e-6)
        grads = torch.autograd.grad(loss, [mult[n] for n in names], retain_graph=True, allow_unused=True) \
            if loss.requires_grad else [None] * len(names)
        for n, g in zip(names, grads):
            got = 0.0 if g is None else float(g) * np.log(10.0)
            assert got == pytest.approx(float(sub[f"grad_{tag}/{n}"]), rel=5e-4, abs=1e-6), (tag, n)


def test_handwritten_symptoms_adjoint_matches_autograd():
    """adjoint_symptoms against autograd through the op-for-op restatement, on random states that put
    agents in every branch (not due, due and progressing, due and recovering, newly infected, dead)."""
    g = torch.Generator().manual_seed(3)
    n, n_stages = 4000, 8
    cur = torch.randint(0, n_stages, (n,), generator=g).float().requires_grad_()
    nxt = torch.randint(0, n_stages, (n,), generator=g).float().requires_grad_()
    ttn = (torch.rand(n, generator=g) * 10).requires_grad_()
    new = (torch.rand(n, generator=g) < 0.2).float().requires_grad_()
    progresses = (torch.rand(n, generator=g) < 0.5).float()
    dwell = torch.rand(n, generator=g) * 5
    age = torch.randint(0, 100, (n,), generator=g)
    c, x, t = O.symptoms_update(age, cur, nxt, ttn, new, 5.0, n_stages, progresses, dwell)
    g_cur, g_nxt, g_ttn = (torch.randn(n, generator=g) for _ in range(3))
    ref = torch.autograd.grad((c * g_cur).sum() + (x * g_nxt).sum() + (t * g_ttn).sum(), (cur, nxt, ttn, new))
    got = O.adjoint_symptoms(cur.detach(), nxt.detach(), ttn.detach(), new.detach(), 5.0, n_stages, progresses,
                             g_cur, g_nxt, g_ttn, dwell)
    for a, b, what in zip(got, ref, ("current_stage", "next_stage", "time_to_next_stage", "new_infected")):
        assert torch.allclose(a.float(), b, rtol=1e-5, atol=1e-5), what


def test_handwritten_adjoint_through_symptoms_matches_reference():
    """Reverse sweep over g3 with the two hand-written adjoints (hot path + symptoms) only."""
    sub, world, tables, names = load_case("g3")
    T = int(sub["n_steps"])
    n_stages = sub["sym_table"].shape[0]
    st = {k[7:]: torch.from_numpy(v) for k, v in sub.items() if k.startswith("state0/") and "/sym/" not in k}
    states, infos = [], []
    for i in range(T):
        s = step_info(sub, i)
        st["current_stage"] = torch.from_numpy(sub[f"step{i}/sym/pre/current_stage"])
        states.append(dict(st))
        infos.append(s)
        out = O.hot_path_step(world, st, now=s["now"], delta_time=s["dt"], day_type=s["day_type"], active=s["active"],
                              betas=s["betas"], leisure_tables=tables, quarantine_thresholds=s["thr"], exp_noise=s["noise"])
        for k in ("susceptibility", "is_infected", "infection_time"):
            st[k] = out[k]
        states[-1]["new_infected"] = out["new_infected"]
    A = world["n_agents"]
    w_cur, w_nxt = torch.from_numpy(sub["w_cur"]).double(), torch.from_numpy(sub["w_nxt"]).double()
    for tag in str(sub["loss_tags"]).split(","):
        if tag == "deaths":
            continue
        total = {n: 0.0 for n in names}
        gs, gi, gt = torch.zeros(A), torch.zeros(A), torch.zeros(A)
        gc, gx = (w_cur, w_nxt) if tag == "stage_lin" else (torch.zeros(A).double(), torch.zeros(A).double())
        for i in reversed(range(T)):
            pre = {k: torch.from_numpy(sub[f"step{i}/sym/pre/{k}"]) for k in ("current_stage", "next_stage", "time_to_next_stage")}
            if tag.startswith("occupancy"):
                k = int(tag[9:])
                gc = gc + (torch.from_numpy(sub[f"step{i}/sym/post/current_stage"]) == k).double() / k
            s = infos[i]
            gc, gx, _, gnew = O.adjoint_symptoms(pre["current_stage"], pre["next_stage"], pre["time_to_next_stage"],
                                                 states[i]["new_infected"], s["now"], n_stages,
                                                 torch.from_numpy(sub[f"step{i}/sym/progresses"]), gc, gx)
            gs, gi, gt, glb, _ = O.adjoint_step(world, states[i], now=s["now"], delta_time=s["dt"], day_type=s["day_type"],
                                                active=s["active"], betas=s["betas"], leisure_tables=tables,
                                                quarantine_thresholds=s["thr"], exp_noise=s["noise"],
                                                g_susc=gs, g_inf=gi, g_time=gt, g_new=gnew)
            for n, v in glb.items():
                total[n] += v
        for n in names:
            assert total[n] == pytest.approx(float(sub[f"grad_{tag}/{n}"]), rel=1e-3, abs=1e-5), (tag, n)


# ------------------------------------------------------------------------------------------------------
# GPU: GradJune in grad mode (HIP forward + hand-written HIP adjoint) against the reference's gradients
# ------------------------------------------------------------------------------------------------------
def _hetero(G, sub, world, device):
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
    st = {k[7:]: torch.from_numpy(v).to(device) for k, v in sub.items() if k.startswith("state0/")}
    ag.infection_parameters = {k: st[k] for k in ("max_infectiousness", "shape", "rate", "shift")}
    for k in ("is_infected", "susceptibility", "infection_time"):
        ag[k] = st[k]
    ag.transmission = torch.zeros(A, device=device)
    ag.symptoms = {"current_stage": st["current_stage"].float(), "next_stage": st["current_stage"].float(),
                   "time_to_next_stage": torch.full((A,), 1e9, device=device)}
    return d


def _model_and_timer(G, case, device):
    from grad_june_amd import infection_networks as inw
    from grad_june_amd.defaults import default_parameters

    if case == "g1":
        nets = G.InfectionNetworks(device=device, household=inw.HouseholdNetwork(0.2, device),
                                   company=inw.CompanyNetwork(0.4, device), school=inw.SchoolNetwork(0.3, device))
        model = G.GradJune(infection_networks=nets, policies=G.Policies.from_policy_list([]), device=device)
        timer = G.Timer(initial_day="2022-02-01", total_days=10, weekday_step_duration=(24,), weekend_step_duration=(24,),
                        weekday_activities=(("company", "school", "household"),),
                        weekend_activities=(("company", "school", "household"),))
        next(timer); next(timer)
        return model, timer
    params = default_parameters(str(device))
    params["policies"]["quarantine"] = {
        "quarantine": {1: {"start_date": "2022-02-03", "end_date": "2022-02-20", "stage_threshold": 4}}}
    params["policies"]["interaction"]["social_distancing"][1]["start_date"] = "2022-02-04"
    for n in params["networks"]:
        params["networks"][n]["log_beta"] += 0.7
    return G.GradJune.from_parameters(params), G.Timer.from_parameters(params)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_backward_matches_reference(device, case):
    import grad_june_amd as G

    sub, world, tables, names = load_case(case)
    model, timer = _model_and_timer(G, case, device)
    data = _hetero(G, sub, world, device)
    for n in names:
        net = model.infection_networks.networks[n]
        net.log_beta = torch.nn.Parameter(net.log_beta.detach().clone())
    series = []
    for i in range(int(sub["n_steps"])):
        s = step_info(sub, i)
        next(timer)
        assert timer.now == s["now"]
        data["agent"].symptoms["current_stage"] = s["stage"].to(device)         # recorded symptom stage (quarantine mask)
        for n in s["active"]:
            assert np.float32(model.infection_networks[n].beta_value(model.policies, timer)) == np.float32(s["betas"][n])
        new, _ = model.hot_path(data, timer, exp_noise=s["noise"])
        assert np.array_equal(data["agent"].is_infected.detach().cpu().numpy(), s["is_infected"]), i
        assert data["agent"].is_infected.requires_grad
        series.append(data["agent"].is_infected.sum())
    params = [model.infection_networks.networks[n].log_beta for n in names]
    for tag, loss in (("last", series[-1]), ("series", torch.stack(series).sum())):
        grads = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
        for n, g in zip(names, grads):
            got = 0.0 if g is None else float(g)
            ref = float(sub[f"grad_{tag}/{n}"])
            assert got == pytest.approx(ref, rel=2e-5, abs=1e-7), (tag, n, got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_kept_forward_sums_equal_the_recomputation(device, case, monkeypatch):
    """A differentiable step keeps the forward's per-agent sums (gj_step_io.agent_sums) and per-venue sums for its
    backward instead of recomputing the two sparse passes (autograd.KEEP_FORWARD_SUMS).  The kept values ARE what the
    recomputation produces - the same kernels on the same inputs, exact sums - so both forms give the same gradients
    bit for bit, through several chained steps, a quarantine policy and the leisure networks (case g2)."""
    import grad_june_amd as G
    from grad_june_amd import autograd as AG

    sub, world, tables, names = load_case(case)

    def run(keep):
        monkeypatch.setattr(AG, "KEEP_FORWARD_SUMS", keep)
        model, timer = _model_and_timer(G, case, device)
        data = _hetero(G, sub, world, device)
        for n in names:
            net = model.infection_networks.networks[n]
            net.log_beta = torch.nn.Parameter(net.log_beta.detach().clone())
        series = []
        for i in range(int(sub["n_steps"])):
            s = step_info(sub, i)
            next(timer)
            data["agent"].symptoms["current_stage"] = s["stage"].to(device)
            model.hot_path(data, timer, exp_noise=s["noise"])
            series.append(data["agent"].is_infected.sum())
        params = [model.infection_networks.networks[n].log_beta for n in names]
        g_last = torch.autograd.grad(series[-1], params, retain_graph=True, allow_unused=True)
        g_all = torch.autograd.grad(torch.stack(series).sum(), params, allow_unused=True)
        return [None if g is None else g.detach().cpu() for g in (*g_last, *g_all)]

    kept, recomputed = run(True), run(False)
    assert any(g is not None and float(g) != 0.0 for g in kept)
    for a, b in zip(kept, recomputed):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b), (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("factor", [1e6, 1e-9, 3e12], ids=["x1e6", "x1e-9", "x3e12"])
def test_hip_backward_is_linear_in_the_loss_scale(device, factor):
    """The backward's sparse passes sum in fixed point with scales chosen for the forward's transmissions; the
    cotangents they carry have whatever size the user's loss has (MSE on case counts: 1e5; a normalised loss: 1e-10).
    They are renormalised by a power of two on the way in and out, so d(c * loss) == c * d(loss) to fp32 rounding,
    far outside the fixed-point window - nothing clipped to the window's edge, nothing flushed to zero."""
    import grad_june_amd as G

    sub, world, tables, names = load_case("g2")
    model, timer = _model_and_timer(G, "g2", device)
    data = _hetero(G, sub, world, device)
    for n in names:
        net = model.infection_networks.networks[n]
        net.log_beta = torch.nn.Parameter(net.log_beta.detach().clone())
    series = []
    for i in range(int(sub["n_steps"])):
        s = step_info(sub, i)
        next(timer)
        data["agent"].symptoms["current_stage"] = s["stage"].to(device)
        model.hot_path(data, timer, exp_noise=s["noise"])
        series.append(data["agent"].is_infected.sum())
    params = [model.infection_networks.networks[n].log_beta for n in names]
    loss = torch.stack(series).sum()
    base = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
    scaled = torch.autograd.grad(loss * factor, params, retain_graph=True, allow_unused=True)
    assert any(g is not None and float(g) != 0.0 for g in base)
    for n, g0, g1 in zip(names, base, scaled):
        a, b = (0.0 if g0 is None else float(g0)), (0.0 if g1 is None else float(g1))
        assert np.isfinite(b)
        assert b == pytest.approx(a * factor, rel=1e-4, abs=1e-6 * factor), (n, a, b)


@pytest.mark.gpu
def test_gradient_is_zero_for_network_not_attended(device):
    """test_model.py:76-143: an agent infected at school carries no gradient to the company network."""
    import grad_june_amd as G
    from grad_june_amd import infection_networks as inw

    sub, world, tables, names = load_case("g1")
    data = _hetero(G, sub, world, device)
    # decoupled halves: agents 0-49 only in one school, 50-99 only in one company (test_model.py:55-74)
    for name, lo in (("school", 0), ("company", 50)):
        data[name].id = torch.tensor([0])
        data[name].people = torch.tensor([50], device=device)
        data["agent", "attends_" + name, name].edge_index = torch.vstack(
            (torch.arange(lo, lo + 50), torch.zeros(50, dtype=torch.long))).to(device)
    nets = G.InfectionNetworks(device=device, company=inw.CompanyNetwork(torch.nn.Parameter(torch.tensor(0.5)), device),
                               school=inw.SchoolNetwork(torch.nn.Parameter(torch.tensor(0.5)), device))
    model = G.GradJune(infection_networks=nets, policies=G.Policies.from_policy_list([]), device=device)
    timer = G.Timer(initial_day="2022-02-01", total_days=10, weekday_step_duration=(24,), weekend_step_duration=(24,),
                    weekday_activities=(("company", "school"),), weekend_activities=(("company", "school"),))
    torch.manual_seed(2)
    seed = data["agent"].is_infected.clone()
    for _ in range(4):
        next(timer)
        model.hot_path(data, timer)
    cases = data["agent"].is_infected
    k = [i for i in range(50) if cases[i] == 1.0 and seed[i] == 0.0]
    assert k, "nobody infected at school"
    gs, gc = torch.autograd.grad(cases[k[0]], [nets["school"].log_beta, nets["company"].log_beta], allow_unused=True)
    assert gs is not None and gs != 0.0
    assert gc is None or gc == 0.0


@pytest.mark.gpu
def test_hip_symptoms_adjoint_matches_oracle(device):
    """gj_adjoint_symptoms against the hand-written CPU adjoint, with injected and with Philox randomness."""
    import ctypes as C

    import grad_june_amd as G
    from grad_june_amd import _native as N
    from grad_june_amd.defaults import default_parameters

    g = torch.Generator().manual_seed(11)
    n = 50_000
    upd = G.SymptomsUpdater.from_parameters(default_parameters(str(device)))
    n_stages = len(upd.symptoms_sampler.stages)
    cur = torch.randint(0, n_stages, (n,), generator=g).float()
    nxt = torch.randint(0, n_stages, (n,), generator=g).float()
    ttn = torch.rand(n, generator=g) * 10
    new = (torch.rand(n, generator=g) < 0.2).float()
    age = torch.randint(0, 100, (n,), generator=g)
    g_cur, g_nxt, g_ttn = (torch.randn(n, generator=g) for _ in range(3))
    for inject in (True, False):
        d = G.HeteroData()
        d["agent"].age, d["agent"].sex = age.to(device), torch.zeros(n, dtype=torch.long, device=device)
        c0, x0, t0 = (t.clone().to(device).requires_grad_() for t in (cur, nxt, ttn))
        d["agent"].symptoms = {"current_stage": c0, "next_stage": x0, "time_to_next_stage": t0}
        nw = new.clone().to(device).requires_grad_()
        if inject:
            progresses = (torch.rand(n, generator=g) < 0.5).float()
            dwell = torch.rand(n, generator=g) * 5
            sym = upd(d, type("T", (), {"now": 5.0})(), nw, progresses=progresses, dwell=dwell)
        else:
            sym = upd(d, type("T", (), {"now": 5.0})(), nw)
            # the branch and the dwell time of the kernel's Philox draw, read off the forward result
            x1 = nxt + new * (2.0 - nxt)
            t1 = ttn + new * (5.0 - ttn)
            progresses = (sym["next_stage"].detach().cpu() == x1 + 1).float()
            dwell = sym["time_to_next_stage"].detach().cpu() - t1
        assert all(sym[k].requires_grad for k in sym)                      # test_symptoms.py:225-231
        loss = ((sym["current_stage"] * g_cur.to(device)).sum() + (sym["next_stage"] * g_nxt.to(device)).sum()
                + (sym["time_to_next_stage"] * g_ttn.to(device)).sum())
        got = torch.autograd.grad(loss, (c0, x0, t0, nw))
        ref = O.adjoint_symptoms(cur, nxt, ttn, new, 5.0, n_stages, progresses, g_cur, g_nxt, g_ttn, dwell)
        for a, b, what in zip(got, ref, ("current_stage", "next_stage", "time_to_next_stage", "new_infected")):
            assert torch.allclose(a.cpu(), b.float(), rtol=1e-4, atol=1e-4), (inject, what)


@pytest.mark.gpu
def test_hip_backward_through_symptoms_matches_reference(device):
    """g3 on the GPU: hot path + symptoms as autograd nodes for 12 steps, the reference's randomness injected;
    the stage-dependent losses (occupancy sums in the deaths series' form, a random linear form of the final
    stages) give the reference's gradients w.r.t. every log_beta."""
    import grad_june_amd as G
    from grad_june_amd.defaults import default_parameters

    sub, world, tables, names = load_case("g3")
    params = default_parameters(str(device))
    for n in params["networks"]:
        params["networks"][n]["log_beta"] += 0.7
    model, timer = G.GradJune.from_parameters(params), G.Timer.from_parameters(params)
    data = _hetero(G, sub, world, device)
    data["agent"].symptoms = {k: torch.from_numpy(sub["state0/sym/" + k]).to(device)
                              for k in ("current_stage", "next_stage", "time_to_next_stage")}
    for n in names:
        net = model.infection_networks.networks[n]
        net.log_beta = torch.nn.Parameter(net.log_beta.detach().clone())
    n_stages = sub["sym_table"].shape[0]
    occupancy = {k: [] for k in range(2, n_stages)}
    for i in range(int(sub["n_steps"])):
        s = step_info(sub, i)
        next(timer)
        assert timer.now == s["now"]
        new, _ = model.hot_path(data, timer, exp_noise=s["noise"])
        assert np.array_equal(data["agent"].is_infected.detach().cpu().numpy(), s["is_infected"]), i
        sym = model.symptoms_updater(data, timer, new, progresses=torch.from_numpy(sub[f"step{i}/sym/progresses"]),
                                     dwell=torch.from_numpy(sub[f"step{i}/sym/dwell"]))
        for k in ("current_stage", "next_stage", "time_to_next_stage"):
            assert np.array_equal(sym[k].detach().cpu().numpy(), sub[f"step{i}/sym/post/{k}"]), (i, k)
        assert sym["current_stage"].requires_grad and sym["time_to_next_stage"].requires_grad
        for k in occupancy:
            occupancy[k].append(occupancy_of(sym["current_stage"], k))
    plist = [model.infection_networks.networks[n].log_beta for n in names]
    for tag, loss in symptom_losses(sub, data["agent"].symptoms, occupancy).items():
        assert float(loss.detach()) == pytest.approx(float(sub["loss_" + tag]), rel=1e-6)
        grads = torch.autograd.grad(loss, plist, retain_graph=True, allow_unused=True)
        for n, g in zip(names, grads):
            got = 0.0 if g is None else float(g)
            ref = float(sub[f"grad_{tag}/{n}"])
            assert got == pytest.approx(ref, rel=1e-4, abs=1e-6), (tag, n, got, ref)
