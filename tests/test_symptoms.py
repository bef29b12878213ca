"""Row f1: disease-stage progression.  CPU: the oracle restatement against every recorded step of
the reference trajectories (randomness injected).  GPU: the fused kernel against the same records,
its Philox mode statistically, and the API-level behaviours of the reference's test_symptoms.py."""
import numpy as np
import pytest
import torch

import gj_oracle as O
import gj_testlib as L

TRAJ = ["june769.npz", "june769_hot.npz", "synth10k.npz"]


def sym_inputs(npz, rec):
    pre = {k[8:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("sym_pre/")}
    post = {k[9:]: v for k, v in rec.items() if k.startswith("sym_post/")}
    return pre, post


@pytest.mark.parametrize("name", TRAJ)
def test_oracle_symptoms_match_reference(name):
    npz = L.load_npz(name)
    age = torch.from_numpy(npz["world/age"])
    n_stages = npz["sym_table"].shape[0]
    n_updates = 0
    for i in range(int(npz["n_steps"])):
        rec = L.step_record(npz, f"step{i}/")
        pre, post = sym_inputs(npz, rec)
        prob = O.symptoms_progress_probability(torch.from_numpy(npz["sym_table"]), age, pre["current_stage"],
                                               pre["next_stage"], pre["time_to_next_stage"],
                                               torch.from_numpy(rec["new_infected"]), float(rec["now"]), n_stages)
        assert np.array_equal(prob.numpy(), rec["sym/prob"])
        got = O.symptoms_update(age, pre["current_stage"], pre["next_stage"], pre["time_to_next_stage"],
                                torch.from_numpy(rec["new_infected"]), float(rec["now"]), n_stages,
                                torch.from_numpy(rec["sym/progresses"]), torch.from_numpy(rec["sym/dwell"]))
        for k, g in zip(("current_stage", "next_stage", "time_to_next_stage"), got):
            assert np.array_equal(g.numpy().astype(np.float32), post[k]), (i, k)
        n_updates += int((post["current_stage"] != pre["current_stage"].numpy()).sum())
    assert n_updates > 0


def _updater(device):
    from grad_june_amd.defaults import default_parameters
    from grad_june_amd.symptoms import SymptomsUpdater

    return SymptomsUpdater.from_parameters(default_parameters(str(device)))


class _T:
    def __init__(self, now):
        self.now = now


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRAJ)
def test_kernel_symptoms_match_reference(device, name):
    """Injected randomness: the kernel reproduces the reference's three symptom arrays exactly."""
    import grad_june_amd as G

    npz = L.load_npz(name)
    upd = _updater(device)
    assert np.array_equal(upd.symptoms_sampler.stage_transition_probabilities.cpu().numpy(), npz["sym_table"])
    d = G.HeteroData()
    d["agent"].age = torch.from_numpy(npz["world/age"]).to(device)
    d["agent"].sex = torch.from_numpy(npz["world/sex"]).to(device)
    for i in range(int(npz["n_steps"])):
        rec = L.step_record(npz, f"step{i}/")
        pre, post = sym_inputs(npz, rec)
        d["agent"].symptoms = {k: v.to(device) for k, v in pre.items()}
        upd(d, _T(float(rec["now"])), torch.from_numpy(rec["new_infected"]).to(device),
            progresses=torch.from_numpy(rec["sym/progresses"]), dwell=torch.from_numpy(rec["sym/dwell"]))
        assert upd.used_kernel
        for k in post:
            assert np.array_equal(d["agent"].symptoms[k].cpu().numpy(), post[k]), (i, k)


@pytest.mark.gpu
def test_kernel_symptoms_statistics(device):
    """Philox mode (reference test_symptoms.py style): everyone exposed and due -> all move to
    'infectious' candidates; progress fraction ~ table value; dwell times ~ LogNormal means."""
    import grad_june_amd as G

    n = 200_000
    upd = _updater(device)
    d = G.HeteroData()
    d["agent"].age = torch.full((n,), 45, device=device)
    d["agent"].sex = torch.zeros(n, dtype=torch.long, device=device)
    d["agent"].symptoms = {"current_stage": torch.full((n,), 2.0, device=device),      # exposed
                           "next_stage": torch.full((n,), 3.0, device=device),          # -> infectious
                           "time_to_next_stage": torch.zeros(n, device=device)}
    upd(d, _T(1.0), torch.zeros(n, device=device))
    s = d["agent"].symptoms
    assert (s["current_stage"] == 3).all()
    onward = (s["next_stage"] == 4).float().mean().item()            # infectious -> symptomatic with p=0.7 at age 45
    assert abs(onward - 0.7) < 0.01
    assert ((s["next_stage"] == 4) | (s["next_stage"] == 0)).all()
    sp = upd.symptoms_sampler
    t_on = s["time_to_next_stage"][s["next_stage"] == 4].mean().item()
    t_rec = s["time_to_next_stage"][s["next_stage"] == 0].mean().item()
    assert abs(t_on - sp.stage_transition_times[3].mean.item()) / sp.stage_transition_times[3].mean.item() < 0.03
    assert abs(t_rec - sp.recovery_times[3].mean.item()) / sp.recovery_times[3].mean.item() < 0.03
    # nobody moves before the due time; susceptibles never move
    d["agent"].symptoms = {"current_stage": torch.ones(n, device=device), "next_stage": torch.ones(n, device=device),
                           "time_to_next_stage": torch.zeros(n, device=device)}
    upd(d, _T(5.0), torch.zeros(n, device=device))
    assert (d["agent"].symptoms["current_stage"] == 1).all() and (d["agent"].symptoms["next_stage"] == 1).all()
    # newly infected: exposed next, due now
    new = torch.zeros(n, device=device)
    new[::2] = 1.0
    upd(d, _T(6.0), new)
    s = d["agent"].symptoms
    assert (s["current_stage"][::2] == 2).all() and (s["current_stage"][1::2] == 1).all()
    assert (s["time_to_next_stage"][::2] > 6.0).all()


@pytest.mark.gpu
def test_kernel_agrees_in_distribution_with_the_restatement(device):
    """The fused kernel with its own (Philox) randomness against the CPU restatement of symptoms.py:82-128 fed
    torch.bernoulli / rsample draws (oracle/gj_oracle.py:symptoms_update): same share of agents progressing, same mean
    dwell time.  The reference's SymptomsSampler.sample_next_stage entry point runs on the same kernel."""
    import grad_june_amd as G
    import gj_oracle as O

    torch.manual_seed(3)
    n = 100_000
    upd = _updater(device)
    sp = upd.symptoms_sampler
    age = torch.arange(n) % 100
    cur0, nxt0, ttn0 = torch.full((n,), 3.0), torch.full((n,), 4.0), torch.zeros(n)
    # restatement on the CPU: the agents move to stage 4 and draw from its table / distributions
    table = sp.stage_transition_probabilities.cpu()
    progresses = torch.bernoulli(table[4, age])
    dwell = torch.where(progresses > 0, sp.stage_transition_times[4].rsample((n,)).cpu(), sp.recovery_times[4].rsample((n,)).cpu())
    _, ref_next, ref_ttn = O.symptoms_update(age, cur0, nxt0, ttn0, torch.zeros(n), 2.0, len(sp.stages), progresses, dwell)
    d = G.HeteroData()
    d["agent"].age = age.to(device)
    d["agent"].sex = torch.zeros(n, dtype=torch.long, device=device)
    d["agent"].symptoms = {"current_stage": cur0.to(device), "next_stage": nxt0.to(device), "time_to_next_stage": ttn0.to(device)}
    upd(d, _T(2.0), torch.zeros(n, device=device))
    s = d["agent"].symptoms
    direct = sp.sample_next_stage(age.to(device), cur0.to(device), nxt0.to(device), ttn0.to(device), 2.0)
    for got_next, got_ttn in ((s["next_stage"], s["time_to_next_stage"]), (direct[1], direct[2])):
        assert abs((got_next == 5).float().mean().item() - (ref_next == 5).float().mean().item()) < 0.01
        assert abs(got_ttn.mean().item() - ref_ttn.mean().item()) / ref_ttn.mean().item() < 0.03
    assert (direct[0] == 4).all() and (s["current_stage"] == 4).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["june769.npz", "june769_hot.npz"])
def test_full_timestep_chain_matches_reference(device, name):
    """The whole GradJune timestep (hot path + symptoms kernel) carried forward for 15 steps on the GPU,
    nothing teacher-forced: with the reference's randomness injected (sampler noise, bernoulli outcomes,
    dwell-time samples) every state array equals the reference's after every step - including the
    quarantine mask that the symptoms state feeds back into the hot path."""
    import json

    import grad_june_amd as G
    from test_host_logic import _cpu

    npz = L.load_npz(name)
    world = L.world_from(npz)
    params = _cpu(json.loads(str(npz["params_json"])))
    params["system"]["device"] = str(device)
    model = G.GradJune.from_parameters(params)
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
    with torch.no_grad():
        for i in range(int(npz["n_steps"])):
            rec = L.step_record(npz, f"step{i}/")
            next(timer)
            assert timer.now == float(rec["now"])
            assert np.array_equal(ag.symptoms["current_stage"].cpu().numpy(), rec["pre/current_stage"].astype(np.float32)), i
            new, _ = model.hot_path(d, timer, exp_noise=torch.from_numpy(rec["exp_noise"]))
            for k in ("susceptibility", "is_infected", "infection_time"):
                assert np.array_equal(ag[k].cpu().numpy(), rec["post/" + k]), (i, k)
            model.symptoms_updater(d, timer, new, progresses=torch.from_numpy(rec["sym/progresses"]),
                                   dwell=torch.from_numpy(rec["sym/dwell"]))
            for k in ("current_stage", "next_stage", "time_to_next_stage"):
                assert np.array_equal(ag.symptoms[k].cpu().numpy(), rec["sym_post/" + k]), (i, k)
    assert float(ag.is_infected.sum()) == float(npz["cases_per_timestep"][-1])


@pytest.mark.gpu
def test_symptoms_differentiable(device):
    """test_symptoms.py:208-231 of the reference: new_infected from a hard Gumbel-softmax on a parameter,
    mortality raised to 1, 100 further steps - the dead agents' stage carries a gradient back to beta and
    all three symptom arrays stay on the autograd graph."""
    import grad_june_amd as G
    from grad_june_amd.defaults import default_parameters

    torch.manual_seed(0)
    params = default_parameters(str(device))
    su = G.SymptomsUpdater.from_parameters(params)
    su.symptoms_sampler.stage_transition_probabilities[2:, :] = 1.0
    timer = G.Timer.from_parameters(params)
    n = 100
    d = G.HeteroData()
    d["agent"].id = torch.arange(n, device=device)
    d["agent"].age = torch.randint(0, 100, (n,), device=device)
    d["agent"].sex = torch.zeros(n, dtype=torch.long, device=device)
    d["agent"].symptoms = {"current_stage": torch.ones(n, device=device), "next_stage": torch.ones(n, device=device),
                           "time_to_next_stage": torch.zeros(n, device=device)}
    beta = torch.nn.Parameter(torch.tensor(10.0, device=device))
    probs = 1 - torch.exp(-beta) * torch.ones(n, device=device)
    new_infected = torch.nn.functional.gumbel_softmax(probs, tau=0.1, hard=True)
    symptoms = su(data=d, timer=timer, new_infected=new_infected)
    for _ in range(100):
        next(timer)
        symptoms = su(data=d, timer=timer, new_infected=torch.zeros(n, device=device))
    dead = float(su.stages_ids[-1])
    deaths = symptoms["current_stage"][symptoms["current_stage"] == dead]
    assert len(deaths) > 0
    deaths.sum().backward()
    assert deaths.requires_grad
    assert all(symptoms[k].requires_grad for k in ("current_stage", "next_stage", "time_to_next_stage"))
    assert beta.grad is not None and torch.isfinite(beta.grad)
