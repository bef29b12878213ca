"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run once, in the build container (the only place /root/reference exists):

    python tests/golden/make_golden.py

The reference's own .py files are imported unmodified from /root/reference with the two
absent third-party modules stood in (see _pyg_standin.py for exactly what is stood in and
what that means for these vectors).  Every case records INPUTS (graph, agent attributes,
state before the step, step scalars, the Exponential(1) noise the sampler drew) and
OUTPUTS per stage of the path.  At generation time the CPU
oracle (oracle/gj_oracle.py) is asserted equal to the reference on every case.

Files written (data only - no reference source text):
    kat6.npz          6-agent / 2-school graph of test/unit/infection_networks/test_base.py:22-44
    c100.npz          100-agent graph of test/conftest.py:36-89 (seed 999), several policy variants
    june769.npz       test/data/data.pkl + configs/default.yaml: 15-step trajectory, 11 networks
    june769_hot.npz   the same with every log_beta raised by 0.9 (a real epidemic wave)
    synth10k.npz      10k-agent synthetic with degree-0/1 venues, a 5k-agent venue, duplicates
    world769.npz      test/data/data.pkl as neutral arrays (graph + agent attributes)
    june769_series.npz  the reference Runner's own result series (cases, daily cases, cases by age bin, deaths) over a
                      90-day run of the shipped world + the per-agent state after every step (row f2)
    grads.npz         d(cases)/d(log_beta) of every network through 4 / 6 timesteps (autograd of the
                      reference), with the noise of every step, on the 100- and 769-agent worlds
    default_params.json   yaml.safe_load(configs/default.yaml) (dates stringified)
"""
from __future__ import annotations

import copy
import datetime
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import _pyg_standin  # noqa: E402

grad_june = _pyg_standin.import_reference()

import torch  # noqa: E402
import yaml  # noqa: E402
from torch_geometric.data import HeteroData  # noqa: E402  (stand-in container)
import torch_geometric.transforms as T  # noqa: E402

import gj_oracle as O  # noqa: E402

from grad_june import GradJune, Timer, Runner  # noqa: E402
from grad_june.infection import infect_people_at_indices  # noqa: E402
from grad_june.infection_networks import InfectionNetworks  # noqa: E402
from grad_june.infection_networks.base import (  # noqa: E402
    CompanyNetwork,
    HouseholdNetwork,
    SchoolNetwork,
)
from grad_june.paths import default_config_path  # noqa: E402
from grad_june.policies import Policies, Quarantine, SocialDistancing, CloseVenue  # noqa: E402
from grad_june.transmission import TransmissionSampler  # noqa: E402

EDGE_SETS = ["household", "company", "school", "university", "care_home", "leisure"]


def seed_all(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)


# ------------------------------------------------------------------------------------------
# neutral views of a reference HeteroData
# ------------------------------------------------------------------------------------------
def world_of(data):
    A = len(data["agent"].id)
    w = {"n_agents": A, "edge_sets": {}}
    if "age" in data["agent"]:
        w["age"] = data["agent"].age.clone()
        w["sex"] = data["agent"].sex.clone()
    else:
        w["age"] = torch.zeros(A, dtype=torch.long)
        w["sex"] = torch.zeros(A, dtype=torch.long)
    for s in EDGE_SETS:
        key = ("agent", "attends_" + s, s)
        if key not in data:
            continue
        ei = data[key].edge_index
        people = data[s]["people"]
        if not isinstance(people, torch.Tensor):
            people = torch.as_tensor(np.asarray(people))
        w["edge_sets"][s] = {"agent": ei[0].clone(), "venue": ei[1].clone(), "people": people.clone()}
    return w


def state_of(data):
    ag = data["agent"]
    st = {k: ag["infection_parameters"][k].detach().clone() for k in ("max_infectiousness", "shape", "rate", "shift")}
    for k in ("infection_time", "is_infected", "susceptibility"):
        st[k] = ag[k].detach().clone()
    st["current_stage"] = ag["symptoms"]["current_stage"].detach().clone()
    return st


def flat_world(w, out, prefix="world/"):
    out[prefix + "n_agents"] = np.int64(w["n_agents"])
    out[prefix + "age"] = w["age"].numpy()
    out[prefix + "sex"] = w["sex"].numpy()
    for s, es in w["edge_sets"].items():
        out[f"{prefix}es/{s}/agent"] = es["agent"].numpy()
        out[f"{prefix}es/{s}/venue"] = es["venue"].numpy()
        out[f"{prefix}es/{s}/people"] = es["people"].numpy()


def tables_of(model):
    tabs = {}
    for name, net in model.infection_networks.networks.items():
        if hasattr(net, "leisure_probabilities"):
            tabs[name] = net.leisure_probabilities.detach().clone()
    return tabs


# ------------------------------------------------------------------------------------------
# one instrumented step: same call sequence as GradJune.forward (model.py:112-144)
# ------------------------------------------------------------------------------------------
def capture_step(model, data, timer, *, run_symptoms=True):
    rec = {}
    pre = state_of(data)
    for k, v in pre.items():
        rec["pre/" + k] = v.numpy().copy()
    policies = model.policies
    nets = model.infection_networks
    rec["now"] = np.float64(timer.now)
    rec["dt"] = np.float64(timer.duration)
    rec["day_type"] = np.int64(0 if timer.day_type == "weekday" else 1)

    with torch.no_grad():
        data["agent"].transmission = model.transmission_updater(data=data, timer=timer)
        rec["transmission"] = data["agent"].transmission.numpy().copy()
        not_inf = nets(data=data, timer=timer, policies=policies)
        rec["not_infected_probs"] = not_inf.numpy().copy()

        # what the container call used: order, betas, quarantine thresholds / mask
        order = timer.get_activity_order()
        if policies.close_venue_policies:
            order = policies.close_venue_policies.apply(edge_types=order, timer=timer)
        rec["active"] = np.array(",".join(order))
        if policies.quarantine_policies:
            qm = policies.quarantine_policies.quarantine_mask
            rec["qmask"] = qm.numpy().copy()
            thr = [
                (float(p.stage_threshold) if p.is_active(timer.date) else np.nan)
                for p in policies.quarantine_policies.policies
            ]
            rec["q_thresholds"] = np.array(thr, dtype=np.float64)
            rec["has_quarantine"] = np.int64(1)
        else:
            rec["has_quarantine"] = np.int64(0)
            rec["q_thresholds"] = np.zeros(0)
        for name in order:
            net = nets[name]
            beta_v = net._get_beta(policies=policies, timer=timer, data=data)
            rec["beta/" + name] = np.float32(beta_v[0].item()) if len(beta_v) else np.float32(
                (10.0 ** net.log_beta).item()
            )
            people = net._get_people_per_group(data)
            pc = torch.maximum(torch.minimum(1.0 / (people - 1), torch.tensor(1.0)), torch.tensor(0.0))
            trans = net._get_transmissions(data=data, policies=policies, timer=timer)
            cum = net.propagate(net._get_edge_index(data), x=trans, y=beta_v * pc)
            rec["cum/" + name] = cum.numpy().copy()
            rec["ts/" + name] = net(data=data, timer=timer, policies=policies).numpy().copy()

        rng = torch.get_rng_state()
        new_inf = model.is_infected_sampler(not_inf)
        after = torch.get_rng_state()
        torch.set_rng_state(rng)
        noise = torch.empty(2, not_inf.shape[0]).exponential_()
        assert torch.equal(torch.get_rng_state(), after), "sampler drew something else"
        rec["exp_noise"] = noise.numpy().copy()
        rec["new_infected"] = new_inf.numpy().copy()
        assert torch.equal(O.sample_infected(not_inf, noise), new_inf)

        model.infect_people(data, timer, new_inf)
        for k in ("susceptibility", "is_infected", "infection_time"):
            rec["post/" + k] = data["agent"][k].numpy().copy()
        if run_symptoms:
            record_symptoms(model, data, timer, new_inf, rec)
    return rec, pre


class _RecordingDist:
    """Proxy of a torch distribution that remembers what rsample returned (same RNG consumption)."""

    def __init__(self, dist, log, tag):
        self.dist, self.log, self.tag = dist, log, tag

    def rsample(self, shape):
        out = self.dist.rsample(shape)
        self.log.append((self.tag, out.clone()))
        return out


def record_symptoms(model, data, timer, new_inf, rec):
    """Run the reference's SymptomsUpdater with its randomness recorded: the bernoulli outcome and,
    per agent, the dwell-time sample it consumed (row f1 of SURVEY section 8)."""
    sampler = model.symptoms_updater.symptoms_sampler
    sym = data["agent"].symptoms
    for k in ("current_stage", "next_stage", "time_to_next_stage"):
        rec["sym_pre/" + k] = sym[k].numpy().astype(np.float32)
    n_stages = len(sampler.stages)
    age = data["agent"].age
    probs = O.symptoms_progress_probability(
        sampler.stage_transition_probabilities, age, sym["current_stage"].float(), sym["next_stage"].float(),
        sym["time_to_next_stage"], new_inf, timer.now, n_stages)
    log = []
    saved = (dict(sampler.stage_transition_times), dict(sampler.recovery_times))
    for i in saved[0]:
        if saved[0][i] is not None:
            sampler.stage_transition_times[i] = _RecordingDist(saved[0][i], log, ("next", i))
        if saved[1][i] is not None:
            sampler.recovery_times[i] = _RecordingDist(saved[1][i], log, ("rec", i))
    bern = {}
    real_bernoulli = torch.bernoulli

    def bernoulli(p, *a, **k):
        out = real_bernoulli(p, *a, **k)
        bern["p"], bern["out"] = p.clone(), out.clone()
        return out

    torch.bernoulli = bernoulli
    try:
        pre = {k: sym[k].clone().float() for k in ("current_stage", "next_stage", "time_to_next_stage")}
        model.symptoms_updater(data=data, timer=timer, new_infected=new_inf)
    finally:
        torch.bernoulli = real_bernoulli
        sampler.stage_transition_times.update(saved[0])
        sampler.recovery_times.update(saved[1])
    assert torch.equal(bern["p"], probs), "progress probabilities"
    progresses = bern["out"]
    # the sample each agent consumed: stage after transition selects (kind, i)
    cur_after = sym["current_stage"].float()
    dwell = torch.zeros(len(age))
    for (kind, i), draw in log:
        use = (cur_after == i) & (progresses.bool() if kind == "next" else ~progresses.bool())
        dwell = torch.where(use, draw, dwell)
    rec["sym/progresses"] = progresses.numpy().astype(np.float32)
    rec["sym/dwell"] = dwell.numpy().astype(np.float32)
    rec["sym/prob"] = probs.numpy().astype(np.float32)
    for k in ("current_stage", "next_stage", "time_to_next_stage"):
        rec["sym_post/" + k] = sym[k].numpy().astype(np.float32)
    got = O.symptoms_update(age, pre["current_stage"], pre["next_stage"], pre["time_to_next_stage"], new_inf,
                            timer.now, n_stages, progresses, dwell)
    for k, g in zip(("current_stage", "next_stage", "time_to_next_stage"), got):
        assert torch.equal(g.float(), sym[k].float()), "oracle symptoms != reference at " + k


def check_oracle(rec, pre, world, tables, atol=0.0):
    """Assert that oracle/gj_oracle.py reproduces what the reference just computed."""
    active = str(rec["active"]).split(",") if str(rec["active"]) else []
    betas = {n: float(rec["beta/" + n]) for n in active}
    thr = None
    if int(rec["has_quarantine"]):
        thr = [None if np.isnan(t) else float(t) for t in rec["q_thresholds"]]
    out = O.hot_path_step(
        world, pre, now=float(rec["now"]), delta_time=float(rec["dt"]), day_type=int(rec["day_type"]),
        active=active, betas=betas, leisure_tables=tables, quarantine_thresholds=thr,
        exp_noise=torch.from_numpy(rec["exp_noise"]), return_intermediates=True,
    )

    def same(a, b, what):
        a = a.numpy() if isinstance(a, torch.Tensor) else a
        if atol == 0.0:
            ok = np.array_equal(a, b, equal_nan=True)
        else:
            ok = np.allclose(a, b, rtol=0, atol=atol, equal_nan=True)
        assert ok, f"oracle != reference at {what}: max|d|={np.nanmax(np.abs(a - b))}"

    same(out["transmission"], rec["transmission"], "transmission")
    for n in active:
        same(out["cum_" + n], rec["cum/" + n], "cum/" + n)
        same(out["ts_" + n], rec["ts/" + n], "ts/" + n)
    same(out["not_infected_probs"], rec["not_infected_probs"], "not_infected_probs")
    same(out["new_infected"], rec["new_infected"], "new_infected")
    for k in ("susceptibility", "is_infected", "infection_time"):
        same(out[k], rec["post/" + k], "post/" + k)


def add_steps(out, recs, prefix="step"):
    out["n_steps"] = np.int64(len(recs))
    for i, rec in enumerate(recs):
        for k, v in rec.items():
            out[f"{prefix}{i}/{k}"] = v


def save(name, out):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


# ------------------------------------------------------------------------------------------
# case: kat6  (test_base.py:22-44)
# ------------------------------------------------------------------------------------------
def make_kat6():
    data = HeteroData()
    data["agent"].id = torch.arange(6)
    data["agent"].transmission = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    data["agent"].susceptibility = torch.tensor([1, 2, 3, 0.5, 0.7, 1.0])
    data["school"].id = torch.arange(2)
    data["school"].people = torch.tensor([2, 2])
    data["agent", "attends_school", "school"].edge_index = torch.vstack(
        (torch.arange(6), torch.tensor([0, 0, 0, 1, 1, 1]))
    )
    data = T.ToUndirected()(data)
    nets = InfectionNetworks(school=SchoolNetwork(log_beta=np.log10(2.0)))
    timer = Timer(initial_day="2022-02-01", total_days=10, weekday_step_duration=(24,),
                  weekend_step_duration=(24,), weekday_activities=(("school",),),
                  weekend_activities=(("school",),))
    p = nets(data=data, timer=timer, policies=Policies())
    expected = np.exp(-np.array([1.2, 2.4, 3.6, 1.5, 2.1, 3]))
    assert np.allclose(p.detach().numpy(), expected)
    out = {}
    flat_world(world_of(data), out)
    out["transmission"] = data["agent"].transmission.numpy()
    out["susceptibility"] = data["agent"].susceptibility.numpy()
    out["beta/school"] = np.float32((10.0 ** nets["school"].log_beta).item())
    out["dt"] = np.float64(timer.duration)
    out["not_infected_probs"] = p.detach().numpy()
    out["expected_exponent"] = np.array([1.2, 2.4, 3.6, 1.5, 2.1, 3.0])
    save("kat6.npz", out)


# ------------------------------------------------------------------------------------------
# case: c100  (conftest.py:36-89 under seed 999)
# ------------------------------------------------------------------------------------------
def conftest_data():
    seed_all(999)
    sampler = TransmissionSampler.from_file()
    n = 100
    data = HeteroData()
    data["agent"].id = torch.arange(0, n)
    data["agent"].age = torch.randint(0, 100, (n,))
    data["agent"].sex = torch.randint(0, 2, (n,))
    v = sampler(n)
    data["agent"].infection_parameters = {
        "max_infectiousness": v[0], "shape": v[1], "rate": v[2], "shift": v[3]}
    data["agent"].transmission = torch.zeros(n)
    data["agent"].susceptibility = torch.ones(n)
    data["agent"].is_infected = torch.zeros(n)
    data["agent"].infection_time = torch.zeros(n)
    data["agent"].symptoms = {
        "current_stage": torch.ones(n, dtype=torch.long),
        "next_stage": torch.ones(n, dtype=torch.long),
        "time_to_next_stage": torch.zeros(n)}
    data["school"].id = torch.arange(0, 4)
    data["school"].people = 25 * torch.ones(4)
    data["company"].id = torch.arange(0, 4)
    data["company"].people = 25 * torch.ones(4)
    data["household"].id = torch.arange(0, 25)
    data["household"].people = 4 * torch.ones(25)
    data["agent", "attends_school", "school"].edge_index = torch.vstack(
        (data["agent"].id, torch.tensor(np.repeat(np.arange(0, 4), 25))))
    data["agent", "attends_company", "company"].edge_index = torch.vstack(
        (data["agent"].id, torch.tensor(np.repeat(np.arange(0, 4), 25))))
    data["agent", "attends_household", "household"].edge_index = torch.vstack(
        (data["agent"].id, torch.tensor(np.repeat(np.arange(0, 25), 4))))
    data = T.ToUndirected()(data)
    return infect_people_at_indices(data, list(range(0, 100, 10)))


def make_c100():
    out = {}
    base = conftest_data()
    world = world_of(base)
    flat_world(world, out)
    variants = {}

    def three_nets(lb=0.5):
        return InfectionNetworks(
            household=HouseholdNetwork(log_beta=lb), company=CompanyNetwork(log_beta=lb),
            school=SchoolNetwork(log_beta=lb))

    def tm(acts, day="2022-02-01"):
        return Timer(initial_day=day, total_days=10, weekday_step_duration=(24,),
                     weekend_step_duration=(24,), weekday_activities=(tuple(acts),),
                     weekend_activities=(tuple(acts),))

    # v0: plain model step at t=3 days (test_model.py:25-33 shape)
    variants["plain_t3"] = (three_nets(), Policies.from_policy_list([]), tm(["company", "school", "household"]), 3, 0.0)
    # v1: no policy collections at all (Policies(): quarantine mask is the scalar 1.0)
    variants["nopolicy_t5"] = (three_nets(0.3), Policies(), tm(["company", "household"]), 5, 0.0)
    # v2: quarantine active, everyone at stage 5 >= 3 (test_quarantine_policies.py:40-72)
    q = Quarantine(stage_threshold=3, start_date="2022-02-01", end_date="2022-03-15", device="cpu")
    variants["quarantine"] = (three_nets(3.0), Policies.from_policy_list([q]), tm(["company", "household"]), 2, 1.0)
    # v3: social distancing (test_interaction_policies.py:92-123)
    sd = SocialDistancing(start_date="2022-02-01", end_date="2022-02-25",
                          beta_factors={"school": 0.3, "company": 0.5}, device="cpu")
    variants["distancing"] = (three_nets(0.2), Policies.from_policy_list([sd]), tm(["company", "school"]), 4, 1.0)
    # v4: close venue (test_close_venue_policies.py:46-69)
    cv = CloseVenue(names=("company",), start_date="2022-02-01", end_date="2022-02-25", device="cpu")
    variants["closed"] = (three_nets(1.0), Policies.from_policy_list([cv]), tm(["company", "household"]), 1, 1.0)
    # v5: two quarantine policies, one inactive, mixed stages
    q1 = Quarantine(stage_threshold=4, start_date="2022-02-01", end_date="2022-03-15", device="cpu")
    q2 = Quarantine(stage_threshold=2, start_date="2023-02-01", end_date="2023-03-15", device="cpu")
    variants["quarantine_mixed"] = (three_nets(0.8), Policies.from_policy_list([q1, q2]), tm(["company", "school", "household"]), 6, 0.5)

    names = []
    for vname, (nets, policies, timer, n_next, add_trans) in variants.items():
        seed_all(1000 + len(names))
        data = copy.deepcopy(base)
        if vname == "quarantine":
            data["agent"]["symptoms"]["current_stage"] = 5 * torch.ones(100)
        if vname == "quarantine_mixed":
            data["agent"]["symptoms"]["current_stage"] = torch.randint(0, 8, (100,)).float()
        for _ in range(n_next):
            next(timer)
        model = GradJune(infection_networks=nets, policies=policies)
        if add_trans:
            # the policy integration tests add a constant to transmission; emulate by shifting
            # max_infectiousness so the recorded path stays a1 -> a9 end to end
            data["agent"]["infection_parameters"]["max_infectiousness"] = (
                data["agent"]["infection_parameters"]["max_infectiousness"] * (1.0 + add_trans))
        rec, pre = capture_step(model, data, timer, run_symptoms=False)
        check_oracle(rec, pre, world, tables_of(model))
        for k, v in rec.items():
            out[f"{vname}/{k}"] = v
        names.append(vname)
    out["variants"] = np.array(",".join(names))
    save("c100.npz", out)


# ------------------------------------------------------------------------------------------
# case: june769  (data.pkl + default.yaml; Runner.forward sequence, runner.py:151-183)
# ------------------------------------------------------------------------------------------
def default_params():
    with open(default_config_path) as f:
        return yaml.safe_load(f)


def jsonable(o):
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, (datetime.date, datetime.datetime)):
        return o.strftime("%Y-%m-%d")
    return o


def make_june769(tag="june769", beta_shift=0.0, write_world=True):
    params = default_params()
    for n in params["networks"]:
        params["networks"][n]["log_beta"] = params["networks"][n]["log_beta"] + beta_shift
    # policies that actually bite inside the 15 simulated days
    params["policies"]["quarantine"] = {
        "quarantine": {1: {"start_date": "2022-02-05", "end_date": "2022-02-12", "stage_threshold": 4}}}
    params["policies"]["close_venue"] = {
        "close_venue": {1: {"start_date": "2022-02-10", "end_date": "2022-02-13", "names": ["school", "pub"]}}}
    params["policies"]["interaction"]["social_distancing"][1]["start_date"] = "2022-02-08"
    seed_all(769)
    runner = Runner.from_parameters(params)
    world = world_of(runner.data)
    out = {}
    flat_world(world, out)
    with open(os.path.join(HERE, "default_params.json"), "w") as f:
        json.dump(jsonable(default_params()), f, indent=1)  # key order matters (age-bin parsing)
    out["params_json"] = np.array(json.dumps(jsonable(params)))
    tabs = tables_of(runner.model)
    for n, t in tabs.items():
        out["table/" + n] = t.numpy()
    for n, net in runner.model.infection_networks.networks.items():
        out["log_beta/" + n] = np.float32(net.log_beta.item())
    out["sym_table"] = runner.model.symptoms_updater.symptoms_sampler.stage_transition_probabilities.numpy()

    # Runner.forward, instrumented
    timer, model, data = runner.timer, runner.model, runner.data
    with torch.no_grad():
        timer.reset()
        runner.restore_initial_data()
        runner.set_initial_cases()
    out["seed/is_infected"] = data["agent"].is_infected.numpy().copy()
    out["seed/current_stage"] = data["agent"].symptoms["current_stage"].numpy().astype(np.float32)
    cases = [float(data["agent"].is_infected.sum())]
    # row f2: the Runner's own per-step reductions (runner.py:158-171, 198-224), taken by the reference's methods on
    # the state after every recorded step - cases by age bin (OPEN intervals lo < age < hi) and the deaths series
    by_age = [runner.get_cases_by_age(data).numpy().copy()]
    runner.store_differentiable_deaths(data)
    recs = []
    while timer.date < timer.final_date:
        next(timer)
        rec, pre = capture_step(model, data, timer, run_symptoms=True)
        check_oracle(rec, pre, world, tabs)
        recs.append(rec)
        cases.append(float(data["agent"].is_infected.sum()))
        by_age.append(runner.get_cases_by_age(data).numpy().copy())
        runner.store_differentiable_deaths(data)
    add_steps(out, recs)
    out["cases_per_timestep"] = np.array(cases, dtype=np.float32)
    out["series/age_bins"] = runner.age_bins.numpy().astype(np.int64)
    out["series/cases_by_age"] = np.stack(by_age).astype(np.float32)                     # [T+1, bins]
    out["series/deaths_per_timestep"] = data["results"]["deaths_per_timestep"].detach().numpy().astype(np.float32)
    out["series/daily_cases_per_timestep"] = torch.diff(torch.tensor(cases), prepend=torch.tensor([0.0])).numpy()
    out["series/dead_stage"] = np.int64(model.symptoms_updater.stages_ids[-1])
    print(tag, "cases:", cases)
    save(tag + ".npz", out)
    if not write_world:
        return

    # the world itself as neutral arrays (+ string attributes), for Runner-level plumbing tests
    w = {}
    flat_world(world, w, prefix="")
    w["agent/id"] = np.asarray(data["agent"].id)
    w["agent/ethnicity"] = np.asarray(data["agent"].ethnicity)
    w["agent/area"] = np.asarray(data["agent"].area)
    for s in EDGE_SETS:
        w[f"venue_id/{s}"] = np.asarray(data[s]["id"])
    save("world769.npz", w)


# ------------------------------------------------------------------------------------------
# case: june769_series  (row f2: the reference Runner's OWN result series over a long run)
# ------------------------------------------------------------------------------------------
class _StateRecorder(torch.nn.Module):
    """Wraps the reference model inside the reference Runner: same forward, keeps the state after every step."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        self.is_infected, self.current_stage = [], []

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("inner"), name)

    def forward(self, data, timer):
        data = self.inner(data, timer)
        self.is_infected.append(data["agent"].is_infected.detach().numpy().astype(np.float32).copy())
        self.current_stage.append(data["agent"].symptoms["current_stage"].detach().numpy().astype(np.float32).copy())
        return data


def make_june769_series():
    """Runner.forward() of the reference, unmodified (runner.py:151-183), on the shipped world for 90 days with every
    log_beta raised by 0.9: its results dict (cases_per_timestep, daily_cases_per_timestep, deaths_per_timestep,
    cases_by_age_18/65/100) next to the per-agent state after every step that those series were reduced from."""
    params = default_params()
    params["timer"]["total_days"] = 90
    for n in params["networks"]:
        params["networks"][n]["log_beta"] = params["networks"][n]["log_beta"] + 0.9
    # the default severity table kills ~1 in 10^4 of the infected: make the later stages likely, so that the deaths
    # series of 769 agents is not identically zero
    for stage in ("symptomatic", "severe", "critical"):
        params["symptoms"]["stage_transition_probabilities"][stage] = {"0-100": 0.6}
    seed_all(7690)
    runner = Runner.from_parameters(params)
    rec = _StateRecorder(runner.model)
    runner.model = rec
    with torch.no_grad():
        results, is_inf = runner()
    # the state the first row was reduced from is gone (the loop has run); rows 1.. have their state recorded
    out = {"age": runner.data["agent"].age.numpy().astype(np.int64),
           "sex": runner.data["agent"].sex.numpy().astype(np.int64),
           "age_bins": runner.age_bins.numpy().astype(np.int64),
           "dead_stage": np.int64(runner.model.symptoms_updater.stages_ids[-1]),
           "post/is_infected": np.stack(rec.is_infected), "post/current_stage": np.stack(rec.current_stage),
           "final/is_infected": is_inf.numpy().astype(np.float32)}
    for k, v in results.items():
        if k != "dates":
            out["results/" + k] = v.detach().numpy().astype(np.float32)
    out["results/n_dates"] = np.int64(len(results["dates"]))
    assert out["results/deaths_per_timestep"][-1] > 0, "no deaths in the run: lengthen it"
    assert len(rec.is_infected) == len(results["dates"]) - 1
    print("june769_series deaths:", out["results/deaths_per_timestep"][-1], "cases:", out["results/cases_per_timestep"][-1])
    save("june769_series.npz", out)


# ------------------------------------------------------------------------------------------
# case: synth10k  (edge cases: deg-0/1 venues, people != degree, 5k venue, duplicates, no-edge agents)
# ------------------------------------------------------------------------------------------
def make_synth10k():
    rng = np.random.default_rng(1234)
    A = 10_000
    params = default_params()
    params["policies"]["quarantine"] = {
        "quarantine": {1: {"start_date": "2022-01-01", "end_date": "2023-01-01", "stage_threshold": 4}}}
    params["policies"]["interaction"]["social_distancing"][1]["start_date"] = "2022-01-01"
    seed_all(4321)
    data = HeteroData()
    data["agent"].id = torch.arange(A)
    data["agent"].age = torch.from_numpy(rng.integers(0, 100, A))
    data["agent"].sex = torch.from_numpy(rng.integers(0, 2, A))

    def add_set(name, agents, venues, n_venues, people=None):
        agents = np.asarray(agents, dtype=np.int64)
        venues = np.asarray(venues, dtype=np.int64)
        perm = rng.permutation(len(agents))          # reference COO is unsorted
        ei = torch.from_numpy(np.vstack((agents[perm], venues[perm])))
        data["agent", "attends_" + name, name].edge_index = ei
        data[name].id = torch.arange(n_venues)
        if people is None:
            people = np.bincount(venues, minlength=n_venues)
        data[name].people = torch.from_numpy(np.asarray(people, dtype=np.int64))

    # household: sizes 1..6, plus 50 empty households at the end; some `people` overridden
    sizes = rng.integers(1, 7, 4000)
    cut = np.searchsorted(np.cumsum(sizes), A)
    sizes = sizes[: cut + 1]
    hv = np.repeat(np.arange(len(sizes)), sizes)[:A]
    nh = len(sizes) + 50
    people = np.bincount(hv, minlength=nh)
    people[:5] = [0, 1, 2, 3, 7]                      # people is an independent input
    add_set("household", rng.permutation(A), hv, nh, people)
    # company: lognormal sizes, 30% of agents unemployed
    workers = rng.permutation(A)[: int(0.7 * A)]
    csz = np.maximum(1, rng.lognormal(np.log(20), 1.0, 400).astype(int))
    cv = np.repeat(np.arange(len(csz)), csz)
    cv = cv[rng.permutation(len(cv))][: len(workers)]
    add_set("company", workers[: len(cv)], cv, len(csz))
    # school: ONE 5k-agent venue + 20 smaller ones + a duplicate edge + an empty school
    pupils = rng.permutation(A)[:7000]
    sv = np.concatenate([np.zeros(5000, dtype=np.int64), rng.integers(1, 21, 2000)])
    add_set("school", np.concatenate([pupils, pupils[:3]]), np.concatenate([sv, sv[:3]]), 22)
    # university: 3 venues, 600 agents ; care_home: 10 venues of ~30, a singleton
    add_set("university", rng.permutation(A)[:600], rng.integers(0, 3, 600), 3)
    ch_a = rng.permutation(A)[:301]
    add_set("care_home", ch_a, np.concatenate([rng.integers(0, 10, 300), [10]]), 11)
    # leisure: every agent attends k=3 distinct of 8 super-area nodes (multi-membership)
    la = np.repeat(np.arange(A), 3)
    lv = np.concatenate([rng.permutation(8)[:3] for _ in range(A)])
    add_set("leisure", la, lv, 8)
    data = T.ToUndirected()(data)

    sampler = TransmissionSampler.from_parameters(params)
    v = sampler(A)
    data["agent"].infection_parameters = {
        "max_infectiousness": v[0], "shape": v[1], "rate": v[2], "shift": v[3]}
    inf = torch.from_numpy((rng.random(A) < 0.05).astype(np.float32))
    data["agent"].transmission = torch.zeros(A)
    data["agent"].is_infected = inf.clone()
    data["agent"].susceptibility = 1.0 - inf
    data["agent"].infection_time = torch.from_numpy((-10 * rng.random(A)).astype(np.float32)) * inf
    data["agent"].symptoms = {
        "current_stage": torch.from_numpy(np.where(inf.numpy() > 0, rng.integers(2, 7, A), 1)).long(),
        "next_stage": torch.ones(A, dtype=torch.long),
        "time_to_next_stage": torch.zeros(A)}

    model = GradJune.from_parameters(params)
    timer = Timer.from_parameters(params)
    world = world_of(data)
    out = {}
    flat_world(world, out)
    tabs = tables_of(model)
    for n, t in tabs.items():
        out["table/" + n] = t.numpy()
    out["sym_table"] = model.symptoms_updater.symptoms_sampler.stage_transition_probabilities.numpy()
    recs = []
    for i in range(7):            # 2022-02-02 .. 02-08: five weekdays + a weekend
        next(timer)
        rec, pre = capture_step(model, data, timer, run_symptoms=True)
        check_oracle(rec, pre, world, tabs)
        recs.append(rec)
    add_steps(out, recs)
    save("synth10k.npz", out)


# ------------------------------------------------------------------------------------------
# case: gradients (row f3): d cases / d log_beta through several timesteps, noise recorded
# ------------------------------------------------------------------------------------------
class _NoiseRecorder(torch.nn.Module):
    """Stands where model.is_infected_sampler is; records the Exponential(1) draw of each call."""

    def __init__(self, inner):
        super().__init__()
        self.inner, self.noise = inner, []

    def forward(self, not_infected_probs):
        rng = torch.get_rng_state()
        out = self.inner(not_infected_probs)
        after = torch.get_rng_state()
        torch.set_rng_state(rng)
        self.noise.append(torch.empty(2, not_infected_probs.shape[0]).exponential_())
        torch.set_rng_state(after)
        return out


def run_with_grads(model, data, timer, n_steps, out, prefix, step_first=True):
    rec = _NoiseRecorder(model.is_infected_sampler)
    model.is_infected_sampler = rec
    names = list(model.infection_networks.networks.keys())
    for n in names:
        net = model.infection_networks.networks[n]
        net.log_beta = torch.nn.Parameter(net.log_beta.detach().clone())
    st0 = state_of(data)
    for k, v in st0.items():
        out[f"{prefix}state0/{k}"] = v.numpy().copy()
    cases_series = []
    for i in range(n_steps):
        if step_first:
            next(timer)
        out[f"{prefix}step{i}/current_stage"] = data["agent"]["symptoms"]["current_stage"].detach().numpy().astype(np.float32)
        out[f"{prefix}step{i}/now"] = np.float64(timer.now)
        out[f"{prefix}step{i}/dt"] = np.float64(timer.duration)
        out[f"{prefix}step{i}/day_type"] = np.int64(0 if timer.day_type == "weekday" else 1)
        order = timer.get_activity_order()
        if model.policies.close_venue_policies:
            order = model.policies.close_venue_policies.apply(edge_types=order, timer=timer)
        out[f"{prefix}step{i}/active"] = np.array(",".join(order))
        for n in order:
            net = model.infection_networks[n]
            b = (10.0 ** net.log_beta.detach())
            if model.policies.interaction_policies:
                b = model.policies.interaction_policies.apply(beta=b, name=n, timer=timer)
            out[f"{prefix}step{i}/beta/{n}"] = np.float32(b.item())
        qp = model.policies.quarantine_policies
        thr = [(float(p.stage_threshold) if p.is_active(timer.date) else np.nan) for p in qp.policies] if qp else []
        out[f"{prefix}step{i}/q_thresholds"] = np.array(thr, dtype=np.float64)
        out[f"{prefix}step{i}/has_quarantine"] = np.int64(1 if qp else 0)
        data = model(data=data, timer=timer)
        cases_series.append(data["agent"].is_infected.sum())
        out[f"{prefix}step{i}/exp_noise"] = rec.noise[-1].numpy().copy()
        out[f"{prefix}step{i}/is_infected"] = data["agent"].is_infected.detach().numpy().copy()
        if not step_first:
            next(timer)
    out[prefix + "n_steps"] = np.int64(n_steps)
    out[prefix + "networks"] = np.array(",".join(names))
    # loss 1: infected count after the last step; loss 2: sum of the series (run_model.py:9-11)
    for tag, loss in (("last", cases_series[-1]), ("series", torch.stack(cases_series).sum())):
        for n in names:
            model.infection_networks.networks[n].log_beta.grad = None
        loss.backward(retain_graph=True)
        for n in names:
            g = model.infection_networks.networks[n].log_beta.grad
            out[f"{prefix}grad_{tag}/{n}"] = np.float32(0.0 if g is None else g.item())
        out[f"{prefix}loss_{tag}"] = np.float32(loss.item())
    print(prefix, {n: float(out[f"{prefix}grad_series/{n}"]) for n in names})
    return data


def make_grads():
    out = {}
    # g1: the 100-agent fixture, three networks, 4 steps (test_model.py:34-53 style)
    seed_all(31)
    data = conftest_data()
    flat_world(world_of(data), out, prefix="g1/world/")
    nets = InfectionNetworks(household=HouseholdNetwork(log_beta=0.2), company=CompanyNetwork(log_beta=0.4),
                             school=SchoolNetwork(log_beta=0.3))
    model = GradJune(infection_networks=nets, policies=Policies.from_policy_list([]))
    timer = Timer(initial_day="2022-02-01", total_days=10, weekday_step_duration=(24,), weekend_step_duration=(24,),
                  weekday_activities=(("company", "school", "household"),),
                  weekend_activities=(("company", "school", "household"),))
    next(timer); next(timer)
    run_with_grads(model, data, timer, 4, out, "g1/", step_first=True)
    # g2: the 769-agent world, default parameters (11 networks, leisure tables) + an active quarantine
    # and social distancing window, 6 steps
    params = default_params()
    params["policies"]["quarantine"] = {
        "quarantine": {1: {"start_date": "2022-02-03", "end_date": "2022-02-20", "stage_threshold": 4}}}
    params["policies"]["interaction"]["social_distancing"][1]["start_date"] = "2022-02-04"
    for n in params["networks"]:
        params["networks"][n]["log_beta"] += 0.7
    seed_all(77)
    runner = Runner.from_parameters(params)
    with torch.no_grad():
        runner.timer.reset()
        runner.restore_initial_data()
        runner.set_initial_cases()
    flat_world(world_of(runner.data), out, prefix="g2/world/")
    for n, t in tables_of(runner.model).items():
        out["g2/table/" + n] = t.numpy()
    run_with_grads(runner.model, runner.data, runner.timer, 6, out, "g2/", step_first=True)
    save("grads.npz", out)


class _SymptomsRecorder(torch.nn.Module):
    """Stands where model.symptoms_updater is: lets the reference run on the autograd graph and records,
    per call, the bernoulli outcome and the dwell-time sample each agent consumed (as record_symptoms)."""

    def __init__(self, inner):
        super().__init__()
        self.inner, self.calls = inner, []

    @property
    def stages_ids(self):
        return self.inner.stages_ids

    def forward(self, data, timer, new_infected):
        sampler = self.inner.symptoms_sampler
        sym = data["agent"].symptoms
        rec = {"pre/" + k: sym[k].detach().clone().float().numpy()
               for k in ("current_stage", "next_stage", "time_to_next_stage")}
        log = []
        saved = (dict(sampler.stage_transition_times), dict(sampler.recovery_times))
        for i in saved[0]:
            if saved[0][i] is not None:
                sampler.stage_transition_times[i] = _RecordingDist(saved[0][i], log, ("next", i))
            if saved[1][i] is not None:
                sampler.recovery_times[i] = _RecordingDist(saved[1][i], log, ("rec", i))
        bern = {}
        real_bernoulli = torch.bernoulli

        def bernoulli(p, *a, **k):
            out = real_bernoulli(p, *a, **k)
            bern["out"] = out.detach().clone()
            return out

        torch.bernoulli = bernoulli
        try:
            out = self.inner(data=data, timer=timer, new_infected=new_infected)
        finally:
            torch.bernoulli = real_bernoulli
            sampler.stage_transition_times.update(saved[0])
            sampler.recovery_times.update(saved[1])
        progresses = bern["out"]
        cur_after = sym["current_stage"].detach().float()
        dwell = torch.zeros(len(progresses))
        for (kind, i), draw in log:
            use = (cur_after == i) & (progresses.bool() if kind == "next" else ~progresses.bool())
            dwell = torch.where(use, draw.detach(), dwell)
        rec["progresses"] = progresses.numpy().astype(np.float32)
        rec["dwell"] = dwell.numpy().astype(np.float32)
        for k in ("current_stage", "next_stage", "time_to_next_stage"):
            rec["post/" + k] = sym[k].detach().float().numpy().copy()
        self.calls.append(rec)
        return out


def make_grads_symptoms():
    """g3: gradients that reach log_beta THROUGH the symptoms state machine (the deaths series of
    runner.py:198-215 is such a loss; test_runner.py:82-90) - the whole timestep on the autograd graph."""
    out = {}
    params = default_params()
    for n in params["networks"]:
        params["networks"][n]["log_beta"] += 0.7
    seed_all(123)
    runner = Runner.from_parameters(params)
    with torch.no_grad():
        runner.timer.reset()
        runner.restore_initial_data()
        runner.set_initial_cases()
    model, data, timer = runner.model, runner.data, runner.timer
    flat_world(world_of(data), out, prefix="g3/world/")
    for n, t in tables_of(model).items():
        out["g3/table/" + n] = t.numpy()
    out["g3/sym_table"] = model.symptoms_updater.symptoms_sampler.stage_transition_probabilities.numpy()
    for k in ("current_stage", "next_stage", "time_to_next_stage"):
        out["g3/state0/sym/" + k] = data["agent"].symptoms[k].detach().float().numpy().copy()
    sym_rec = _SymptomsRecorder(model.symptoms_updater)
    model.symptoms_updater = sym_rec
    n_steps = 12
    A = len(data["agent"].id)
    dead = int(sym_rec.stages_ids[-1])
    stage_series = {k: [] for k in range(2, dead + 1)}
    hook = {}

    # run_with_grads drives the steps; sample the occupancy series right after every model call
    real_forward = sym_rec.forward

    def forward_and_sample(data, timer, new_infected):
        res = real_forward(data, timer, new_infected)
        cur = data["agent"].symptoms["current_stage"]
        for k in stage_series:
            stage_series[k].append(((cur == k) * cur / k).sum())
        hook["data"] = data
        return res

    sym_rec.forward = forward_and_sample
    data = run_with_grads(model, data, timer, n_steps, out, "g3/", step_first=True)
    for i, rec in enumerate(sym_rec.calls):
        for k, v in rec.items():
            out[f"g3/step{i}/sym/{k}"] = v
    names = list(model.infection_networks.networks.keys())
    g = torch.Generator().manual_seed(5)
    w_cur, w_nxt = torch.rand(A, generator=g), torch.rand(A, generator=g)
    out["g3/w_cur"], out["g3/w_nxt"] = w_cur.numpy(), w_nxt.numpy()
    sym = data["agent"].symptoms
    losses = {"stage_lin": (w_cur * sym["current_stage"]).sum() + (w_nxt * sym["next_stage"]).sum(),
              "deaths": torch.stack(stage_series[dead]).sum()}
    for k in range(2, dead):
        losses[f"occupancy{k}"] = torch.stack(stage_series[k]).sum()
    for tag, loss in losses.items():
        for n in names:
            model.infection_networks.networks[n].log_beta.grad = None
        if loss.requires_grad:
            loss.backward(retain_graph=True)
        for n in names:
            gr = model.infection_networks.networks[n].log_beta.grad
            out[f"g3/grad_{tag}/{n}"] = np.float32(0.0 if gr is None else gr.item())
        out[f"g3/loss_{tag}"] = np.float32(loss.item())
        print("g3", tag, float(loss), {n: float(out[f"g3/grad_{tag}/{n}"]) for n in names})
    out["g3/loss_tags"] = np.array(",".join(losses))
    save("grads_symptoms.npz", out)


if __name__ == "__main__":
    torch.set_num_threads(1)
    if sys.argv[1:] == ["june769"]:          # only the trajectories of the shipped world
        make_june769()
        make_june769("june769_hot", beta_shift=0.9, write_world=False)
        make_june769_series()
        sys.exit(0)
    if sys.argv[1:] == ["series"]:
        make_june769_series()
        sys.exit(0)
    make_kat6()
    make_c100()
    make_june769()
    make_june769("june769_hot", beta_shift=0.9, write_world=False)
    make_synth10k()
    make_grads()
    make_grads_symptoms()
    make_june769_series()
    print("all golden cases generated; oracle == reference on every recorded stage")
