"""CPU: host mirror of the reference's timer / policies / network descriptors - the scalar logic
that decides what the kernels are launched with.  Checked against what the REFERENCE used on every
recorded step of the golden trajectories (now, dt, day type, active order, beta, thresholds)."""
import datetime
import json

import numpy as np
import pytest
import torch

import gj_testlib as L
from grad_june_amd import _native as N
from grad_june_amd.defaults import default_parameters
from grad_june_amd.infection_networks import InfectionNetworks, SchoolNetwork
from grad_june_amd.policies import CloseVenue, Policies, Quarantine, SocialDistancing
from grad_june_amd.timer import Timer
from grad_june_amd.utils import parse_age_probabilities, read_date


def _cpu(params):
    params = json.loads(json.dumps(params))
    params["system"]["device"] = "cpu"
    for kind in ("weekday", "weekend"):          # json stringifies the shift keys
        params["timer"]["step_duration"][kind] = {int(k): v for k, v in params["timer"]["step_duration"][kind].items()}
        params["timer"]["step_activities"][kind] = {int(k): v for k, v in params["timer"]["step_activities"][kind].items()}
    return params


@pytest.mark.parametrize("name", ["june769.npz", "synth10k.npz"])
def test_step_scalars_match_reference_trajectory(name):
    npz = L.load_npz(name)
    if "params_json" in npz:
        params = _cpu(json.loads(str(npz["params_json"])))
    else:   # synth10k: default parameters + the two overrides make_golden applied
        params = default_parameters("cpu")
        params["policies"]["quarantine"] = {"quarantine": {1: {"start_date": "2022-01-01", "end_date": "2023-01-01", "stage_threshold": 4}}}
        params["policies"]["interaction"]["social_distancing"][1]["start_date"] = "2022-01-01"
    timer = Timer.from_parameters(params)
    policies = Policies.from_parameters(params)
    nets = InfectionNetworks.from_parameters(params)
    for i in range(int(npz["n_steps"])):
        rec = L.step_record(npz, f"step{i}/")
        next(timer)
        assert timer.now == float(rec["now"]) and timer.duration == float(rec["dt"])
        assert (0 if timer.day_type == "weekday" else 1) == int(rec["day_type"])
        active = nets.active_networks(timer, policies)
        assert [n.name for n in active] == str(rec["active"]).split(",")
        for n in active:
            assert np.float32(n.beta_value(policies, timer)) == rec["beta/" + n.name], n.name
        policies.quarantine_policies.apply(symptom_stages=torch.from_numpy(rec["pre/current_stage"]), timer=timer)
        assert policies.quarantine_policies.threshold == L.q_threshold(
            [None if np.isnan(t) else float(t) for t in rec["q_thresholds"]])
        if "qmask" in rec:
            assert np.array_equal(policies.quarantine_policies.quarantine_mask.numpy(), rec["qmask"])


def test_leisure_tables_match_reference():
    npz = L.load_npz("june769.npz")
    nets = InfectionNetworks.from_parameters(default_parameters("cpu"))
    for name, tab in L.tables_from(npz).items():
        assert torch.equal(nets[name].leisure_probabilities, tab), name
        assert nets[name].spec().mask_kind == (N.MASK_QL_AGE75 if name == "care_visit" else N.MASK_QL)
    assert nets["household"].spec().mask_kind == N.MASK_RAW and nets["school"].spec().mask_kind == N.MASK_Q
    assert nets["pub"].edge_set == "leisure" and nets["care_home"].edge_set == "care_home"


def test_timer_shifts_and_weekend():
    """reference test_timer.py semantics: 3 weekday shifts of 8 h, 2 weekend shifts of 12 h."""
    t = Timer(initial_day="2022-02-01", total_days=10, weekday_step_duration=(8, 8, 8),
              weekend_step_duration=(12, 12),
              weekday_activities=(("company", "school", "household"), ("pub", "household"), ("household",)),
              weekend_activities=(("pub",), ("household",)))
    assert t.now == 0 and t.duration == 8 / 24 and t.day_of_week == "Tuesday" and not t.is_weekend
    assert t.get_activity_order() == ["school", "company", "household"]
    next(t)
    assert t.shift == 1 and t.get_activity_order() == ["pub", "household"] and t.now == 8 / 24
    next(t); next(t)
    assert t.shift == 0 and t.date == datetime.datetime(2022, 2, 2) and t.day == 1
    while not t.is_weekend:
        next(t)
    assert t.day_type == "weekend" and t.duration == 0.5 and t.activities == ("pub",)
    t.reset()
    assert t.date == t.initial_date and t.shift == 0
    assert Timer.from_parameters(default_parameters("cpu")).final_date == datetime.datetime(2022, 2, 16)


def test_policies_windows_and_parsing():
    sd = SocialDistancing(start_date="2022-02-01", end_date="2022-02-05", beta_factors={"school": 0.3, "all": 0.8})
    t = Timer(initial_day="2022-02-01", total_days=10, weekday_step_duration=(24,), weekend_step_duration=(24,),
              weekday_activities=(("school",),), weekend_activities=(("school",),))
    assert np.isclose(float(sd.apply(beta=torch.tensor(3.0), name="school", timer=t)), 0.9)
    assert np.isclose(float(sd.apply(beta=torch.tensor(3.0), name="pub", timer=t)), 2.4)
    for _ in range(4):
        next(t)
    assert float(sd.apply(beta=torch.tensor(3.0), name="school", timer=t)) == 3.0
    cv = CloseVenue(names=("company",), start_date="2022-02-01", end_date="2022-02-06")
    assert cv.apply(edge_types=["company", "school"], timer=t) == ["school"]
    q = Quarantine(stage_threshold=3, start_date="2022-02-01", end_date="2022-02-09")
    assert q.apply(timer=t, symptom_stages=torch.tensor([0, 1, 2, 3, 4])).tolist() == [1, 1, 1, 0, 0]
    p = Policies.from_policy_list([sd, cv, q])
    assert len(p.interaction_policies.policies) == 1 and len(p.quarantine_policies.policies) == 1
    assert bool(Policies.from_policy_list([]).quarantine_policies)       # empty collection is truthy
    assert Policies().quarantine_policies is None
    with pytest.raises(ValueError):
        Policies._parse_policy_config({1: {"start_date": "2022-01-01"}}, "quarantine", "cpu")
    assert read_date("2022-03-01") == datetime.datetime(2022, 3, 1)
    with pytest.raises(TypeError):
        read_date(5)


def test_network_naming_and_lookup():
    assert SchoolNetwork._get_name() == "school"
    from grad_june_amd.infection_networks import CareVisitNetwork, CareHomeNetwork

    assert CareVisitNetwork._get_name() == "care_visit" and CareHomeNetwork._get_name() == "care_home"
    nets = InfectionNetworks.from_parameters(default_parameters("cpu"))
    assert set(nets.networks) == set(L.HIERARCHY)
    assert float(nets["household"].log_beta) == pytest.approx(-0.4)
    nets.networks["household"].log_beta = torch.nn.Parameter(nets["household"].log_beta)   # run_model.py:6-8
    assert isinstance(nets["household"].log_beta, torch.nn.Parameter)


def test_age_probabilities():
    assert parse_age_probabilities({"0-50": 0.5, "50-100": 0.2})[49:51] == [0.5, 0.2]
    assert parse_age_probabilities({"10-20": 0.3}, fill_value=7)[9:11] == [7, 0.3]
    cv = parse_age_probabilities({"0-75": 0.0, "75-85": 0.25, "75-100": 0.5})
    assert cv[74] == 0.0 and cv[99] == 0.5


def test_no_cpu_path():
    """The product refuses to compute the infection path off-GPU instead of falling back."""
    from grad_june_amd.world import require_hip

    with pytest.raises(RuntimeError, match="HIP device only"):
        require_hip("cpu")
    from grad_june_amd.graph import HeteroData

    data = HeteroData()
    data["agent"].id = torch.arange(4)
    data["agent"].transmission = torch.zeros(4)
    data["agent"].susceptibility = torch.ones(4)
    nets = InfectionNetworks(device="cpu", school=SchoolNetwork(log_beta=0.0, device="cpu"))
    t = Timer(weekday_activities=(("school",),), weekend_activities=(("school",),), weekday_step_duration=(24,))
    with pytest.raises(RuntimeError, match="HIP device only"):
        nets(data=data, timer=t, policies=Policies())


def test_roofline_accounting_matches_survey_8d():
    """bench.py's algorithmic bytes: B_step = 8*sum_sets E_s + 8*sum_n E_n + 12*sum_n V_n + (8N + 64)*A;
    C3's sizes give the 3.05 GB quoted there, and the per-launch shares add up to it."""
    import bench as B
    from grad_june_amd.synthetic import NETWORKS, algorithmic_bytes, network_edges

    A = 10_000_000
    sizes = {"n_agents": A, "edge_sets": {
        "household": {"n_edges": 15_000_000, "n_venues": 6_000_000}, "care_home": {"n_edges": 15_000_000, "n_venues": 300_000},
        "company": {"n_edges": 15_000_000, "n_venues": 750_000}, "school": {"n_edges": 15_000_000, "n_venues": 30_000},
        "university": {"n_edges": 15_000_000, "n_venues": 7_500}, "leisure": {"n_edges": 15_000_000, "n_venues": 3_000}}}
    nets = NETWORKS["c3"]
    assert network_edges(sizes, nets) == 120_000_000
    b = algorithmic_bytes(sizes, nets)
    want = 8 * 90_000_000 + 8 * 120_000_000 + 12 * (6_000_000 + 300_000 + 750_000 + 30_000 + 7_500 + 3 * 3_000) + (8 * 8 + 64) * A
    assert b == want and abs(b - 3.05e9) < 0.02e9
    world = {"n_agents": A, "edge_sets": {k: {"agent": range(v["n_edges"]), "people": range(v["n_venues"])}
                                          for k, v in sizes["edge_sets"].items()}}
    kb = B.kernel_bytes(world, nets)
    assert kb["transmission"] + kb["tile_scatter"] + kb["tile_venues"] + kb["tile_agents"] == b
    assert kb["tile_venues_B"] + kb["tile_venues_C"] == kb["tile_venues"]
    assert kb["transmission"] + kb["venue_reduce"] + kb["agent_gather"] == b


def test_june_preset_has_the_reference_loaders_membership_structure():
    """synthetic.make_world("june"): what june_world_loader emits for a JUNE world - every person in exactly one
    household, at most one primary activity (school / university / company / care home, by age), leisure venues = super
    areas attended by everybody who lives in one of their k nearest super areas (leisure_loader.py:38-73), ``people`` =
    attendance, no duplicate edge - with the reference's default eleven networks; and a household-major order under which
    every household is a run of consecutive agents."""
    import numpy as np

    from grad_june_amd.synthetic import JUNE_WORLD, SUPER_AREA_AGENTS, make_world, reorder_agents
    from grad_june_amd.timer import activity_hierarchy

    A = 60_000
    w = make_world("june", n_agents=A, seed=5)
    assert sorted(w["networks"], key=activity_hierarchy.index) == w["networks"] and len(w["networks"]) == 11
    deg = {k: np.bincount(es["agent"], minlength=A) for k, es in w["edge_sets"].items()}
    assert (deg["household"] == 1).all()
    primary = deg["school"] + deg["university"] + deg["company"] + deg["care_home"]
    assert primary.max() == 1 and 0.4 < (primary == 1).mean() < 0.6
    age = w["age"]
    assert (age[deg["school"] == 1] < 18).all() and (age[deg["care_home"] == 1] >= 75).all()
    assert ((age >= 5) & (age < 18) & (deg["school"] == 0)).sum() == 0
    for k, es in w["edge_sets"].items():
        assert np.array_equal(np.bincount(es["venue"], minlength=len(es["people"])), es["people"]), k
        key = es["agent"] * len(es["people"]) + es["venue"]
        assert len(np.unique(key)) == len(key), k
    lei = w["edge_sets"]["leisure"]
    n_sa = -(-A // SUPER_AREA_AGENTS)
    assert len(lei["people"]) == n_sa and len(lei["agent"]) == JUNE_WORLD["k_leisure"] * A
    own = lei["venue"] == lei["agent"] // SUPER_AREA_AGENTS
    assert np.bincount(lei["agent"][own], minlength=A).min() == 1          # everybody attends the own super area's venue
    re = reorder_agents(w, by="household")
    hh = re["edge_sets"]["household"]
    v_of = np.empty(A, dtype=np.int64)
    v_of[hh["agent"]] = hh["venue"]
    assert (np.diff(v_of) >= 0).all()
