"""CPU: the reference's own known-answer tests for the path, restated against the oracle.

Each test names the reference test it restates (paths relative to /root/reference/test/unit/).
The expected numbers are the reference tests' own constants."""
import numpy as np
import torch

import gj_oracle as O


def _six_agent_world():
    return {"agent": torch.arange(6), "venue": torch.tensor([0, 0, 0, 1, 1, 1]), "people": torch.tensor([2, 2])}


def test_infection_passing_exact():
    """infection_networks/test_base.py:16-44: 6 agents, 2 schools, beta=2, people=[2,2], dt=1."""
    es = _six_agent_world()
    beta = 10.0 ** torch.tensor(float(np.log10(2.0)))
    ts = O.infection_network(kind="plain", beta=beta, people=es["people"], agent_index=es["agent"],
                             venue_index=es["venue"], transmission=torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5, 0.6]),
                             susceptibility=torch.tensor([1, 2, 3, 0.5, 0.7, 1.0]))
    p = O.not_infected_probabilities([ts], 6, 1.0)
    assert np.allclose(p.numpy(), np.exp(-np.array([1.2, 2.4, 3.6, 1.5, 2.1, 3])))


def test_p_contact_edge_cases():
    """base.py:63-69 on people = 0,1,2,3,25."""
    pc = O.p_contact(torch.tensor([0, 1, 2, 3, 26]))
    assert pc.tolist() == [0.0, 1.0, 1.0, 0.5, float(np.float32(1.0) / np.float32(25))]


def test_leisure_masks_exact():
    """infection_networks/test_leisure_network.py:61-77: weekday/weekend lookup [day, sex, age]."""
    table = torch.zeros(2, 2, 100)
    table[0, 0, :50], table[0, 0, 50:] = 0.5, 0.2     # weekday male 0-50: .5, 50-100: .2
    table[0, 1, :] = 0.5
    table[1] = 1.0
    age, sex = torch.tensor([1, 60, 20, 30, 50]), torch.tensor([0, 1, 1, 0, 0])
    wd = O.leisure_agent_probabilities(table, sex, age, 0)
    we = O.leisure_agent_probabilities(table, sex, age, 1)
    assert torch.equal(wd, torch.tensor([0.5, 0.5, 0.5, 0.5, 0.2]))
    assert torch.equal(we, torch.ones(5))
    assert torch.equal(1.0 * wd * (0.5 * torch.ones(5)), 0.5 * torch.tensor([0.5, 0.5, 0.5, 0.5, 0.2]))


def test_quarantine_mask():
    """policies/test_quarantine_policies.py:16-38: stages [0..4], threshold 3 -> [1,1,1,0,0]."""
    m = O.quarantine_mask(torch.tensor([0, 1, 2, 3, 4]), [3])
    assert m.tolist() == [1, 1, 1, 0, 0]
    assert O.quarantine_mask(torch.tensor([0, 5]), [None]).tolist() == [1, 1]


def test_social_distancing_ratio():
    """policies/test_interaction_policies.py:92-123: -log(p) scales with the beta factor exactly."""
    torch.manual_seed(0)
    n = 100
    es = {"agent": torch.arange(n), "venue": torch.arange(n) // 25, "people": 25 * torch.ones(4)}
    trans = torch.rand(n) + 1.0
    susc = torch.ones(n)
    susc[::10] = 0.0
    out = []
    for f in (0.5, 0.2):
        beta = 10.0 ** torch.tensor(0.0) * torch.tensor(f)
        ts = O.infection_network(kind="plain", beta=beta, people=es["people"], agent_index=es["agent"],
                                 venue_index=es["venue"], transmission=trans, susceptibility=susc)
        p = O.not_infected_probabilities([ts], n, 1.0)
        t = -torch.log(p)
        out.append(t)
    keep = out[0] > 5e-6
    assert np.allclose((out[1][keep] / out[0][keep]).numpy(), 0.2 / 0.5)


def test_close_venue_and_seed_survival():
    """policies/test_close_venue_policies.py:46-69: beta=1e3, transmission+1: only the 10 seeds
    (susceptibility 0) keep p=1; with the network closed nobody is exposed."""
    n = 100
    es = {"agent": torch.arange(n), "venue": torch.arange(n) // 25, "people": 25 * torch.ones(4)}
    susc = torch.ones(n)
    susc[::10] = 0.0
    ts = O.infection_network(kind="plain", beta=10.0 ** torch.tensor(3.0), people=es["people"],
                             agent_index=es["agent"], venue_index=es["venue"], transmission=torch.ones(n),
                             susceptibility=susc)
    assert np.isclose(O.not_infected_probabilities([ts], n, 1.0).sum().item(), 10.0)
    assert np.isclose(O.not_infected_probabilities([], n, 1.0).sum().item(), n)


def test_household_ignores_quarantine():
    """policies/test_quarantine_policies.py:40-72: everyone quarantined -> company passes nothing,
    household (raw values, base.py:144-149) still does."""
    n = 100
    comp = {"agent": torch.arange(n), "venue": torch.arange(n) // 25, "people": 25 * torch.ones(4)}
    house = {"agent": torch.arange(n), "venue": torch.arange(n) // 4, "people": 4 * torch.ones(25)}
    susc = torch.ones(n)
    susc[::10] = 0.0
    q = O.quarantine_mask(5 * torch.ones(n), [3])
    kw = dict(beta=10.0 ** torch.tensor(3.0), transmission=torch.ones(n), susceptibility=susc, qmask=q)
    ts_c = O.infection_network(kind="plain", people=comp["people"], agent_index=comp["agent"], venue_index=comp["venue"], **kw)
    assert np.isclose(O.not_infected_probabilities([ts_c], n, 1.0).sum().item(), n)
    ts_h = O.infection_network(kind="household", people=house["people"], agent_index=house["agent"], venue_index=house["venue"], **kw)
    assert np.isclose(O.not_infected_probabilities([ts_c, ts_h], n, 1.0).sum().item(), 10.0)


def test_transmission_profile():
    """test_transmission.py:22-33: zero when nobody is infected, > 0 for infected agents at t=5."""
    torch.manual_seed(1)
    n = 1000
    mx = torch.distributions.LogNormal(0.0, 0.5).sample((n,))
    shp = torch.distributions.Normal(1.56, 0.08).sample((n,))
    rt = torch.distributions.Normal(0.53, 0.03).sample((n,))
    sh = torch.distributions.Normal(-2.12, 0.1).sample((n,))
    z = torch.zeros(n)
    assert O.transmission_update(mx, shp, rt, sh, z, z, 5.0).sum() == 0
    inf = torch.zeros(n)
    inf[::10] = 1.0
    tr = O.transmission_update(mx, shp, rt, sh, z, inf, 5.0)
    assert (tr[::10] > 0).all() and tr.sum() == tr[::10].sum()


def test_sampler_statistics():
    """infection_networks/test_is_infected_sampler.py:7-24: mean of draws ~ 1-p (rtol 0.1)."""
    torch.manual_seed(999)
    p = torch.tensor([0.2, 0.5, 0.7, 0.3])
    acc = torch.zeros(4)
    n = 2000
    for _ in range(n):
        acc += O.sample_infected(p, O.draw_exp_noise(4))
    assert np.allclose((acc / n).numpy(), (1 - p).numpy(), rtol=0.1)


def test_infect_people_floor_reinfection():
    """model.py:103-110: is_infected is additive (can reach 2.0)."""
    s, i, t = O.infect_people(torch.tensor([0.0, 1.0]), torch.tensor([1.0, 0.0]), torch.tensor([2.0, 0.0]),
                              torch.tensor([1.0, 1.0]), 7.0)
    assert s.tolist() == [0.0, 0.0] and i.tolist() == [2.0, 1.0] and t.tolist() == [7.0, 7.0]
