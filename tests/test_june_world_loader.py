"""CPU: row f4 - the vectorised JUNE world loader.  Restates the reference's
test/unit/test_june_world_loader.py:16-161 on the same HDF5 world (converted to .npz by
tests/golden/make_h5_fixture.py) and additionally requires the built graph to be IDENTICAL to the one
the reference pickled from that file (test/data/data.pkl, here as world769.npz)."""
import os

import numpy as np
import pytest
import torch

import gj_testlib as L
from grad_june_amd import june_world_loader as W
from grad_june_amd.graph import HeteroData

PATH = os.path.join(L.GOLDEN, "june_world_h5.npz")


def test_agent_properties():
    d = W.AgentDataLoader(PATH).load_agent_data(HeteroData())
    ag = d["agent"]
    assert len(ag["id"]) == len(ag["area"]) == len(ag["age"]) == len(ag["sex"]) == 769
    assert ag["age"][14] == 6 and ag["sex"][14] == 1 and ag["age"][22] == 8 and ag["sex"][22] == 0
    assert ag["area"][14] == "E00023664" and ag["area"][300] == "E00079478"
    assert ag["socioeconomic_index"].dtype == torch.int8


@pytest.mark.parametrize("cls,ids,expected", [
    (W.HouseholdNetworkLoader, (0, 50), ([272], [220, 248])),
    (W.CompanyNetworkLoader, (0, 11), ([177, 551], [69, 75, 136, 570, 695])),
    (W.SchoolNetworkLoader, (0,), ([4, 5],)),
    (W.UniversityNetworkLoader, (38,), ([57, 58, 65, 59, 60, 61, 64, 62, 63],)),
])
def test_people_per_group(cls, ids, expected):
    ret = cls(PATH)._get_people_per_group()
    for i, exp in zip(ids, expected):
        assert set(exp).issubset(set(ret[i]))


@pytest.mark.parametrize("spec,cls,n_groups,total,group_ids,n_people", [
    ("household", W.HouseholdNetworkLoader, 355, 745, (2, 20, 209), (1, 6, 1)),
    ("care_home", W.CareHomeNetworkLoader, 1, 27, (0,), (27,)),
    ("company", W.CompanyNetworkLoader, 1980, 333, (0, 10, 1455), (2, 0, 7)),
    ("school", W.SchoolNetworkLoader, 1, 78, (0,), (78,)),
    ("university", W.UniversityNetworkLoader, 39, 43, (38, 23), (9, 17)),
])
def test_load_network(spec, cls, n_groups, total, group_ids, n_people):
    d = HeteroData()
    cls(PATH).load_network(d)
    assert len(d[spec].id) == n_groups and len(d[f"attends_{spec}"].edge_index[0]) == total
    for g, n in zip(group_ids, n_people):
        assert d[spec].people[g] == n


def test_leisure_network():
    ll = W.LeisureNetworkLoader(PATH, k=3)
    per = ll._get_people_per_super_area()
    assert len(per[0]) == 294 and 50 in per[0] and len(per[2]) == 325 and 464 in per[2]
    assert (ll._get_closest_super_areas(0, k=3) == [0, 2, 1]).all()
    assert (ll._get_closest_super_areas(1, k=3) == [1, 0, 2]).all()
    assert (ll._get_closest_super_areas(2, k=3) == [2, 0, 1]).all()
    close = ll._get_close_people_per_super_area(k=3)
    assert [len(close[i]) for i in range(3)] == [769, 769, 769]
    assert len(ll._get_close_people_per_super_area(k=2)[1]) == 444
    d = HeteroData()
    ll.load_network(d)
    assert len(d["attends_leisure"]["edge_index"][0]) > 1500 and len(d["leisure"]["id"]) == 3
    assert d["leisure"]["people"][0] == 769 and d["leisure"]["people"][2] == 769


def test_graph_is_identical_to_the_references_pickle():
    """GraphLoader(k_leisure=1) + AgentDataLoader == test/data/data.pkl (example_scripts/make_data.py)."""
    d = W.GraphLoader(PATH, k_leisure=1).load_graph(HeteroData())
    W.AgentDataLoader(PATH).load_agent_data(d)
    ref = L.load_npz("world769.npz")
    expected = {"household": (355, 745), "company": (1980, 333), "school": (1, 78), "university": (39, 43),
                "care_home": (1, 27), "leisure": (3, 769)}
    for s, (nv, ne) in expected.items():
        ei = d["attends_" + s].edge_index.numpy()
        assert len(d[s]["id"]) == nv and ei.shape == (2, ne)
        assert np.array_equal(ei[0], ref[f"es/{s}/agent"]) and np.array_equal(ei[1], ref[f"es/{s}/venue"]), s
        assert np.array_equal(np.asarray(d[s]["people"]), ref[f"es/{s}/people"]), s
        assert np.array_equal(np.asarray(d[s]["id"]), ref[f"venue_id/{s}"]), s
        assert torch.equal(d["rev_attends_" + s].edge_index, d["attends_" + s].edge_index.flip(0))
    ag = d["agent"]
    assert np.array_equal(ag.age.numpy(), ref["age"]) and np.array_equal(ag.sex.numpy(), ref["sex"])
    assert np.array_equal(ag.ethnicity, ref["agent/ethnicity"]) and np.array_equal(ag.area, ref["agent/area"])
    school = set(d["attends_school"].edge_index[0].tolist())
    assert not school & set(d["attends_company"].edge_index[0].tolist())
    assert len(set(d["attends_care_home"].edge_index[0].tolist()) & set(d["attends_household"].edge_index[0].tolist())) == 3
