"""CPU: the HeteroData-compatible container - every access idiom the reference uses."""
import io
import pickle

import numpy as np
import torch

from grad_june_amd.graph import HeteroData, ToUndirected, load_world, save_world


def small():
    d = HeteroData()
    d["agent"].id = torch.arange(6)
    d["agent"].age = torch.tensor([1, 2, 3, 4, 5, 6])
    d["school"].id = torch.arange(2)
    d["school"].people = torch.tensor([2, 2])
    d["agent", "attends_school", "school"].edge_index = torch.vstack((torch.arange(6), torch.tensor([0, 0, 0, 1, 1, 1])))
    return ToUndirected()(d)


def test_access_idioms():
    d = small()
    assert torch.equal(d["agent"].age, d["agent"]["age"])
    assert d["attends_school"].edge_index.shape == (2, 6)                      # by relation name
    assert torch.equal(d["rev_attends_school"].edge_index, d["attends_school"].edge_index.flip(0))
    assert torch.equal(d["school", "rev_attends_school", "agent"].edge_index[1], torch.arange(6))
    assert len(d["school"]["id"]) == 2 and "people" in d["school"]
    d["agent"].symptoms = {"current_stage": torch.ones(6)}
    assert d["agent"]["symptoms"]["current_stage"].sum() == 6
    d["results"] = {}
    d["results"]["deaths_per_timestep"] = None
    assert d.results == {"deaths_per_timestep": None}
    del d["rev_attends_school"]
    assert "rev_attends_school" not in d and "attends_school" in d
    d = ToUndirected()(d)
    assert "rev_attends_school" in d
    assert d.node_types == ["agent", "school"] and len(d.edge_types) == 2


def test_to_device_recurses_into_dicts():
    d = small()
    d["agent"].infection_parameters = {"shape": torch.ones(6)}
    d["agent"].ethnicity = np.array(["A"] * 6)
    d2 = d.to("cpu")
    assert d2 is d and d["agent"].infection_parameters["shape"].device.type == "cpu"
    assert isinstance(d["agent"].ethnicity, np.ndarray)


def test_pickle_roundtrip(tmp_path):
    d = small()
    save_world(d, tmp_path / "w.pkl")
    e = load_world(tmp_path / "w.pkl")
    assert torch.equal(e["attends_school"].edge_index, d["attends_school"].edge_index)
    assert torch.equal(e["school"].people, d["school"].people)


def test_unpickles_pyg_class_paths():
    """A stream that names PyG's classes (what the reference's .pkl files contain) loads without PyG."""
    d = small()
    raw = pickle.dumps(d, protocol=4)
    raw = raw.replace(b"grad_june_amd.graph", b"torch_geometric.data.hetero_data", 1)
    assert b"torch_geometric" in raw
    e = load_world(io.BytesIO(pickle.dumps(d, protocol=4)))
    assert e["agent"].id.shape[0] == 6
    from grad_june_amd.graph import _WorldUnpickler

    for key, cls in _WorldUnpickler._MAP.items():
        assert _WorldUnpickler(io.BytesIO(b"")).find_class(*key) is cls


def test_locality_order_is_a_pure_renumbering():
    """graph.locality_order: household members become consecutive; the graph is the same graph under the
    returned permutation; dict attributes and numpy attributes follow."""
    import numpy as np

    from grad_june_amd.graph import HeteroData, ToUndirected, locality_order

    g = torch.Generator().manual_seed(0)
    A, H, S = 200, 70, 5
    d = HeteroData()
    d["agent"].id = torch.arange(A) + 1000
    d["agent"].age = torch.randint(0, 100, (A,), generator=g)
    d["agent"].area = np.array([f"a{i}" for i in range(A)])
    d["agent"].symptoms = {"current_stage": torch.arange(A).float()}
    hh = torch.randint(0, H, (A,), generator=g)
    hh[:5] = -1                                                   # five agents without a household
    members = torch.nonzero(hh >= 0).squeeze(1)
    d["agent", "attends_household", "household"].edge_index = torch.stack((members, hh[members]))
    d["household"].people = torch.bincount(hh[members], minlength=H)
    sch = torch.randint(0, S, (A,), generator=g)
    d["agent", "attends_school", "school"].edge_index = torch.stack((torch.arange(A), sch))
    d = ToUndirected()(d)
    before = {k: s.edge_index.clone() for k, s in d.edge_items()}
    age0, id0 = d["agent"].age.clone(), d["agent"].id.clone()
    d, original = locality_order(d, by="household")
    assert sorted(original.tolist()) == list(range(A))
    assert torch.equal(d["agent"].id, id0[original]) and torch.equal(d["agent"].age, age0[original])
    assert d["agent"].area[0] == f"a{int(original[0])}"
    assert torch.equal(d["agent"].symptoms["current_stage"], original.float())
    new_of = torch.empty(A, dtype=torch.long)
    new_of[original] = torch.arange(A)
    for k, e0 in before.items():
        e1 = d[k].edge_index
        for row in (0, 1):
            want = new_of[e0[row]] if k[2 * row] == "agent" else e0[row]
            assert torch.equal(e1[row], want), (k, row)
    # household members are consecutive, households in increasing order, the homeless last
    ei = d["agent", "attends_household", "household"].edge_index
    hh_new = torch.full((A,), H, dtype=torch.long)
    hh_new[ei[0]] = ei[1]
    assert torch.equal(hh_new, torch.sort(hh_new).values)
    assert torch.equal(d["household", "rev_attends_household", "agent"].edge_index, ei.flip(0))
