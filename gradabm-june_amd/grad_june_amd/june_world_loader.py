"""JUNE world (HDF5) -> contact graph in the reference's format - row f4 of SURVEY section 8.

Same public surface as the reference's ``june_world_loader`` package (graph_loader.py:11-39,
network_loader.py:5-44, leisure_loader.py:9-73, agent_data_loader.py:6-33): ``GraphLoader(path,
k_leisure).load_graph(data)``, ``AgentDataLoader(path).load_agent_data(data)`` and one
``<Venue>NetworkLoader`` per venue type.  The reference walks every person in Python loops; this
version is vectorised numpy and emits **bit-identical** node / edge stores (same edge order: grouped by
venue in order of first appearance, members in scan order), so worlds of 10^7-10^8 agents build in
seconds.  Offline, host-side code: nothing here is on the per-timestep path.

Input: a JUNE ``.h5`` file (needs ``h5py``, imported lazily) or the same datasets as a ``.npz`` with
``"<group>/<dataset>"`` keys (tests/golden/make_h5_fixture.py).
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from .graph import HeteroData, ToUndirected

_DATASETS = {
    "population": ("id", "age", "sex", "ethnicity", "area", "super_area", "group_ids", "group_specs"),
    "households": ("id",), "care_homes": ("id",), "companies": ("id",), "schools": ("id",), "universities": ("id",),
    "geography": ("super_area_coordinates", "super_area_id", "area_name", "area_socioeconomic_indices"),
}
_CACHE: Dict[str, Dict[str, np.ndarray]] = {}


def read_world_arrays(path) -> Dict[str, np.ndarray]:
    """The datasets the loaders need, as numpy arrays keyed ``"group/dataset"`` (strings as ``<U``)."""
    key = str(path)
    if key in _CACHE:
        return _CACHE[key]
    if key.endswith(".npz"):
        with np.load(key, allow_pickle=False) as z:
            arrays = {k: z[k] for k in z.files}
    else:
        try:
            import h5py
        except ImportError as e:
            raise ImportError("reading a JUNE .h5 world needs h5py; alternatively convert it to .npz "
                              "(tests/golden/make_h5_fixture.py shows how)") from e
        arrays = {}
        with h5py.File(key, "r") as f:
            for group, names in _DATASETS.items():
                if group not in f:
                    continue
                for name in names:
                    if name in f[group]:
                        a = f[group][name][:]
                        arrays[f"{group}/{name}"] = a.astype("U") if a.dtype.kind in "SO" else a
    _CACHE[key] = arrays
    return arrays


class NetworkLoader:
    spec = None       # value of population/group_specs that marks membership
    plural = None     # HDF5 group holding the venue ids
    columns = ()      # columns of population/group_ids scanned, in order

    def __init__(self, june_world_path):
        self.june_world_path = june_world_path

    def _memberships(self) -> Tuple[np.ndarray, np.ndarray]:
        """(person, venue) pairs in the reference's scan order: column by column, person by person."""
        w = read_world_arrays(self.june_world_path)
        ids, specs = w["population/group_ids"], np.char.strip(w["population/group_specs"])
        people, venues = [], []
        for c in self.columns:
            hit = np.flatnonzero(specs[:, c] == self.spec)
            people.append(hit)
            venues.append(ids[hit, c])
        return np.concatenate(people), np.concatenate(venues)

    def _get_people_per_group(self) -> Dict[int, list]:
        people, venues = self._memberships()
        out: Dict[int, list] = {}
        for p, v in zip(people.tolist(), venues.tolist()):
            out.setdefault(v, []).append(p)
        return out

    def _get_group_ids(self) -> np.ndarray:
        return read_world_arrays(self.june_world_path)[f"{self.plural}/id"]

    def load_network(self, data):
        people, venues = self._memberships()
        # the reference emits edges venue by venue, venues in order of first appearance (dict order)
        _, first, inverse = np.unique(venues, return_index=True, return_inverse=True)
        order = np.argsort(first[inverse], kind="stable")
        group_ids = self._get_group_ids()
        counts = np.zeros(int(group_ids.max()) + 1 if len(group_ids) else 0, dtype=np.int64)
        np.add.at(counts, venues, 1)
        data[self.spec].id = group_ids
        data[self.spec].people = torch.from_numpy(counts[group_ids])
        data["agent", f"attends_{self.spec}", self.spec].edge_index = torch.from_numpy(
            np.vstack((people[order], venues[order])).astype(np.int64))


class HouseholdNetworkLoader(NetworkLoader):
    spec, plural, columns = "household", "households", (0,)


class CareHomeNetworkLoader(NetworkLoader):
    spec, plural, columns = "care_home", "care_homes", (0, 1)


class CompanyNetworkLoader(NetworkLoader):
    spec, plural, columns = "company", "companies", (1,)


class SchoolNetworkLoader(NetworkLoader):
    spec, plural, columns = "school", "schools", (1,)


class UniversityNetworkLoader(NetworkLoader):
    spec, plural, columns = "university", "universities", (1,)


class LeisureNetworkLoader:
    """One "leisure" node per super area, attended by everyone living in its k nearest super areas
    (haversine distance between centroids; reference leisure_loader.py:9-73)."""

    def __init__(self, june_world_path, k=1):
        self.june_world_path = june_world_path
        self.k = k
        w = read_world_arrays(june_world_path)
        self._super_area_coordinates = np.deg2rad(w["geography/super_area_coordinates"])
        self._super_area_ids = w["geography/super_area_id"]

    def _nearest(self, k):
        from sklearn.neighbors import BallTree

        tree = BallTree(self._super_area_coordinates, metric="haversine")
        _, ind = tree.query(self._super_area_coordinates, k=k)
        return ind                                       # [n_super_areas, k] indices, nearest first

    # the reference's helper names (leisure_loader.py:29-56), vectorised
    def _get_people_per_super_area(self):
        lives_in = read_world_arrays(self.june_world_path)["population/super_area"]
        return {int(sa): list(np.flatnonzero(lives_in == sa)) for sa in self._super_area_ids}

    def _get_closest_super_areas(self, super_area, k=3):
        return self._nearest(k)[super_area]

    def _get_close_people_per_super_area(self, k):
        per_sa = self._get_people_per_super_area()
        near = self._nearest(k)
        return {int(sa): [p for j in near[row] for p in per_sa[int(self._super_area_ids[j])]]
                for row, sa in enumerate(self._super_area_ids)}

    def load_network(self, data):
        w = read_world_arrays(self.june_world_path)
        lives_in = w["population/super_area"]
        ids = self._super_area_ids
        residents = {int(sa): np.flatnonzero(lives_in == sa) for sa in ids}
        near = self._nearest(self.k)
        agents, venues, sizes = [], [], []
        for row, sa in enumerate(ids):
            members = np.concatenate([residents[int(ids[j])] for j in near[row]]) if len(ids) else np.zeros(0, np.int64)
            agents.append(members)
            venues.append(np.full(len(members), int(sa), dtype=np.int64))
            sizes.append(len(members))
        data["agent", "attends_leisure", "leisure"].edge_index = torch.from_numpy(
            np.vstack((np.concatenate(agents), np.concatenate(venues))).astype(np.int64))
        data["leisure"].id = torch.tensor([int(sa) for sa in ids])
        data["leisure"].people = torch.tensor(sizes)


class AgentDataLoader:
    def __init__(self, june_world_path):
        self.june_world_path = june_world_path

    def _get_socioeconomic_indices(self):
        w = read_world_arrays(self.june_world_path)
        idx = w["geography/area_socioeconomic_indices"][w["population/area"]]
        return torch.tensor(np.digitize(idx, [0, 0.20, 0.4, 0.6, 0.8, 1.0]), dtype=torch.int8)

    def load_agent_data(self, data):
        w = read_world_arrays(self.june_world_path)
        ag = data["agent"]
        ag.id = torch.from_numpy(w["population/id"])
        ag.age = torch.from_numpy(w["population/age"])
        ag.ethnicity = w["population/ethnicity"].astype("U")
        ag.socioeconomic_index = self._get_socioeconomic_indices()
        ag.area = w["geography/area_name"][w["population/area"]].astype("U")
        ag.sex = torch.from_numpy((np.char.strip(w["population/sex"]) == "f").astype(np.int64))   # m = 0, f = 1
        return data


class GraphLoader:
    def __init__(self, june_world_path, k_leisure=3):
        self.june_world_path = june_world_path
        self.k_leisure = k_leisure

    def load_graph(self, data=None, load_leisure=True,
                   loaders: Sequence[type] = (HouseholdNetworkLoader, CareHomeNetworkLoader, CompanyNetworkLoader,
                                              SchoolNetworkLoader, UniversityNetworkLoader)):
        data = HeteroData() if data is None else data
        for cls in loaders:
            cls(self.june_world_path).load_network(data)
        if load_leisure:
            LeisureNetworkLoader(self.june_world_path, k=self.k_leisure).load_network(data)
        return ToUndirected()(data)
