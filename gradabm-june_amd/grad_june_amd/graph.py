"""Heterogeneous contact-graph container: the data surface of the drop-in boundary.

The reference stores a world as a pickled PyG ``HeteroData`` (emitted by
``june_world_loader``: /root/reference/grad_june/june_world_loader/graph_loader.py:16-39,
consumed by ``Runner.get_data``: /root/reference/grad_june/runner.py:65-91).  PyG is not a
dependency of this package, so this module provides a small dict-backed container that

* accepts every access idiom the reference uses on the object:
  ``data["agent"].x``, ``data["agent"]["x"]``, ``data["attends_school"].edge_index`` (lookup
  by relation name), ``data["agent", "attends_school", "school"]``, ``len(data[name]["id"])``,
  ``del data["rev_attends_school"]``, ``data["results"] = {}``, ``data.results``,
  ``data.to(device)`` (recursing into dict-valued attributes);
* unpickles worlds written by the reference (class paths
  ``torch_geometric.data.hetero_data.HeteroData`` / ``torch_geometric.data.storage.*``)
  without PyG installed (:func:`load_world`);
* offers the ``ToUndirected`` transform the loader applies to a bipartite world
  (adds ``(venue, "rev_" + rel, "agent")`` with the two index rows swapped).

Nothing in here is on the per-timestep path; the per-step kernels read the CSR plan built
once by :mod:`grad_june_amd.plan`.
"""
from __future__ import annotations

import io
import pickle
from typing import Any, Dict, Iterator, Tuple, Union

import torch

EdgeKey = Tuple[str, str, str]


def _map_tensors(value: Any, fn):
    """Apply ``fn`` to every tensor inside ``value`` (tensors, dicts, lists, tuples)."""
    if isinstance(value, torch.Tensor):
        return fn(value)
    if isinstance(value, dict):
        return {k: _map_tensors(v, fn) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return type(value)(_map_tensors(v, fn) for v in value)
    return value


class BaseStorage:
    """Attribute bag with both ``store.x`` and ``store["x"]`` access."""

    def __init__(self, _key=None, **attrs):
        object.__setattr__(self, "_mapping", dict(attrs))
        object.__setattr__(self, "_key", _key)

    # attribute style -------------------------------------------------------------
    def __getattr__(self, name: str):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        mapping = self.__dict__.get("_mapping")
        if mapping is None:  # half-constructed object during unpickling
            raise AttributeError(name)
        try:
            return mapping[name]
        except KeyError:
            raise AttributeError(
                f"'{type(self).__name__}' object has no attribute '{name}'"
            ) from None

    def __setattr__(self, name: str, value):
        if name in ("_mapping", "_key", "_parent"):
            object.__setattr__(self, name, value)
        else:
            self._mapping[name] = value

    def __delattr__(self, name: str):
        try:
            del self._mapping[name]
        except KeyError:
            raise AttributeError(name) from None

    # item style ------------------------------------------------------------------
    def __getitem__(self, name: str):
        return self._mapping[name]

    def __setitem__(self, name: str, value):
        self._mapping[name] = value

    def __delitem__(self, name: str):
        del self._mapping[name]

    def __contains__(self, name: str) -> bool:
        return name in self._mapping

    def __iter__(self) -> Iterator[str]:
        return iter(self._mapping)

    def __len__(self) -> int:
        return len(self._mapping)

    def get(self, name: str, default=None):
        return self._mapping.get(name, default)

    def keys(self):
        return self._mapping.keys()

    def values(self):
        return self._mapping.values()

    def items(self):
        return self._mapping.items()

    def to_dict(self) -> Dict[str, Any]:
        return dict(self._mapping)

    def apply(self, fn) -> "BaseStorage":
        for k, v in list(self._mapping.items()):
            self._mapping[k] = _map_tensors(v, fn)
        return self

    # pickling: tolerate the state layout PyG writes ({_mapping, _parent, _key}) -----
    def __getstate__(self):
        return {"_mapping": self._mapping, "_key": self._key}

    def __setstate__(self, state):
        object.__setattr__(self, "_mapping", dict(state.get("_mapping", {})))
        object.__setattr__(self, "_key", state.get("_key"))

    def __repr__(self) -> str:
        def short(v):
            if isinstance(v, torch.Tensor):
                return f"[{', '.join(str(s) for s in v.shape)}]"
            if hasattr(v, "shape"):
                return f"[{', '.join(str(s) for s in v.shape)}]"
            if isinstance(v, dict):
                return "{" + ", ".join(f"{k}={short(x)}" for k, x in v.items()) + "}"
            return repr(v)

        body = ", ".join(f"{k}={short(v)}" for k, v in self._mapping.items())
        return f"{type(self).__name__}({body})"


class NodeStorage(BaseStorage):
    pass


class EdgeStorage(BaseStorage):
    pass


class HeteroData:
    """Typed node stores + typed (src, relation, dst) edge stores + global attributes."""

    def __init__(self):
        self.__dict__["_global_store"] = BaseStorage()
        self.__dict__["_node_store_dict"] = {}
        self.__dict__["_edge_store_dict"] = {}

    # key handling ------------------------------------------------------------------
    def _canonical(self, key) -> Union[str, EdgeKey]:
        """Resolve a user key to a node type (str) or a full edge triple."""
        if isinstance(key, tuple):
            if len(key) == 1:
                key = key[0]
            elif len(key) == 3:
                return tuple(key)
            elif len(key) == 2:
                hits = [
                    k for k in self._edge_store_dict if k[0] == key[0] and k[2] == key[1]
                ]
                if len(hits) == 1:
                    return hits[0]
                return (key[0], "to", key[1])
            else:
                raise KeyError(key)
        if isinstance(key, str) and key not in self._node_store_dict:
            hits = [k for k in self._edge_store_dict if k[1] == key]
            if len(hits) == 1:
                return hits[0]
        return key

    def __getitem__(self, key):
        key = self._canonical(key)
        if isinstance(key, str):
            hit = self._global_store.get(key, None)
            if hit is not None:
                return hit
            store = self._node_store_dict.get(key)
            if store is None:
                store = self._node_store_dict[key] = NodeStorage(_key=key)
            return store
        store = self._edge_store_dict.get(key)
        if store is None:
            store = self._edge_store_dict[key] = EdgeStorage(_key=key)
        return store

    def __setitem__(self, key: str, value):
        if key in self._node_store_dict:
            raise AttributeError(f"'{key}' is already present as a node type")
        self._global_store[key] = value

    def __delitem__(self, key):
        key = self._canonical(key)
        if isinstance(key, str):
            if key in self._node_store_dict:
                del self._node_store_dict[key]
            elif key in self._global_store:
                del self._global_store[key]
            else:
                raise KeyError(key)
        else:
            del self._edge_store_dict[key]

    def __contains__(self, key) -> bool:
        key = self._canonical(key)
        if isinstance(key, str):
            return key in self._node_store_dict or key in self._global_store
        return key in self._edge_store_dict

    def __getattr__(self, name: str):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        gs = self.__dict__.get("_global_store")
        if gs is not None and name in gs:
            return gs[name]
        raise AttributeError(f"'HeteroData' object has no attribute '{name}'")

    def __setattr__(self, name: str, value):
        if name in ("_global_store", "_node_store_dict", "_edge_store_dict"):
            self.__dict__[name] = value
        else:
            self._global_store[name] = value

    # views -----------------------------------------------------------------------
    @property
    def node_types(self):
        return list(self._node_store_dict)

    @property
    def edge_types(self):
        return list(self._edge_store_dict)

    @property
    def node_stores(self):
        return list(self._node_store_dict.values())

    @property
    def edge_stores(self):
        return list(self._edge_store_dict.values())

    def node_items(self):
        return list(self._node_store_dict.items())

    def edge_items(self):
        return list(self._edge_store_dict.items())

    # device movement -----------------------------------------------------------------
    def apply(self, fn) -> "HeteroData":
        self._global_store.apply(fn)
        for s in self._node_store_dict.values():
            s.apply(fn)
        for s in self._edge_store_dict.values():
            s.apply(fn)
        return self

    def to(self, device, non_blocking: bool = False) -> "HeteroData":
        return self.apply(lambda t: t.to(device, non_blocking=non_blocking))

    def cpu(self) -> "HeteroData":
        return self.to("cpu")

    def cuda(self, device=None) -> "HeteroData":
        return self.to("cuda" if device is None else device)

    # pickling ------------------------------------------------------------------------
    def __getstate__(self):
        return {
            "_global_store": self._global_store,
            "_node_store_dict": self._node_store_dict,
            "_edge_store_dict": self._edge_store_dict,
        }

    def __setstate__(self, state):
        self.__dict__["_global_store"] = state.get("_global_store") or BaseStorage()
        self.__dict__["_node_store_dict"] = dict(state.get("_node_store_dict", {}))
        self.__dict__["_edge_store_dict"] = dict(state.get("_edge_store_dict", {}))

    def __repr__(self) -> str:
        rows = [f"  {k}={v!r}" for k, v in self._node_store_dict.items()]
        rows += [f"  {k}={v!r}" for k, v in self._edge_store_dict.items()]
        return "HeteroData(\n" + ",\n".join(rows) + "\n)"


class ToUndirected:
    """Add the reverse relation of every bipartite edge type.

    For ``(src, rel, dst)`` with ``src != dst`` a store ``(dst, "rev_" + rel, src)`` is added
    whose ``edge_index`` is the row-flip of the forward one, in the same edge order - this
    is what the reference's ``T.ToUndirected()`` produces on its worlds
    (/root/reference/grad_june/june_world_loader/graph_loader.py:38).
    """

    def __call__(self, data: HeteroData) -> HeteroData:
        for key, store in data.edge_items():
            src, rel, dst = key
            if "edge_index" not in store or rel.startswith("rev_"):
                continue
            if src == dst:
                ei = store.edge_index
                store.edge_index = torch.cat([ei, ei.flip(0)], dim=1)
                continue
            rev = data[dst, "rev_" + rel, src]
            rev.edge_index = store.edge_index.flip(0)
        return data


class _WorldUnpickler(pickle.Unpickler):
    """Maps PyG's class paths onto this module's classes; everything else is stock."""

    _MAP = {
        ("torch_geometric.data.hetero_data", "HeteroData"): HeteroData,
        ("torch_geometric.data.storage", "BaseStorage"): BaseStorage,
        ("torch_geometric.data.storage", "NodeStorage"): NodeStorage,
        ("torch_geometric.data.storage", "EdgeStorage"): EdgeStorage,
        ("torch_geometric.data.storage", "GlobalStorage"): BaseStorage,
    }

    def find_class(self, module, name):
        hit = self._MAP.get((module, name))
        if hit is not None:
            return hit
        return super().find_class(module, name)


def load_world(path_or_file) -> HeteroData:
    """Read a pickled world: one written by the reference (PyG classes) or by this package."""
    if hasattr(path_or_file, "read"):
        return _WorldUnpickler(path_or_file).load()
    with open(path_or_file, "rb") as f:
        return _WorldUnpickler(io.BufferedReader(f)).load()


def save_world(data: HeteroData, path) -> None:
    with open(path, "wb") as f:
        pickle.dump(data, f, protocol=4)


def locality_order(data: HeteroData, by: str = "household"):
    """Graph-compile-time locality permutation: renumber the agents so that the members
    of one venue of type ``by`` are consecutive (ordered by their first such venue; agents without one keep
    their relative order at the end).  Tiles of that edge set then sit on the slice/block diagonal, and a
    contiguous multi-GPU partition keeps most such venues rank-local.

    Returns ``(data, original)``: the same container with every per-agent attribute (tensors, numpy arrays,
    and the tensors inside dict attributes such as ``symptoms`` / ``infection_parameters``) and every agent
    row of an ``edge_index`` renumbered; ``original[new] = old`` index, so a per-agent result ``r`` of a run
    on the reordered world is reported in the original order as ``out[original] = r``.  ``agent.id`` keeps
    the original identifiers (it is permuted like any other attribute)."""
    import numpy as np

    agent = data["agent"]
    A = len(agent["id"])
    key = None
    for k, _ in data.edge_items():
        if k[0] == "agent" and k[1] == "attends_" + by:
            key = k
    if key is None:
        raise KeyError(f"no ('agent', 'attends_{by}', ...) edge type in the world")
    ei = data[key].edge_index
    dev = ei.device
    first = torch.full((A,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    first.scatter_reduce_(0, ei[0].long(), ei[1].long(), reduce="amin")
    original = torch.argsort(first, stable=True)                 # new position -> old index
    new_of = torch.empty(A, dtype=torch.int64, device=dev)
    new_of[original] = torch.arange(A, device=dev)
    orig_np = original.cpu().numpy()

    def permute(v):
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == A:
            return v[original.to(v.device)]
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == A:
            return v[orig_np]
        if isinstance(v, dict):
            return {k: permute(x) for k, x in v.items()}
        return v

    for name in list(agent.keys()):
        agent[name] = permute(agent[name])
    for k, store in data.edge_items():
        if "edge_index" not in store:
            continue
        e = store.edge_index
        rows = [new_of.to(e.device)[e[i].long()].to(e.dtype) if k[2 * i] == "agent" else e[i] for i in (0, 1)]
        store.edge_index = torch.stack(rows)
    return data, original
