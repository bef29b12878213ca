"""In-memory stand-in for the two third-party modules the reference imports but this image lacks.

TEST INFRASTRUCTURE, used only by ``make_golden.py`` (and by hand, to run the reference's own
test-suite in the build container).  It is never imported by the product path, by ``bench.py``
or by the ``-m gpu`` tests, and /root/reference does not exist on the GPU box.

* ``torch_geometric`` (pin ``>=2.3``, /root/reference/requirements.txt:5) is not installed and
  cannot be fetched (no network).  The reference uses four things from it
  (/root/reference/grad_june/infection_networks/base.py:5,11-13,79-83; utils.py:10-11;
  june_world_loader/graph_loader.py:1,38):
    - ``MessagePassing(aggr="add", node_dim=-1).propagate(edge_index, x=..., y=...)`` with a
      ``message(self, x_j, y_i)`` hook.  Published semantics for ``flow="source_to_target"``:
      ``x_j = x.index_select(node_dim, edge_index[0])``, ``y_i = y.index_select(node_dim,
      edge_index[1])``, ``out = zeros(dim_size).scatter_add_(node_dim, edge_index[1], message)``
      with ``dim_size = y.size(node_dim)``.  Restated below in ``MessagePassing.propagate``.
    - ``HeteroData`` / ``NodeStorage`` / ``EdgeStorage`` containers and ``T.ToUndirected`` -
      served by this repo's own container (``grad_june_amd.graph``).
* ``h5py`` is only needed by ``june_world_loader`` (offline graph build, out of scope); an empty
  module satisfies the import.

With these registered in ``sys.modules`` the reference's .py files run UNMODIFIED from
/root/reference (never copied).  What the resulting golden vectors therefore pin: every line of
the reference's own code on the path (masks, beta, p_contact, clamp/exp, Gumbel sampler,
transmission profile, infect_people, timer, policies); the sparse gather/scatter arithmetic in
them is PyG's published algorithm as restated here, additionally pinned by the reference's exact
known-answer test (/root/reference/test/unit/infection_networks/test_base.py:39-44).
"""
from __future__ import annotations

import inspect
import os
import sys
import types

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "gradabm-june_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from grad_june_amd import graph as _graph  # noqa: E402


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", node_dim=-2, flow="source_to_target", **_):
        super().__init__()
        assert aggr == "add" and flow == "source_to_target"
        self.aggr = aggr
        self.node_dim = node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        params = [p for p in inspect.signature(self.message).parameters]
        dim = self.node_dim
        msg_kwargs = {}
        dim_size = None
        for p in params:
            if p.endswith("_j"):
                msg_kwargs[p] = kwargs[p[:-2]].index_select(dim, edge_index[0])
            elif p.endswith("_i"):
                src = kwargs[p[:-2]]
                dim_size = src.size(dim)
                msg_kwargs[p] = src.index_select(dim, edge_index[1])
            else:
                msg_kwargs[p] = kwargs[p]
        msg = self.message(**msg_kwargs)
        if size is not None and size[1] is not None:
            dim_size = size[1]
        shape = list(msg.shape)
        shape[dim] = dim_size
        out = msg.new_zeros(shape)
        return out.scatter_add_(dim, edge_index[1].expand_as(msg), msg)

    def message(self, x_j):  # pragma: no cover - overridden by the reference
        return x_j


def install():
    """Register the stand-in modules.  Idempotent."""
    if "torch_geometric" in sys.modules and getattr(
        sys.modules["torch_geometric"], "_gj_standin", False
    ):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tg = mod("torch_geometric", _gj_standin=True, __version__="0.0-standin")
    data = mod("torch_geometric.data", HeteroData=_graph.HeteroData)
    hd = mod("torch_geometric.data.hetero_data", HeteroData=_graph.HeteroData)
    st = mod(
        "torch_geometric.data.storage",
        BaseStorage=_graph.BaseStorage,
        NodeStorage=_graph.NodeStorage,
        EdgeStorage=_graph.EdgeStorage,
    )
    tr = mod("torch_geometric.transforms", ToUndirected=_graph.ToUndirected)
    nn = mod("torch_geometric.nn")
    conv = mod("torch_geometric.nn.conv", MessagePassing=MessagePassing)
    nn.conv = conv
    nn.MessagePassing = MessagePassing
    data.hetero_data = hd
    data.storage = st
    tg.data, tg.transforms, tg.nn = data, tr, nn
    mod("h5py")


def import_reference(path="/root/reference"):
    """Import the reference package (read-only, no bytecode written)."""
    sys.dont_write_bytecode = True
    install()
    if path not in sys.path:
        sys.path.insert(0, path)
    import grad_june  # noqa: F401

    return grad_june
