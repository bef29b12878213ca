"""Bridge between the reference-shaped ``HeteroData`` world and the compiled device plan.

``engine_for(data, networks)`` compiles ``data`` once (plan.py) and caches the plan + engine
against the identity of the world's edge tensors, so the Python API mirrors
(``InfectionNetworks.forward``, ``GradJune.forward``) can be called with the same ``data``
object every timestep - exactly how the reference's Runner drives them (runner.py:163-166).
"""
from __future__ import annotations

import weakref
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

import os

from .engine import AgentBuffers, InfectionEngine
from .plan import DevicePlan, NetworkSpec, compile_plan

_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()

#: device layout the API mirrors compile worlds into: "tiled" (LDS-tiled fast path, default) or "csr"
DEFAULT_LAYOUT = os.environ.get("GRAD_JUNE_AMD_LAYOUT", "tiled")
#: where the contact graph is compiled: "auto" (default) = by the library's compile kernels (gj_compile_*, tiling_native)
#: when the world's edge lists already live on the HIP device - what Runner.get_data leaves behind - else with numpy on
#: the host; "1" / "0" force one or the other.  The plans are identical array for array.
DEVICE_COMPILE = os.environ.get("GRAD_JUNE_AMD_DEVICE_COMPILE", "auto")


#: tile geometry of mid-size worlds: "auto" (default) = when the plan is compiled on the device (milliseconds per
#: candidate) and the world has 2e5 .. 4e7 set-edges, compile it under a few candidate geometries, time the two
#: sparse passes on dummy data and keep the fastest - the size-based defaults of tiling.py are right for ~1.5
#: memberships per agent and set, denser worlds (BASELINE's C2: 5) want larger tiles (0.156 -> 0.110 ms per step);
#: "0" = always the defaults.  The geometry cannot change a bit of a result: every edge's term goes to fixed point
#: BEFORE anything is added to it (phase B merges a venue's runs as integers - round 2 merged them in fp32, where the
#: geometry reached the last bits - phase D adds fixed-point terms, the direct form sums in COO order), so the race
#: between candidates decides the speed only (tests/test_gpu_api.py::test_tile_geometry_cannot_change_a_bit).
TUNE = os.environ.get("GRAD_JUNE_AMD_TUNE", "auto")
#: run form of the edge set the agents are ordered by (plan.compile_plan(runs=...)): "auto" (default: a household-major
#: world - system.locality_order - whose household set is too large for the direct form keeps one edge per agent out
#: of the tiled arrays), "off", or a comma-separated list of set names that must take it (tests: small worlds)
RUNS = os.environ.get("GRAD_JUNE_AMD_RUNS", "auto")


def _runs_option():
    if RUNS == "auto":
        return None
    if RUNS in ("off", "0", ""):
        return False
    return tuple(x for x in (RUNS if isinstance(RUNS, str) else ",".join(RUNS)).split(",") if x)
TUNE_CANDIDATES = ({}, {"eb_target": 131072, "sv_max": 16384}, {"eb_target": 65536, "sv_max": 16384},
                   {"eb_target": 32768, "sv_max": 16384})      # (round 4: a 1 M-agent world has ~1 venue block per CU otherwise)
TUNE_MIN_EDGES, TUNE_MAX_EDGES = 200_000, 40_000_000
N_MAX_TUNE_NETS = 16


def _time_passes(engine: InfectionEngine, specs, steps: int = 8) -> float:
    """ms per (scatter, venues, agents without decision) on dummy data - what the tuner compares."""
    import time

    plan = engine.plan
    dev = plan.device
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    bufs = AgentBuffers(plan, susceptibility=torch.ones(plan.host.n_agents, dtype=torch.float32, device=dev),
                        transmission=torch.rand(plan.host.n_ext_agents, dtype=torch.float32, device=dev, generator=g))
    names = [n.name for n in specs]
    p = engine.params(now=1.0, delta_time=1.0, day_type=0, active=names, betas=dict.fromkeys(names, 1.0))
    io = engine.io(not_infected_probs=torch.empty(plan.host.n_agents, dtype=torch.float32, device=dev))
    best = float("inf")
    for rep in range(4):             # a warm-up loop, then the fastest of three: one host hiccup must not decide the race
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps if rep else 2):
            for phase in (1, 2, 4):
                engine.step_phase(bufs, p, io, phase)
        torch.cuda.synchronize(dev)
        if rep:
            best = min(best, 1e3 * (time.perf_counter() - t0) / steps)
    return best


def _np(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def require_hip(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(
            f"grad_june_amd computes the infection path on a HIP device only; got device={device}. "
            "There is no CPU or eager-PyTorch fallback (set system.device: cuda:0)."
        )
    _check_device_once(device)
    return device


_DEVICE_OK = set()


def _check_device_once(device: torch.device) -> None:
    """gj_check_device on first use of a device: the library's kernels exist for gfx950 (MI355X) only."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _DEVICE_OK:
        return
    from . import _native as N

    with torch.cuda.device(idx):
        N.check(N.load().gj_check_device(), f"gj_check_device (cuda:{idx})")
    _DEVICE_OK.add(idx)


def edge_sets_of(data, set_names: Sequence[str]) -> Dict[str, dict]:
    """Forward COO of every requested edge set present in ``data`` (reference format)."""
    out = {}
    for s in set_names:
        key = "attends_" + s
        if key not in data:
            continue
        ei = data[key].edge_index
        out[s] = {"agent": _np(ei[0]), "venue": _np(ei[1]), "people": _np(data[s]["people"]), "_ei": ei}
    return out


def _signature(data, set_names, networks) -> Tuple:
    sig = []
    for s in set_names:
        key = "attends_" + s
        if key in data:
            ei = data[key].edge_index
            people = data[s]["people"]
            sig.append((s, ei.data_ptr(), tuple(ei.shape),
                        people.data_ptr() if isinstance(people, torch.Tensor) else id(people)))
    nets = tuple((n.name, n.edge_set, n.mask_kind, None if n.table is None else n.table.tobytes())
                 for n in networks)
    return tuple(sig), nets, len(data["agent"]["id"])


def engine_for(data, specs: Sequence[NetworkSpec], device) -> InfectionEngine:
    """Compile (or fetch the cached) plan for ``data`` and the configured networks."""
    device = require_hip(device)
    set_names = []
    for n in specs:
        if n.edge_set not in set_names:
            set_names.append(n.edge_set)
    sig = _signature(data, set_names, specs) + (str(device), DEFAULT_LAYOUT)
    per_data = _CACHE.get(data)
    if per_data is None:
        per_data = _CACHE[data] = {}
    hit = per_data.get(sig)
    if hit is not None:
        return hit
    if not specs:
        # a launch that needs no edge set (transmission profile) can ride on any plan of this world
        for other_sig, eng in per_data.items():
            if other_sig[2] == sig[2] and other_sig[3:] == sig[3:]:
                return eng
    sets = edge_sets_of(data, set_names)
    agent = data["agent"]
    n_agents = len(agent["id"])
    age = _np(agent["age"]) if "age" in agent else None
    sex = _np(agent["sex"]) if "sex" in agent else None
    if age is None and any(n.table is not None for n in specs):
        raise KeyError("leisure networks need data['agent'].age and .sex")
    on_device = all(isinstance(v["_ei"], torch.Tensor) and v["_ei"].device.type == "cuda" for v in sets.values())
    use_device = DEVICE_COMPILE in ("1", True) or (DEVICE_COMPILE == "auto" and on_device and len(sets) > 0)
    if use_device and DEFAULT_LAYOUT == "tiled":
        on_dev = {k: {"agent": v["_ei"][0], "venue": v["_ei"][1], "people": v["people"]} for k, v in sets.items()}
        n_edges = sum(int(v["_ei"].shape[1]) for v in sets.values())
        best = None
        if TUNE == "auto" and TUNE_MIN_EDGES <= n_edges <= TUNE_MAX_EDGES and len(specs) <= N_MAX_TUNE_NETS:
            with torch.cuda.device(device):
                for cand in TUNE_CANDIDATES:
                    h = compile_plan(n_agents, on_dev, age=age, sex=sex, layout="tiled", device=device,
                                     runs=_runs_option(), **cand)
                    e = InfectionEngine(DevicePlan(h, [n for n in specs if n.edge_set in h.set_index], device))
                    ms = _time_passes(e, [n for n in specs if n.edge_set in h.set_index])
                    if best is None or ms < best[0]:
                        best = (ms, h, e)
                    del h, e
        if best is not None:
            host, engine = best[1], best[2]
            if len(per_data) >= 4:
                per_data.pop(next(iter(per_data)))
            per_data[sig] = engine
            return engine
        host = compile_plan(n_agents, on_dev, age=age, sex=sex, layout="tiled", device=device, runs=_runs_option())
    else:
        host = compile_plan(n_agents, {k: {kk: vv for kk, vv in v.items() if kk != "_ei"} for k, v in sets.items()},
                            age=age, sex=sex, layout=DEFAULT_LAYOUT, runs=_runs_option())
    present = [n for n in specs if n.edge_set in host.set_index]
    engine = InfectionEngine(DevicePlan(host, present, device))
    if len(per_data) >= 4:          # worlds whose edges are rebuilt repeatedly: keep the cache small
        per_data.pop(next(iter(per_data)))
    per_data[sig] = engine
    return engine


def _f32(store, key, device, n):
    """Fetch a per-agent tensor as contiguous float32 on ``device`` (converted in place in the store)."""
    t = store[key]
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if t.dtype != torch.float32 or t.device != device or not t.is_contiguous():
        t = t.detach().to(device=device, dtype=torch.float32).contiguous()
        store[key] = t
    if t.numel() != n:
        raise ValueError(f"agent.{key}: expected {n} values, got {t.numel()}")
    return t


def agent_buffers(engine: InfectionEngine, data, *, need_params: bool, need_stage: bool,
                  need_infection_state: bool = True) -> AgentBuffers:
    """Device views of ``data['agent']`` for one launch.  State tensors are used IN PLACE."""
    dev = engine.plan.device
    ag = data["agent"]
    n = engine.plan.host.n_agents
    kw = {}
    if need_params:
        ip = ag["infection_parameters"]
        for k in ("max_infectiousness", "shape", "rate", "shift"):
            kw[k] = _f32(ip, k, dev, n)
    for k in ("infection_time", "is_infected", "susceptibility"):
        if need_infection_state or k == "susceptibility":
            kw[k] = _f32(ag, k, dev, n)
        else:
            kw[k] = None
    if "transmission" not in ag:
        ag["transmission"] = torch.zeros(n, dtype=torch.float32, device=dev)
    kw["transmission"] = _f32(ag, "transmission", dev, n)
    if need_stage:
        stage = ag["symptoms"]["current_stage"]
        if stage.dtype != torch.float32 or stage.device != dev or not stage.is_contiguous():
            stage = stage.detach().to(device=dev, dtype=torch.float32).contiguous()   # copy for this call
        kw["current_stage"] = stage
    return AgentBuffers(engine.plan, **kw)
