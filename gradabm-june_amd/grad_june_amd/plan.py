"""One-time graph compile: the reference's unsorted int64 COO edge lists -> an immutable plan.

Input is the data surface the reference's ``june_world_loader`` emits:
per venue type ``data["attends_<set>"].edge_index`` (int64 ``[2, E]``, row 0 = agent, row 1 =
venue, unsorted; /root/reference/grad_june/june_world_loader/network_loader.py:30-44) and
``data[<set>].people``.  Output, per edge set, is what the gfx950 kernels stream:

* CSR by venue  (``v_rowptr``, ``v_agent``)  - pass 1 reads the edge list venue-major, coalesced;
* CSR by agent  (``a_rowptr``, ``a_venue``)  - pass 2 reads it agent-major;
* ``v_pcontact`` = clamp(1/(people-1), 0, 1) (reference base.py:63-69), static, computed once;
* a workgroup schedule for pass 1 (``gj_block``): consecutive venues packed into blocks of at most
  2048 edges ("STREAM": LDS-staged, 1/4/16/64 lanes per venue chosen from the block's mean degree)
  and venues above 2048 edges cut into chunks of ``LONG_CHUNK`` edges ("LONG": multi-workgroup,
  partial sums combined in chunk order by a second kernel - deterministic, no float atomics).

Both CSR views keep the COO order inside a row (stable sort), so a sum taken in row order has the
reference's ``scatter_add_`` order.  All indices are int32.

Host part is numpy/torch-CPU only and is covered by the CPU test-suite; :class:`DevicePlan`
uploads it and owns the per-step workspaces.
"""
from __future__ import annotations

import os

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native as N
from . import tiling as TL

LONG_CHUNK = 8192        # edges per workgroup for venues above the stream capacity
MAX_ROWS_PER_BLOCK = 2048


@dataclass
class HostEdgeSet:
    name: str
    n_venues: int
    n_edges: int
    v_rowptr: Optional[np.ndarray]   # int32 [V+1]   (None in a tiled-only plan)
    v_agent: Optional[np.ndarray]    # int32 [E]
    v_pcontact: np.ndarray           # float32 [V]
    a_rowptr: Optional[np.ndarray]   # int32 [A+1]
    a_venue: Optional[np.ndarray]    # int32 [E]
    tiled: Optional[TL.TiledEdgeSet] = None
    max_venue_edges: int = 0         # edges of the set's largest venue: bounds the terms of one venue sum (gj_tiled_set)
    max_agent_edges: int = 0         # edges of the owned agent with the most edges in this set


def p_contact(people) -> np.ndarray:
    """clamp(1/(people-1), 0, 1) in the reference's dtype flow (int64 or float people -> fp32)."""
    people = torch.as_tensor(np.asarray(people)) if not isinstance(people, torch.Tensor) else people.cpu()
    pc = torch.clamp(1.0 / (people - 1), max=1.0)
    pc = torch.clamp(pc, min=0.0)
    return pc.to(torch.float32).numpy()


def _csr(rows: np.ndarray, cols: np.ndarray, n_rows: int):
    """Stable CSR of (rows -> cols): rowptr int32 [n_rows+1], cols reordered row-major."""
    order = np.argsort(rows, kind="stable")
    counts = np.bincount(rows, minlength=n_rows)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr.astype(np.int32), cols[order].astype(np.int32)


def compile_edge_set(name: str, agent_index, venue_index, people, n_agents: int,
                     n_ext_agents: Optional[int] = None, csr: bool = True) -> HostEdgeSet:
    """COO -> two CSR views.  ``agent_index`` may reference halo agents in [n_agents, n_ext)."""
    agent_index = np.asarray(agent_index).astype(np.int64, copy=False).ravel()
    venue_index = np.asarray(venue_index).astype(np.int64, copy=False).ravel()
    n_ext = n_agents if n_ext_agents is None else n_ext_agents
    n_venues = int(len(people))
    E = int(agent_index.shape[0])
    if E >= 2**31 or n_ext >= 2**31 or n_venues >= 2**31:
        raise ValueError("edge set exceeds int32 indexing")
    if E:
        if agent_index.min() < 0 or agent_index.max() >= n_ext:
            raise ValueError(f"{name}: agent index out of range")
        if venue_index.min() < 0 or venue_index.max() >= n_venues:
            raise ValueError(f"{name}: venue index out of range")
    owned = agent_index < n_agents
    mve = int(np.bincount(venue_index, minlength=1).max()) if E else 0
    mae = int(np.bincount(agent_index[owned], minlength=1).max()) if E and owned.any() else 0
    if not csr:
        return HostEdgeSet(name, n_venues, E, None, None, p_contact(people), None, None, max_venue_edges=mve,
                           max_agent_edges=mae)
    v_rowptr, v_agent = _csr(venue_index, agent_index, n_venues)
    a_rowptr, a_venue = _csr(agent_index[owned], venue_index[owned], n_agents)
    return HostEdgeSet(name, n_venues, E, v_rowptr, v_agent, p_contact(people), a_rowptr, a_venue, max_venue_edges=mve,
                       max_agent_edges=mae)


def _host(a) -> np.ndarray:
    """numpy view of an array that may be a torch tensor (possibly on a device)."""
    if hasattr(a, "detach"):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def _edge_set_on_device(name: str, es: dict, n_ext: int, n_agents: Optional[int] = None) -> HostEdgeSet:
    """compile_edge_set(csr=False) for a set whose edge lists stay on the device (the index range checks are part of
    the compile kernels, tiling_native)."""
    people = _host(es["people"])
    n_venues = int(len(people))
    E = int(es["agent"].numel() if hasattr(es["agent"], "numel") else np.asarray(es["agent"]).size)
    if E >= 2**31 or n_ext >= 2**31 or n_venues >= 2**31:
        raise ValueError("edge set exceeds int32 indexing")
    mve = mae = 0
    if E and isinstance(es["venue"], torch.Tensor):
        mve = int(torch.bincount(es["venue"].reshape(-1), minlength=1).max())
        a = es["agent"].reshape(-1)
        if n_agents is not None:
            a = a[a < n_agents]
        mae = int(torch.bincount(a, minlength=1).max()) if a.numel() else 0
    elif E:
        mve = int(np.bincount(np.asarray(es["venue"]).ravel(), minlength=1).max())
        a = np.asarray(es["agent"]).ravel()
        a = a[a < (n_ext if n_agents is None else n_agents)]
        mae = int(np.bincount(a, minlength=1).max()) if len(a) else 0
    return HostEdgeSet(name, n_venues, E, None, None, p_contact(people), None, None, max_venue_edges=mve, max_agent_edges=mae)


def _lanes_for(mean_degree: float) -> int:
    if mean_degree <= 6:
        return 1
    if mean_degree <= 24:
        return 4
    if mean_degree <= 96:
        return 16
    return 64


def build_schedule(v_rowptr: np.ndarray, set_id: int, slot_base: int = 0,
                   stream_edges: int = N.GJ_STREAM_EDGES, long_chunk: int = LONG_CHUNK):
    """Pass-1 workgroup schedule of one edge set.

    Returns (blocks int32 [nb, 8], long_rows int32 [nl, 4], n_slots).  Block columns follow
    ``gj_block``: set, kind, v0, v1, e0, e1, slot, lanes.  Every venue is covered exactly once:
    by one STREAM block or by the LONG chunks of its own row.
    """
    rp = v_rowptr.astype(np.int64)
    V = len(rp) - 1
    deg = np.diff(rp)
    blocks: List[tuple] = []
    long_rows: List[tuple] = []
    slot = slot_base
    long_idx = np.flatnonzero(deg > stream_edges)
    bounds = np.concatenate(([-1], long_idx, [V]))
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i] + 1, bounds[i + 1]      # stream segment [lo, hi)
        v = lo
        while v < hi:
            # largest end with rp[end]-rp[v] <= stream_edges, end <= hi, rows <= MAX_ROWS
            end = int(np.searchsorted(rp, rp[v] + stream_edges, side="right")) - 1
            end = min(end, hi, v + MAX_ROWS_PER_BLOCK)
            ne = int(rp[end] - rp[v])
            blocks.append((set_id, 0, v, end, int(rp[v]), int(rp[end]), 0, _lanes_for(ne / (end - v))))
            v = end
        if hi < V:                                   # the long venue that ends this segment
            e0, e1 = int(rp[hi]), int(rp[hi + 1])
            s0 = slot
            for c0 in range(e0, e1, long_chunk):
                blocks.append((set_id, 1, hi, hi + 1, c0, min(c0 + long_chunk, e1), slot, 64))
                slot += 1
            long_rows.append((set_id, hi, s0, slot))
    b = np.array(blocks, dtype=np.int32).reshape(-1, 8)
    lr = np.array(long_rows, dtype=np.int32).reshape(-1, 4)
    return b, lr, slot - slot_base


#: suffix of the partial-sum half of an edge set that the multi-GPU partition split in two (distributed.mode_of
#: "split"), and of the twin networks that run on it
SPLIT_SUFFIX = "~big"


@dataclass
class NetworkSpec:
    """One configured infection network (reference: an ``InfectionNetwork`` subclass instance)."""
    name: str
    edge_set: str
    mask_kind: int
    table: Optional[np.ndarray] = None   # [2,2,100] leisure probabilities, or None


@dataclass
class HostPlan:
    n_agents: int
    n_ext_agents: int
    sets: List[HostEdgeSet]
    agent_class: np.ndarray              # uint8 [n_ext]
    blocks: np.ndarray                   # int32 [nb, 8]
    long_rows: np.ndarray                # int32 [nl, 4]
    n_partial_slots: int
    set_index: Dict[str, int] = field(default_factory=dict)
    layout: str = "csr"
    n_slices: int = 0
    slice_agents: int = 0
    work: Optional[np.ndarray] = None    # int32 [n_work, 2] (set, block), heaviest first (tiled)

    @property
    def n_edges(self) -> int:
        return sum(s.n_edges for s in self.sets)


def save_plan(plan: HostPlan, path) -> None:
    """Write a compiled plan as one .npz (the native on-disk form of a world: compile once, reload in
    seconds; arrays are stored uncompressed so that ``np.load(..., mmap_mode="r")`` can map them)."""
    out = {"meta/n_agents": plan.n_agents, "meta/n_ext_agents": plan.n_ext_agents, "meta/layout": plan.layout,
           "meta/n_slices": plan.n_slices, "meta/slice_agents": plan.slice_agents,
           "meta/n_partial_slots": plan.n_partial_slots, "meta/sets": ",".join(s.name for s in plan.sets),
           "agent_class": plan.agent_class, "blocks": plan.blocks, "long_rows": plan.long_rows}
    if plan.work is not None:
        out["work"] = plan.work
    for s in plan.sets:
        p = f"set/{s.name}/"
        out[p + "n_venues"], out[p + "n_edges"], out[p + "v_pcontact"] = s.n_venues, s.n_edges, s.v_pcontact
        out[p + "max_edges"] = np.array([s.max_venue_edges, s.max_agent_edges], dtype=np.int64)
        for k in ("v_rowptr", "v_agent", "a_rowptr", "a_venue"):
            if getattr(s, k) is not None:
                out[p + k] = getattr(s, k)
        if s.tiled is not None:
            t = s.tiled
            for k in ("blk_v0", "blk_e0", "e_lv", "e_cls", "a_la", "tile_sptr", "tile_jpos", "chunk_ptr", "chunk_desc", "ell",
                      "slot_idx", "multi_slots"):
                if getattr(t, k) is not None:
                    a = _host(getattr(t, k))
                    out[p + "tiled/" + k] = a.view(np.uint16) if (k in ("e_lv", "a_la", "ell") and a.dtype == np.int16) else a
            if t.runs is not None:
                for k in ("vmin", "pv_blk", "pv_win", "blk_r0", "win_lo", "win_n"):
                    a = _host(getattr(t.runs, k))
                    out[p + "tiled/runs/" + k] = a.view(np.uint16) if a.dtype == np.int16 else a
                out[p + "tiled/runs/meta"] = np.array([t.runs.n_primary, t.runs.max_window], dtype=np.int64)
            out[p + "tiled/meta"] = np.array([t.n_slices, t.n_blocks, t.n_slots, int(t.desc_wide), int(t.ell_k),
                                              int(t.presum)], dtype=np.int64)
    np.savez(path, **out)


def load_plan(path) -> HostPlan:
    with np.load(path, allow_pickle=False) as z:
        a = {k: z[k] for k in z.files}
    sets = []
    for name in str(a["meta/sets"]).split(","):
        if not name:
            continue
        p = f"set/{name}/"
        hs = HostEdgeSet(name, int(a[p + "n_venues"]), int(a[p + "n_edges"]), a.get(p + "v_rowptr"), a.get(p + "v_agent"),
                         a[p + "v_pcontact"], a.get(p + "a_rowptr"), a.get(p + "a_venue"))
        if p + "max_edges" in a:
            hs.max_venue_edges, hs.max_agent_edges = (int(x) for x in a[p + "max_edges"])
        if p + "tiled/meta" in a:
            meta = [int(x) for x in a[p + "tiled/meta"]]
            S, J, n_slots, wide = meta[:4]
            ell_k = meta[4] if len(meta) > 4 else 0
            hs.tiled = TL.TiledEdgeSet(name=name, n_venues=hs.n_venues, n_edges=hs.n_edges, n_slices=S, n_blocks=J,
                                       blk_v0=a[p + "tiled/blk_v0"], blk_e0=a[p + "tiled/blk_e0"], e_lv=a[p + "tiled/e_lv"],
                                       e_cls=a.get(p + "tiled/e_cls"), a_la=a[p + "tiled/a_la"],
                                       tile_sptr=a[p + "tiled/tile_sptr"], tile_jpos=a[p + "tiled/tile_jpos"],
                                       v_pcontact=hs.v_pcontact, n_slots=n_slots, chunk_ptr=a[p + "tiled/chunk_ptr"],
                                       chunk_desc=a[p + "tiled/chunk_desc"], desc_wide=bool(wide),
                                       ell=a.get(p + "tiled/ell"), ell_k=ell_k, slot_idx=a.get(p + "tiled/slot_idx"),
                                       multi_slots=a.get(p + "tiled/multi_slots"),
                                       presum=bool(meta[5]) if len(meta) > 5 else False)
            if p + "tiled/runs/meta" in a:
                r = p + "tiled/runs/"
                hs.tiled.n_edges = int(len(a[p + "tiled/a_la"]))       # the tiled arrays hold the non-primary edges
                hs.tiled.runs = TL.RunForm(n_primary=int(a[r + "meta"][0]), keep=None, vmin=a[r + "vmin"],
                                           pv_blk=a[r + "pv_blk"], pv_win=a[r + "pv_win"], blk_r0=a[r + "blk_r0"],
                                           win_lo=a[r + "win_lo"], win_n=a[r + "win_n"], max_window=int(a[r + "meta"][1]))
        sets.append(hs)
    return HostPlan(int(a["meta/n_agents"]), int(a["meta/n_ext_agents"]), sets, a["agent_class"], a["blocks"],
                    a["long_rows"], int(a["meta/n_partial_slots"]), {s.name: i for i, s in enumerate(sets)},
                    layout=str(a["meta/layout"]), n_slices=int(a["meta/n_slices"]),
                    slice_agents=int(a["meta/slice_agents"]), work=a.get("work"))


def agent_class_of(age, sex) -> np.ndarray:
    age = np.asarray(age).astype(np.int64)
    sex = np.asarray(sex).astype(np.int64)
    if age.size and (age.min() < 0 or age.max() > 99 or sex.min() < 0 or sex.max() > 1):
        raise ValueError("age must be in 0..99 and sex in {0,1} (leisure tables are [2,2,100])")
    return (sex * 100 + age).astype(np.uint8)


def compile_plan(n_agents: int, edge_sets: Dict[str, dict], age=None, sex=None,
                 n_ext_agents: Optional[int] = None, block_order: str = "interleave",
                 layout: str = "csr", leisure_sets: Sequence[str] = ("leisure",),
                 sv_max: int = TL.SV_MAX, eb_target: Optional[int] = None, slices=None, tile_pad: int = 1,
                 nets_per_set: Optional[Dict[str, int]] = None, progress=None,
                 desc_wide: Optional[bool] = None, device=None, direct=None, runs=None, presum=None,
                 desc_explicit: Optional[bool] = None, multi_net_block_div: Optional[int] = None) -> HostPlan:
    """edge_sets: {name: {"agent": i64[E], "venue": i64[E], "people": [V]}} (insertion order = set ids).

    layout: "csr" (deterministic CSR kernels), "tiled" (LDS-tiled fast path) or "both".
    nets_per_set: infection networks that may be active on a set at once (a venue block keeps one
    8-byte LDS sum per venue and network; default 1, and 6 for the leisure sets).
    desc_wide: chunk descriptor format of the tiled layout (None: per set, from its tile sizes).
    device: build the tiled arrays on this HIP device with the library's compile kernels (tiling_native /
    csrc/gj_compile.hip: the same arrays, born in HBM); the edge lists may then be torch tensors.  Default: numpy on
    the host (tiling.py, the specification).
    direct: which sets take pass 2 in the "direct" form (tiling.build_ell: phase C skipped, phase D reads the
    venues' cum from an LDS table).  None = every set whose sizes allow it (tiling.direct_eligible), False =
    none, or a collection of set names (must be eligible).
    runs: which sets keep one edge per owned agent in the "run form" (tiling.split_primary_runs: the set the agents
    are ordered by - households under graph.locality_order - needs no index arrays for those edges).  None = every
    single-network set whose agents are so ordered, that is not in the direct form and where it pays; False = none;
    or a collection of set names (must be possible; takes precedence over the direct form).
    presum: True = a set in the direct form whose edges all belong to owned agents takes pass 1 in the direct form too
    (its ELL rows + per-workgroup LDS tables of fixed-point sums, k_tile_presum; bit-identical results).  Off by
    default: measured on C3 (round 3, tools/ab.py) it moves 0.2 GB less per step and is 15 % SLOWER - the 64-bit LDS
    atomics it takes out of the venue launch (115 M per step) cost the same there, and its own launch pays a memory
    round trip per batch (one workgroup per CU: the tables fill the LDS).  A second attempt in the same round (batches
    software-pipelined, a parallel table reduction) brought it to 8 % slower: still opt-in.
    multi_net_block_div: the edges per venue block of a set that carries several networks (the leisure sets) are
    ``eb_target`` divided by this (None: GJ_MULTI_NET_BLOCK_DIV or 1): a venue-block workgroup sums every slot once per
    network, so such a block is nets times as heavy as its edge count says.
    tile_pad: experiment of the numpy compile (tiling.build_tiled): tiles padded to a multiple of this many positions in
    both orders (16 = whole 64-byte sectors of the workspace); measured -1 % on C3, not adopted, not in the device compile.
    """
    if len(edge_sets) > N.GJ_MAX_SETS:
        raise ValueError(f"at most {N.GJ_MAX_SETS} edge sets")
    if layout not in ("csr", "tiled", "both"):
        raise ValueError(layout)
    n_ext = n_agents if n_ext_agents is None else n_ext_agents
    want_csr, want_tiled = layout in ("csr", "both"), layout in ("tiled", "both")
    cls_all = None if age is None else agent_class_of(age, sex)
    S, SA = slices if slices is not None else TL.choose_slices(n_ext)
    if eb_target is None:
        eb_target = TL.choose_block_edges(S)
    if multi_net_block_div is None:
        multi_net_block_div = int(os.environ.get("GJ_MULTI_NET_BLOCK_DIV", "1"))
    if want_tiled and n_ext != n_agents and (-(-n_agents // SA)) * SA > n_ext:
        raise ValueError("halo agents must start on a slice boundary (pad the owned range to a multiple of SA)")
    sets, all_blocks, all_long, work = [], [], [], []
    slot = 0
    n_presum = 0
    for sid, (name, es) in enumerate(edge_sets.items()):
        if device is not None and not want_csr:
            hs = _edge_set_on_device(name, es, n_ext, n_agents)
        else:
            hs = compile_edge_set(name, _host(es["agent"]), _host(es["venue"]), _host(es["people"]), n_agents, n_ext,
                                  csr=want_csr)
        sets.append(hs)
        if want_tiled:
            k = (nets_per_set or {}).get(name, 6 if name in leisure_sets else 1)
            use_cls = cls_all if (name in leisure_sets and cls_all is not None) else None
            forced_run = runs is not None and runs is not False and name in runs
            plan_direct = None
            if not forced_run:
                plan_direct = _direct_plan(name, es, hs.n_venues, n_agents, SA, k, direct, device)
            rf, es_t = None, es
            if runs is not False and (forced_run or (plan_direct is None and runs is None)) and k == 1 \
                    and name not in leisure_sets:
                rf, es_t = _split_runs(name, es, n_agents, hs.n_venues, SA, device, forced_run)
            if device is not None and tile_pad > 1:
                raise NotImplementedError("tile_pad is an experiment of the numpy compile (compile_plan without device=)")
            eb_set = eb_target if k <= 1 else max(8192, eb_target // max(1, multi_net_block_div))
            if device is not None:
                from .tiling_native import build_tiled_native

                hs.tiled = build_tiled_native(name, es_t["agent"], es_t["venue"], hs.n_venues, hs.v_pcontact, S, SA,
                                              agent_class=use_cls, sv_max=max(16, sv_max // max(1, k)),
                                              eb_target=eb_set, wide=desc_wide, device=device, n_ext_agents=n_ext,
                                              explicit=desc_explicit)
            else:
                hs.tiled = TL.build_tiled(name, es_t["agent"], es_t["venue"], hs.n_venues, hs.v_pcontact, S, SA,
                                          agent_class=use_cls, sv_max=max(16, sv_max // max(1, k)),
                                          eb_target=eb_set, wide=desc_wide, explicit=desc_explicit,
                                          tile_pad=tile_pad)
            t = hs.tiled
            if rf is not None:
                t.runs = _finish_runs(rf, t, n_agents, SA, device)
            elif plan_direct is not None:
                t.ell, t.ell_k = plan_direct()
                t.presum = presum is True and plan_direct.all_owned and n_presum < 6      # (GJ_MAX_PRESUM)
                n_presum += int(t.presum)
            blk_e0, blk_v0 = _host(t.blk_e0).astype(np.int64), _host(t.blk_v0).astype(np.int64)
            prim = np.diff(_host(t.runs.blk_r0).astype(np.int64)) if t.runs is not None else np.zeros(t.n_blocks, np.int64)
            for j in range(t.n_blocks):
                work.append((int(blk_e0[j + 1] - blk_e0[j]) + int(blk_v0[j + 1] - blk_v0[j]) + int(prim[j]), sid, j))
        if progress:
            progress(f"compiled edge set {name}")
        if not want_csr:
            continue
        b, lr, ns = build_schedule(hs.v_rowptr, sid, slot)
        slot += ns
        all_blocks.append(b)
        all_long.append(lr)
    blocks = np.concatenate(all_blocks) if all_blocks else np.zeros((0, 8), np.int32)
    long_rows = np.concatenate(all_long) if all_long else np.zeros((0, 4), np.int32)
    if block_order == "interleave" and len(blocks):
        # LONG chunks first (heaviest workgroups start early), then stream blocks
        order = np.argsort(-(blocks[:, 5] - blocks[:, 4]), kind="stable")
        blocks = blocks[order]
    if age is None:
        cls = np.zeros(n_ext, dtype=np.uint8)
    else:
        cls = cls_all
        if len(cls) != n_ext:
            raise ValueError("age/sex must cover owned + halo agents")
    if want_tiled and sum(s.max_agent_edges for s in sets) > TL.MAX_AGENT_EDGES:
        raise ValueError(f"an agent with more than {TL.MAX_AGENT_EDGES} edges: its pass-2 sum (64-bit fixed point, terms "
                         f"up to 262 144) could leave the 64 bits")
    work.sort(key=lambda w: -w[0])
    order = os.environ.get("GJ_WORK_ORDER", "heavy")       # experiments (bench.py --work-order): the default is heaviest first
    if order == "light":
        work.reverse()
    elif order == "mixed":                                  # heavy, light, heavy, light ...
        h, l = work[: (len(work) + 1) // 2], work[(len(work) + 1) // 2:][::-1]
        work = [w for pair in zip(h, l + [None] * (len(h) - len(l))) for w in pair if w is not None]
    elif order == "set":
        work.sort(key=lambda w: (w[1], w[2]))
    elif order.startswith("stagger") and len(work) > 512:
        # heaviest first, but every third slot of the first round of 256 workgroups goes to one of the LIGHTEST items:
        # the heavy items of the first round all take the same time, end together and leave the dispatcher 256 slots to
        # refill at once (tools/venue_timeline.py: the resident workgroups dip to 107-130 of 256 at ~40 % of the makespan)
        n_light = int(order[7:] or 85)
        light, rest = work[len(work) - n_light:], work[: len(work) - n_light]
        first, later = rest[: 256 - n_light], rest[256 - n_light:]
        mixed, li = [], 0
        step = max(1, len(first) // max(1, n_light))
        for i, w in enumerate(first):
            mixed.append(w)
            if (i + 1) % step == 0 and li < len(light):
                mixed.append(light[li])
                li += 1
        work = mixed + later + light[li:]
    work_arr = np.array([(w[1], w[2]) for w in work], dtype=np.int32).reshape(-1, 2)
    return HostPlan(n_agents, n_ext, sets, cls, np.ascontiguousarray(blocks), long_rows, slot,
                    {s.name: i for i, s in enumerate(sets)}, layout=layout,
                    n_slices=S if want_tiled else 0, slice_agents=SA if want_tiled else 0,
                    work=work_arr if want_tiled else None)


def _direct_plan(name: str, es: dict, n_venues: int, n_agents: int, slice_agents: int, nets: int, direct, device):
    """Decide whether pass 2 of this set runs in the direct form; returns None or a callable that builds its ELL
    table -> (ell, K)."""
    n_edges = int(es["agent"].numel() if hasattr(es["agent"], "numel") else np.asarray(es["agent"]).size)
    if direct is False or n_edges == 0:
        return None
    n_owned_slices = max(1, -(-n_agents // slice_agents))
    builder = None
    if device is not None:
        from .tiling_native import EllBuilder

        builder = EllBuilder(es["agent"], es["venue"], n_venues, n_agents, slice_agents, device)
        e_owned, dmax = builder.degrees()
    else:
        agent = np.asarray(_host(es["agent"]), dtype=np.int64).ravel()
        venue = _host(es["venue"])
        own = agent[agent < n_agents]
        e_owned = int(len(own))
        dmax = int(np.bincount(own, minlength=max(1, n_agents)).max()) if e_owned else 0
    ok = TL.direct_eligible(n_venues, e_owned, n_agents, dmax, nets, slice_agents)
    forced = direct is not None and name in direct
    if forced and not ok:
        raise ValueError(f"edge set {name}: not eligible for the direct form of pass 2")
    if not ok or (direct is not None and not forced):
        return None
    build = (lambda: builder.build(dmax)) if builder is not None else (
        lambda: TL.build_ell(agent, venue, n_agents, n_owned_slices, slice_agents))
    build.all_owned = e_owned == n_edges          # no halo agent attends: the rows hold every edge of the set
    return build


def _split_runs(name: str, es: dict, n_agents: int, n_venues: int, slice_agents: int, device, forced: bool):
    """(RunForm or None, the edge lists that stay in the tiled arrays)."""
    min_share = 0.0 if forced else TL.RUN_MIN_SHARE
    if device is not None:
        from .tiling_native import split_primary_runs_native

        rf, rest = split_primary_runs_native(es["agent"], es["venue"], n_agents, n_venues, slice_agents, device, min_share)
    else:
        agent, venue = _host(es["agent"]), _host(es["venue"])
        rf = TL.split_primary_runs(agent, venue, n_agents, n_venues, slice_agents, min_share)
        rest = None if rf is None else {"agent": np.asarray(agent).ravel()[rf.keep], "venue": np.asarray(venue).ravel()[rf.keep]}
    if rf is None:
        if forced:
            raise ValueError(f"edge set {name}: the run form needs the owned agents ordered by their smallest venue of "
                             f"the set (graph.locality_order), agents without an edge last")
        return None, es
    return rf, dict(es, **rest)


def _finish_runs(rf, t, n_agents: int, slice_agents: int, device):
    if device is not None:
        from .tiling_native import finish_run_form_native

        return finish_run_form_native(rf, t.blk_v0, n_agents, slice_agents, device)
    return TL.finish_run_form(rf, _host(t.blk_v0), n_agents, slice_agents)


class DevicePlan:
    """The plan resident in HBM + the ctypes ``gj_plan`` that points at it."""

    def __init__(self, host: HostPlan, networks: Sequence[NetworkSpec], device, flat_cum_sets: Sequence[str] = (),
                 split_epilogue: bool = False, direct_table_floats: int = 0):
        """flat_cum_sets: edge sets whose ``cum`` workspaces are carved from ONE contiguous buffer
        (``self.flat_cum``) so that a single collective can combine them across ranks."""
        self.host = host
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("grad_june_amd runs the infection path on a HIP device only (no CPU path)")
        self.networks = {n.name: n for n in networks}
        dev = self.device

        def up(a, dtype=None):
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
            if dtype is not None:
                t = t.to(dtype)
            return t.to(dev).contiguous()

        # leisure tables: one [2][200] row block per network that has one
        self.table_index: Dict[str, int] = {}
        tabs = []
        for n in networks:
            if n.table is not None:
                self.table_index[n.name] = len(tabs)
                tabs.append(np.asarray(n.table, dtype=np.float32).reshape(N.GJ_TABLE_SIZE))
        self.tables = up(np.stack(tabs)) if tabs else None

        per_set_nets: Dict[str, int] = {}
        for n in networks:
            if n.edge_set not in host.set_index:
                raise KeyError(f"network {n.name}: edge set {n.edge_set!r} not in the world")
            per_set_nets[n.edge_set] = per_set_nets.get(n.edge_set, 0) + 1
        self.keep = []
        self.cum: List[torch.Tensor] = []
        self.tiled_c = None
        # carved in the order the caller lists the sets (a collective may cover a leading / trailing run)
        by_name = {s.name: s for s in host.sets}
        flat_sizes = {n: max(1, by_name[n].n_venues) * max(1, per_set_nets.get(n, 1))
                      for n in flat_cum_sets if n in by_name}
        self.flat_cum = (torch.zeros(sum(flat_sizes.values()), dtype=torch.float32, device=dev)
                         if flat_sizes else None)
        self.flat_offsets: Dict[str, tuple] = {}
        off = 0
        for n, size in flat_sizes.items():
            self.flat_offsets[n] = (off, size)
            off += size
        plan = N.Plan()
        plan.n_agents = host.n_agents
        plan.n_ext_agents = host.n_ext_agents
        plan.n_sets = len(host.sets)
        for i, s in enumerate(host.sets):
            stride = max(1, per_set_nets.get(s.name, 1))
            if stride > N.GJ_MAX_NETS_PER_SET:
                raise ValueError(f"edge set {s.name}: more than {N.GJ_MAX_NETS_PER_SET} networks")
            t = dict(v_pc=up(s.v_pcontact))
            if s.v_rowptr is not None:
                t.update(v_rowptr=up(s.v_rowptr), v_agent=up(s.v_agent), a_rowptr=up(s.a_rowptr), a_venue=up(s.a_venue))
            if s.name in flat_sizes:
                o, size = self.flat_offsets[s.name]
                cum = self.flat_cum[o:o + size]
            else:
                cum = torch.zeros(max(1, s.n_venues) * stride, dtype=torch.float32, device=dev)
            e = plan.sets[i]
            e.n_venues, e.n_edges = s.n_venues, s.n_edges
            e.v_pcontact = t["v_pc"].data_ptr()
            if s.v_rowptr is not None:
                e.v_rowptr, e.v_agent = t["v_rowptr"].data_ptr(), t["v_agent"].data_ptr()
                e.a_rowptr, e.a_venue = t["a_rowptr"].data_ptr(), t["a_venue"].data_ptr()
            e.cum, e.cum_stride = cum.data_ptr(), stride
            if s.tiled is not None:
                ts = s.tiled
                if self.tiled_c is None:
                    self.tiled_c = N.Tiled()
                    self.tiled_c.n_slices, self.tiled_c.slice_agents = host.n_slices, host.slice_agents
                    self.tiled_c.direct_table_floats = int(direct_table_floats)
                u16 = lambda a: up(a if isinstance(a, torch.Tensor) else a.view(np.int16))   # uint16 bit patterns
                t.update(blk_v0=up(ts.blk_v0), blk_e0=up(ts.blk_e0), e_lv=u16(ts.e_lv), a_la=u16(ts.a_la),
                         tile_sptr=up(ts.tile_sptr), tile_jpos=up(ts.tile_jpos),
                         chunk_ptr=up(ts.chunk_ptr), chunk_desc=up(ts.chunk_desc.reshape(-1)) if len(ts.chunk_desc) else None,
                         val=torch.zeros(max(8, ts.n_slots), dtype=torch.float32, device=dev))
                if ts.e_cls is not None:
                    t["e_cls"] = up(ts.e_cls)
                if ts.ell_k:
                    t["ell"] = u16(ts.ell).reshape(-1)
                c = self.tiled_c.sets[i]
                c.n_blocks = ts.n_blocks
                c.max_block_venues = int(np.diff(_host(ts.blk_v0)).max()) if ts.n_blocks else 0
                c.max_venue_edges = int(min(s.max_venue_edges, 2**31 - 1))
                c.desc_wide = 1 if ts.desc_wide else 0
                if ts.slot_idx is not None and ts.n_edges > 0:      # tiles of a few edges: explicit slots, no descriptors
                    t["chunk_desc"] = up(ts.slot_idx)
                    c.desc_wide = 2
                elif ts.multi_slots is not None and len(ts.multi_slots):
                    t["multi_slots"] = up(ts.multi_slots.reshape(-1))
                    c.multi_slots = t["multi_slots"].data_ptr()
                c.blk_v0, c.blk_e0 = t["blk_v0"].data_ptr(), t["blk_e0"].data_ptr()
                c.e_lv, c.a_la = t["e_lv"].data_ptr(), t["a_la"].data_ptr()
                c.e_cls = N.ptr(t.get("e_cls"))
                c.tile_sptr, c.tile_jpos, c.val = t["tile_sptr"].data_ptr(), t["tile_jpos"].data_ptr(), t["val"].data_ptr()
                c.chunk_ptr, c.chunk_desc = t["chunk_ptr"].data_ptr(), N.ptr(t["chunk_desc"])
                c.ell_k, c.ell = int(ts.ell_k), N.ptr(t.get("ell"))
                if ts.runs is not None:
                    rf = ts.runs
                    pad16 = lambda a: torch.cat([a, torch.full((8,), -1, dtype=a.dtype, device=a.device)])  # 16-byte groups
                    t.update(run_pv_blk=pad16(u16(rf.pv_blk)), run_pv_win=pad16(u16(rf.pv_win)), run_blk_r0=up(rf.blk_r0),
                             run_win_lo=up(rf.win_lo), run_win_n=up(rf.win_n))
                    c.run_pv_blk, c.run_pv_win = t["run_pv_blk"].data_ptr(), t["run_pv_win"].data_ptr()
                    c.run_blk_r0, c.run_win_lo, c.run_win_n = (t["run_blk_r0"].data_ptr(), t["run_win_lo"].data_ptr(),
                                                               t["run_win_n"].data_ptr())
                    c.run_max_window = int(rf.max_window)
                    c.run_tiled_edges = int(ts.n_edges)
            self.keep.append(t)
            self.cum.append(cum)
        self.blocks = up(host.blocks.reshape(-1)) if len(host.blocks) else None
        self.long_rows = up(host.long_rows.reshape(-1)) if len(host.long_rows) else None
        self.partial = (torch.zeros(host.n_partial_slots * N.GJ_MAX_NETS_PER_SET, dtype=torch.float32, device=dev)
                        if host.n_partial_slots else None)
        cls_pad = np.zeros(-(-len(host.agent_class) // 4) * 4 + 4, dtype=np.uint8)   # quads of classes are one load
        cls_pad[: len(host.agent_class)] = _host(host.agent_class)
        self.agent_class = up(cls_pad)
        plan.n_blocks = len(host.blocks)
        plan.n_long_rows = len(host.long_rows)
        plan.n_partial_slots = host.n_partial_slots
        plan.blocks = N.ptr(self.blocks)
        plan.long_rows = N.ptr(self.long_rows)
        plan.partial = N.ptr(self.partial)
        plan.agent_class = N.ptr(self.agent_class)
        plan.tables = N.ptr(self.tables)
        plan.n_tables = len(tabs)
        if self.tiled_c is not None:
            self.work = up(host.work.reshape(-1)) if len(host.work) else None
            self.tiled_c.n_work = len(host.work)
            self.tiled_c.work = N.ptr(self.work)
            self.agent_scratch = (torch.zeros(max(1, host.n_agents), dtype=torch.float32, device=dev)
                                  if split_epilogue else None)
            # pass 1 in the direct form: one table of fixed-point sums per workgroup and set
            wgs = int(min(256, max(1, -(-host.n_agents // 4096))))
            for i, s_ in enumerate(host.sets):
                if s_.tiled is not None and s_.tiled.presum and s_.tiled.ell_k:
                    stride = int(plan.sets[i].cum_stride)
                    buf = torch.zeros(wgs * max(1, s_.n_venues) * stride, dtype=torch.int64, device=dev)
                    self.keep[i]["presum"] = buf
                    self.tiled_c.sets[i].presum = buf.data_ptr()
                    self.tiled_c.presum_wgs = wgs
            self.tiled_c.agent_scratch = N.ptr(self.agent_scratch)
            plan.tiled = C.pointer(self.tiled_c)
        self.c = plan

    def cum_of(self, set_name: str) -> torch.Tensor:
        i = self.host.set_index[set_name]
        stride = self.c.sets[i].cum_stride
        return self.cum[i][: self.host.sets[i].n_venues * stride].view(self.host.sets[i].n_venues, stride)

    def bytes_resident(self) -> int:
        n = 0
        for t in self.keep:
            n += sum(x.numel() * x.element_size() for x in t.values() if x is not None)
        n += sum(c.numel() * 4 for c in self.cum)
        for x in (self.blocks, self.long_rows, self.partial, self.agent_class, self.tables):
            if x is not None:
                n += x.numel() * x.element_size()
        return n
