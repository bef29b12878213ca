"""Multi-GPU hot path: agents partitioned over the ranks of one node, one process per GPU.

The reference is single-process; this is new design.  Everything per-agent
(a1, a2, a4, a7-a9) is local to the rank that owns the agent.  Only pass 1 needs remote data, and
per edge set one of two exchanges is used (decided once, identically on every rank):

* ``halo``    (small venues: households, ...): every venue with a local attendee is computed in full
  on this rank, so the transmissions of its REMOTE attendees ("halo agents") are received once per
  step by ONE ``all_to_all_single`` with per-peer split sizes (RCCL over xGMI: direct peer-to-peer
  sends on all links at once, no ring).  Halo agents extend the local agent index range; the
  receive buffer IS the halo part of the transmission array (no unpack).
* ``partial`` (venues that span ranks: schools, leisure, ...): each rank sums its own attendees and
  the per-venue partial sums of all such sets - one flat fp32 buffer - are combined by ONE
  ``all_reduce``; volume is bounded by the number of venues, not by the attendees.
A heavy-tailed set (power-law venue sizes) is ``split`` into a halo half (its small venues) and a partial-sum half
(its large ones), see ``mode_of``.

Round 4: the exchange is decided PER VENUE (``classify_venues``): a venue whose attendees all live on one
rank is *local* (computed there, never communicated), one with few remote attendees is a *halo* venue, one that spans
ranks / is large goes the *partial-sum* way - and only those venues are in the all-reduce buffer.  A set that has both
kinds is cut in two (the "split" form: ``<set>`` keeps the local + halo venues, ``<set>~big`` the partial-sum ones);
the per-set rule of rounds 1-3 (``mode_of``) remains as ``GJ_EXCHANGE_RULE=set`` for comparison.

Per step:  transmission -> [all_to_all halo] -> phase A, phase B -> [all_reduce partial sums]
           -> phase C, phase D.   Pass 2 and the epilogue are purely local.
Sampling noise is Philox keyed by the GLOBAL agent id, so results do not depend on the partition.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import tiling as TL
from .plan import SPLIT_SUFFIX

HALO_MAX_MEAN_DEGREE = 8.0     # sets whose venues average more attendees use partial sums
SPLIT_MIN_WEIGHTED_SIZE = 64.0  # venue size seen by the average EDGE of a small-mean set above which the set is split
PIPELINE_MIN_EDGES = 20_000_000  # set-edges per rank from which the largest partial-sum set gets its own all-reduce
#: "venue" (default): per-venue classes from the venue's own attendees (classify_venues); "set": the per-set rule of
#: rounds 1-3 (mode_of: global sizes only) - kept so that tools/rank_share.py can measure one against the other
EXCHANGE_RULE = os.environ.get("GJ_EXCHANGE_RULE", "venue")
PARTIAL_COST_FACTOR = 1.0        # a venue goes the partial-sum way when its halo floats exceed this x its all-reduce floats
MIN_SPLIT_HALO_FLOATS = 4096     # ... and a set is only cut in two when that saves at least this many halo floats per step
MIN_SPLIT_HALO_SHARE = 0.05      # ... and at least this share of the halo floats of the whole set (a household set that
                                 # loses 1 % of its venues to a partial-sum half loses its run form for nothing)
MIN_SPLIT_VENUE_SHARE = 0.10     # ... and takes at least this share of the set's venues out of the all-reduce


def classify_venues(agent: np.ndarray, venue: np.ndarray, n_venues: int, bounds: np.ndarray, nets_on_set: int = 1,
                    factor: float = PARTIAL_COST_FACTOR):
    """Exchange class of every venue of one edge set, from the venue's own attendee list (global data: every rank
    computes the same answer).  With n attendees living on T ranks:

      T <= 1                 *local*: the owning rank computes it, nothing is communicated;
      halo floats n (T - 1)  what the halo form moves per step (every touching rank receives the attendees it does not
                             own) against the 2 k (R - 1) floats a venue with k networks costs in an all-reduce over
                             R ranks: *halo* while n (T - 1) <= factor * 2 k (R - 1) - and always up to
                             HALO_MAX_MEAN_DEGREE (8) attendees -, else *partial-sum*.

    A boundary household is a halo venue, a school with one pupil from the next rank or a 50 000-attendee venue a
    partial-sum one.  Returns (partial[V] bool, n[V], T[V])."""
    R = len(bounds) - 1
    if isinstance(agent, torch.Tensor):          # the same on the device (a streamed 1e8-agent world never leaves it)
        b = torch.as_tensor(np.asarray(bounds), device=agent.device)
        owner = torch.searchsorted(b, agent, right=True) - 1
        cnt = torch.bincount(venue * R + owner, minlength=n_venues * R).view(n_venues, R)
        n = cnt.sum(1)
        T = (cnt > 0).sum(1)
        del cnt, owner
        return (n > HALO_MAX_MEAN_DEGREE) & (n * (T - 1) > factor * 2.0 * max(1, nets_on_set) * (R - 1)), n, T
    owner = np.searchsorted(bounds, agent, side="right") - 1
    cnt = np.bincount(venue * R + owner, minlength=n_venues * R).reshape(n_venues, R)
    n = cnt.sum(1)
    T = (cnt > 0).sum(1)
    del cnt
    return (n > HALO_MAX_MEAN_DEGREE) & (n * (T - 1) > factor * 2.0 * max(1, nets_on_set) * (R - 1)), n, T


def reduce_groups(floats: Dict[str, int], min_floats: int = 1 << 16) -> List[List[str]]:
    """Partial-sum edge sets -> all-reduce groups.  ``floats``: cum-buffer length per set (venues x networks;
    global numbers, so every rank derives the same grouping).  The largest set gets an all-reduce of its
    own, issued first, when it is big enough to be worth overlapping with the other sets' phases."""
    names = sorted(floats, key=lambda n: (-floats[n], n))
    if len(names) >= 2 and floats[names[0]] >= min_floats:
        return [[names[0]], names[1:]]
    return [names] if names else []


def partition_bounds(n_agents: int, world_size: int) -> np.ndarray:
    """Contiguous, near-equal agent ranges: rank r owns [b[r], b[r+1])."""
    return (np.arange(world_size + 1, dtype=np.int64) * n_agents) // world_size


def choose_modes(world: dict, world_size: int, override: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """Exchange mode per edge set - a function of global sizes only, so all ranks agree."""
    modes = {}
    for name, es in world["edge_sets"].items():
        modes[name] = mode_of(len(es["agent"]), len(es["people"]), world_size, es["people"])
    if override:
        modes.update(override)
    return modes


@dataclass
class RankWorld:
    """The part of the world one rank compiles, in its local ("extended") index spaces."""
    rank: int
    world_size: int
    bounds: np.ndarray                 # global partition
    n_local: int
    slice_agents: int
    n_local_pad: int                   # first halo index: n_local rounded up to a slice boundary
    n_ext: int                         # n_local_pad + n_halo
    n_slices: int
    halo_global: np.ndarray            # int64 [n_halo] global ids of the halo agents, grouped by owner; inside an owner's
                                       # group by (first halo set, venue there, id) - see order_halo
    halo_from: np.ndarray              # int64 [R] halo agents received from each peer
    age: np.ndarray                    # [n_ext]
    sex: np.ndarray
    edge_sets: Dict[str, dict]         # local COO: agent = extended index, venue = local/global id
    modes: Dict[str, str]
    venue_global: Dict[str, Optional[np.ndarray]]   # the set's venue id of each local venue (None: the same numbering)

    @property
    def n_halo(self) -> int:
        return len(self.halo_global)


def mode_of(n_edges: int, n_venues: int, world_size: int, people=None) -> str:
    """Exchange mode of ONE edge set from its global sizes (what choose_modes applies to every set).

    "halo" pays per remote attendee of a touched venue, "partial" per venue.  A heavy-tailed set (power-law venue
    sizes, BASELINE config 5: most venues hold one or two agents, most EDGES lie in venues of thousands) is wrong
    for both as a whole - in halo mode one 50 000-attendee venue makes everybody a halo agent of every rank, in
    partial-sum mode the millions of tiny venues become a per-step all-reduce of hundreds of MB - so it is "split"
    (``people`` = the venues' sizes): venues of up to HALO_MAX_MEAN_DEGREE attendees go the halo way, the larger ones
    the partial-sum way, as two edge sets (``RankPartitioner.add_set``)."""
    if world_size == 1:
        return "local"
    if n_edges / max(1, n_venues) > HALO_MAX_MEAN_DEGREE:
        return "partial"
    if people is not None and n_edges:
        sizes = np.asarray(people, dtype=np.float64)
        if float((sizes * sizes).sum() / max(1.0, sizes.sum())) > SPLIT_MIN_WEIGHTED_SIZE \
                and (sizes > HALO_MAX_MEAN_DEGREE).any() and (sizes <= HALO_MAX_MEAN_DEGREE).any():
            return "split"
    return "halo"


def _lexsort_torch(keys):
    """``np.lexsort`` for torch tensors: the LAST key is the primary one (successive stable sorts)."""
    order = torch.arange(keys[0].numel(), device=keys[0].device)
    for k in keys:
        order = order[torch.sort(k[order], stable=True)[1]]
    return order


#: "venue" (default): an owner's halo agents are ordered by the halo set and venue that needs them; "id": by agent id
#: (rounds 1-3) - kept for comparison (tools/rank_share.py)
HALO_ORDER = os.environ.get("GJ_HALO_ORDER", "venue")


def order_halo(lists, bounds, device=None):
    """The halo agents of one rank in the order of its extended index range, from the per-set lists of (remote attendee,
    its smallest local venue of the set): grouped by owner (what the all-to-all delivers), and inside an owner's group by
    (first halo set that needs the agent, its venue there, id).  A halo agent has ~1 halo edge; in id order the halo
    slices' tiles hold a handful of edges each (a rank's C5 share spent a third of its step scattering them); in this
    order the agents a venue block needs are neighbours, so a halo slice meets few venue blocks and its tiles are long.
    Returns (ids, owner) as numpy arrays, or as torch tensors when the lists hold tensors."""
    on_torch = bool(lists) and isinstance(lists[0][0], torch.Tensor)
    if not lists:
        if device is not None:
            z = torch.zeros(0, dtype=torch.int64, device=device)
            return z, z.clone()
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    if on_torch:
        ids = torch.cat([a for a, _ in lists])
        ven = torch.cat([v for _, v in lists])
        si = torch.cat([torch.full((a.numel(),), i, dtype=torch.int64, device=a.device) for i, (a, _) in enumerate(lists)])
        o = _lexsort_torch((si, ids))
        ids, ven, si = ids[o], ven[o], si[o]
        first = torch.ones(ids.numel(), dtype=torch.bool, device=ids.device)
        first[1:] = ids[1:] != ids[:-1]
        ids, ven, si = ids[first], ven[first], si[first]
        owner = torch.searchsorted(torch.as_tensor(np.asarray(bounds), device=ids.device), ids, right=True) - 1
        if HALO_ORDER == "venue":
            o = _lexsort_torch((ids, ven, si, owner))
            ids, owner = ids[o], owner[o]
        return ids, owner
    ids = np.concatenate([a for a, _ in lists])
    ven = np.concatenate([v for _, v in lists])
    si = np.concatenate([np.full(len(a), i, dtype=np.int64) for i, (a, _) in enumerate(lists)])
    o = np.lexsort((si, ids))
    ids, ven, si = ids[o], ven[o], si[o]
    first = np.ones(len(ids), dtype=bool)
    first[1:] = ids[1:] != ids[:-1]
    ids, ven, si = ids[first], ven[first], si[first]
    owner = np.searchsorted(bounds, ids, side="right") - 1
    if HALO_ORDER == "venue":
        o = np.lexsort((ids, ven, si, owner))
        ids, owner = ids[o], owner[o]
    return ids, owner


class RankPartitioner:
    """Cuts a world into the parts of one or several ranks, ONE EDGE SET AT A TIME: a set's global COO is needed
    only while ``add_set`` runs, so a rank never holds more than one set of the whole world (a 10^8-agent world
    is ~14 GB of COO; ``bench.py --gpus N`` streams it set by set out of the generator) and a single process that
    wants every rank's part (``PartitionedHotPath``) sorts each set once instead of scanning it once per rank."""

    def __init__(self, n_agents: int, world_size: int, ranks: Optional[Sequence[int]] = None,
                 modes: Optional[Dict[str, str]] = None, networks: Optional[Sequence[str]] = None,
                 n_sets: Optional[int] = None, log=None):
        """``networks``: the infection networks that will run on the world (names; their edge sets by
        ``synthetic.edge_set_of``) and ``n_sets``: how many edge sets will be added - what the split mode needs to
        stay inside the library's limits (a split set becomes two sets and every network on it gets a twin)."""
        from . import _native as N_
        from .synthetic import edge_set_of

        self.n_agents, self.world_size = int(n_agents), int(world_size)
        self._limits = (N_.GJ_MAX_SETS, N_.GJ_MAX_NETS, N_.GJ_MAX_NETS_PER_SET)
        self._nets_on: Dict[str, int] = {}
        for n in (networks or ()):
            self._nets_on[edge_set_of(n)] = self._nets_on.get(edge_set_of(n), 0) + 1
        self._n_nets = sum(self._nets_on.values())            # grows by a set's networks with every split
        self._n_sets_expected = n_sets
        self._n_sets_added = 0                                 # parts, i.e. a split set counts twice
        self._n_sets_seen = 0
        self._log = log
        self.ranks = list(range(world_size)) if ranks is None else [int(r) for r in ranks]
        self.bounds = partition_bounds(self.n_agents, self.world_size)
        self.mode_override = dict(modes or {})
        self.modes: Dict[str, str] = {}
        self.local_sets = {r: {} for r in self.ranks}
        self.halo_lists = {r: [] for r in self.ranks}
        self.venue_global = {r: {} for r in self.ranks}
        self.total_edges = 0
        self.sizes: Dict[str, tuple] = {}          # global (edges, venues) per set
        self.classes: Dict[str, dict] = {}         # per set: venues by exchange class (global counts; reporting)

    def add_set(self, name: str, agent, venue, people) -> str:
        """``agent`` / ``venue`` / ``people``: numpy arrays, or torch tensors (any device) - the part is then cut out with
        torch ops where the tensors live and the rank world's edge lists stay there (``compile_plan(device=...)``)."""
        on_torch = isinstance(agent, torch.Tensor)
        if on_torch:
            agent, venue = agent.to(torch.int64).reshape(-1), venue.to(torch.int64).reshape(-1)
            people = torch.as_tensor(people, device=agent.device)
        else:
            agent = np.asarray(agent, dtype=np.int64).ravel()
            venue = np.asarray(venue, dtype=np.int64).ravel()
            people = np.asarray(people)
        mode = self.mode_override.get(name)
        big = None
        if mode is None and (EXCHANGE_RULE == "set" or self.world_size == 1):
            mode = mode_of(len(agent), len(people), self.world_size,
                           people.cpu().numpy() if on_torch and self.world_size > 1 else (None if on_torch else people))
        elif mode is None:
            mode, big = self._mode_by_venue(name, agent, venue, len(people))
        self.total_edges += len(agent)
        self.sizes[name] = (len(agent), len(people))
        self._n_sets_seen += 1
        if mode == "split":
            mode = self._split_or_fallback(name, len(people))
        self._n_sets_added += 2 if mode == "split" else 1
        if mode == "split":
            # two edge sets with venue numberings of their own: the local + halo venues exchange halo transmissions, the
            # others partial sums; every network on the set gets a twin on the second (expand_split_networks).  (The
            # forced / per-set form of rounds 2-3 cuts by size: venues of up to HALO_MAX_MEAN_DEGREE attendees are halo.)
            if big is None:
                big = people > HALO_MAX_MEAN_DEGREE
            if on_torch and not isinstance(big, torch.Tensor):
                big = torch.as_tensor(big, device=agent.device)
            e_big = big[venue]
            xp = torch if on_torch else np
            for part, sel_v, sel_e, m in ((name, ~big, ~e_big, "halo"), (name + SPLIT_SUFFIX, big, e_big, "partial")):
                remap = xp.cumsum(sel_v, 0) - 1
                ids = torch.nonzero(sel_v).reshape(-1) if on_torch else np.flatnonzero(sel_v)
                self._add_part(part, agent[sel_e], remap[venue[sel_e]], people[sel_v], m, ids)
            return mode
        self._add_part(name, agent, venue, people, mode)
        return mode

    def _mode_by_venue(self, name: str, agent: np.ndarray, venue: np.ndarray, n_venues: int):
        """The set's exchange mode from its venues' classes: "halo" (every venue local or halo), "partial" (next to none
        of them), or "split" with the mask of the partial-sum venues."""
        k = self._nets_on.get(name, 1)
        partial, n, T = classify_venues(agent, venue, n_venues, self.bounds, k)
        used = n > 0
        n_part, n_ex = int((partial & used).sum()), int((~partial & used).sum())
        halo_saved = float((n * (T - 1))[partial].sum()) if n_part else 0.0
        halo_all = float((n * (T - 1)).sum())
        self.classes[name] = {"venues": int(used.sum()), "local": int((T == 1).sum()),
                              "halo": int((~partial & (T > 1)).sum()), "partial_sum": n_part,
                              "edges_in_partial_sum_venues": int(n[partial].sum())}
        if n_part == 0 or halo_saved < max(MIN_SPLIT_HALO_FLOATS, MIN_SPLIT_HALO_SHARE * halo_all):
            mode = "halo"
        elif n_ex < MIN_SPLIT_VENUE_SHARE * (n_part + n_ex):
            mode = "partial"
        else:
            return "split", partial
        # the set runs whole: what its venues' classes are IN EFFECT (the counts by the cost rule are kept beside them)
        c = self.classes[name]
        c["by_cost"] = {k: c[k] for k in ("local", "halo", "partial_sum")}
        if mode == "halo":
            c["halo"], c["partial_sum"], c["edges_in_partial_sum_venues"] = c["halo"] + n_part, 0, 0
        else:
            c["local"], c["halo"], c["partial_sum"] = 0, 0, c["venues"]
            c["edges_in_partial_sum_venues"] = int(n.sum())
        c["set_runs_as"] = mode
        return mode, None

    def _split_or_fallback(self, name: str, n_venues: int) -> str:
        """A split set becomes two edge sets and every network on it gets a twin: only while GJ_MAX_SETS,
        GJ_MAX_NETS and GJ_MAX_NETS_PER_SET hold (six sets split = 12 = GJ_MAX_SETS; the reference's eleven networks
        split everywhere = 22 > 16).  Otherwise the set runs unsplit: partial sums when its cum buffer is no larger
        than one float per agent of the world, else halo - a function of global sizes, so every rank agrees."""
        max_sets, max_nets, max_per_set = self._limits
        k = self._nets_on.get(name, 1)
        still_to_come = (self._n_sets_expected - self._n_sets_seen) if self._n_sets_expected is not None else 0
        fits = (self._n_sets_added + 2 + max(0, still_to_come) <= max_sets and self._n_nets + k <= max_nets
                and k <= max_per_set)
        if fits:
            self._n_nets += k
            return "split"
        mode = "partial" if n_venues * max(1, k) <= self.n_agents else "halo"
        if self._log:
            self._log(f"edge set {name}: not split (the library holds {max_sets} edge sets / {max_nets} networks; "
                      f"{self._n_sets_added} sets and {self._n_nets} networks so far) - runs in {mode} mode")
        return mode

    def _add_part(self, name: str, agent: np.ndarray, venue: np.ndarray, people: np.ndarray, mode: str,
                  part_venues: Optional[np.ndarray] = None) -> None:
        """``part_venues``: the set's venue ids of this part's venues (a split set's half has a numbering of its own);
        ``venue_global`` then maps a rank's local venues to ids of the WHOLE set."""
        self.modes[name] = mode
        b = self.bounds
        if isinstance(agent, torch.Tensor):
            return self._add_part_torch(name, agent, venue, people, mode, part_venues)
        if len(self.ranks) > 2:
            # every rank's own edges with one stable sort by owner (COO order kept inside a rank)
            owner = np.searchsorted(b, agent, side="right") - 1
            order = np.argsort(owner, kind="stable")
            cut = np.searchsorted(owner[order], np.arange(self.world_size + 1))
            mine_idx = {r: order[cut[r]:cut[r + 1]] for r in self.ranks}
            del owner, order
        else:
            mine_idx = {r: np.flatnonzero((agent >= b[r]) & (agent < b[r + 1])) for r in self.ranks}
        for r in self.ranks:
            idx = mine_idx[r]
            if mode in ("partial", "local"):
                self.local_sets[r][name] = {"agent_global": agent[idx], "venue": venue[idx], "people": people}
                self.venue_global[r][name] = part_venues
                continue
            touched = np.zeros(len(people), dtype=bool)
            touched[venue[idx]] = True
            keep = np.flatnonzero(touched[venue])          # every edge of a touched venue, local or remote, COO order
            vg = np.flatnonzero(touched)
            remap = np.full(len(people), -1, dtype=np.int64)
            remap[vg] = np.arange(len(vg))
            ag = agent[keep]
            self.local_sets[r][name] = {"agent_global": ag, "venue": remap[venue[keep]], "people": people[vg]}
            self.venue_global[r][name] = vg if part_venues is None else part_venues[vg]
            rem = (ag < b[r]) | (ag >= b[r + 1])
            ra, rv = ag[rem], self.local_sets[r][name]["venue"][rem]
            o = np.lexsort((rv, ra))                               # per remote attendee: its smallest local venue of this part
            ra, rv = ra[o], rv[o]
            first = np.ones(len(ra), dtype=bool)
            first[1:] = ra[1:] != ra[:-1]
            self.halo_lists[r].append((ra[first], rv[first]))

    def _add_part_torch(self, name, agent, venue, people, mode, part_venues) -> None:
        """``_add_part`` with torch ops on the tensors' device; the part's ``people`` go to the host (the graph
        compile takes p_contact from there), everything per edge stays."""
        b = self.bounds
        V = int(people.numel())
        for r in self.ranks:
            mine = (agent >= int(b[r])) & (agent < int(b[r + 1]))
            if mode in ("partial", "local"):
                self.local_sets[r][name] = {"agent_global": agent[mine], "venue": venue[mine],
                                            "people": people.cpu().numpy()}
                self.venue_global[r][name] = None if part_venues is None else part_venues.cpu().numpy()
                continue
            touched = torch.zeros(V, dtype=torch.bool, device=agent.device)
            touched[venue[mine]] = True
            keep = touched[venue]                                  # every edge of a touched venue, local or remote, COO order
            vg = torch.nonzero(touched).reshape(-1)
            remap = torch.full((V,), -1, dtype=torch.int64, device=agent.device)
            remap[vg] = torch.arange(vg.numel(), device=agent.device)
            ag = agent[keep]
            self.local_sets[r][name] = {"agent_global": ag, "venue": remap[venue[keep]], "people": people[vg].cpu().numpy()}
            self.venue_global[r][name] = (vg if part_venues is None else part_venues[vg]).cpu().numpy()
            rem = (ag < int(b[r])) | (ag >= int(b[r + 1]))
            ra, rv = ag[rem], self.local_sets[r][name]["venue"][rem]
            o = _lexsort_torch((rv, ra))
            ra, rv = ra[o], rv[o]
            first = torch.ones(ra.numel(), dtype=torch.bool, device=ra.device)
            first[1:] = ra[1:] != ra[:-1]
            self.halo_lists[r].append((ra[first], rv[first]))

    def _finish_torch(self, age, sex, slice_agents):
        """``RankPartitioner.finish`` for parts cut out on a device: the extended indices are computed there, the edge lists
        stay there; the halo list, age and sex of the extended range come to the host (small)."""
        out = {}
        for r in self.ranks:
            a0, a1 = int(self.bounds[r]), int(self.bounds[r + 1])
            n_local = a1 - a0
            hl = self.halo_lists[r]
            dev = None
            for ls in self.local_sets[r].values():
                if isinstance(ls["agent_global"], torch.Tensor):
                    dev = ls["agent_global"].device
            halo_t, owner_t = order_halo(hl, self.bounds, device=dev)
            halo_global = halo_t.cpu().numpy()
            halo_from = np.bincount(owner_t.cpu().numpy(), minlength=self.world_size).astype(np.int64)
            halo_sorted, sorter_t = torch.sort(halo_t, stable=True)
            sa = slice_agents
            if sa is None:
                _, sa = TL.choose_slices(n_local + len(halo_global) if os.environ.get("GJ_RANK_SLICES", "owned") == "ext"
                                         else n_local)
            n_local_pad = -(-n_local // sa) * sa if len(halo_global) else n_local
            n_ext = n_local_pad + len(halo_global)
            n_slices = max(1, -(-n_ext // sa))
            edge_sets = {}
            for name, ls in self.local_sets[r].items():
                g = ls.pop("agent_global")
                if not isinstance(g, torch.Tensor):
                    g = torch.as_tensor(g, device=dev)
                ext = g - a0
                rem = (g < a0) | (g >= a1)
                if len(halo_global):
                    ext = torch.where(rem, n_local_pad + sorter_t[torch.searchsorted(halo_sorted, g).clamp_(max=halo_t.numel() - 1)],
                                      ext)
                elif bool(rem.any()):
                    raise RuntimeError("a remote attendee without a halo list")
                edge_sets[name] = {"agent": ext, "venue": ls["venue"], "people": ls["people"]}

            def ext_attr(v):
                if isinstance(v, torch.Tensor):
                    o = torch.full((n_ext,), int(v[0]), dtype=v.dtype, device=v.device)   # (padding rows: agent 0's, as finish())
                    o[:n_local] = v[a0:a1]
                    if len(halo_global):
                        o[n_local_pad:] = v[halo_t.to(v.device)]
                    return o.cpu().numpy()
                v = np.asarray(v)
                o = np.full(n_ext, v[0], dtype=v.dtype)
                o[:n_local] = v[a0:a1]
                o[n_local_pad:] = v[halo_global]
                return o

            out[r] = RankWorld(r, self.world_size, self.bounds, n_local, sa, n_local_pad, n_ext, n_slices, halo_global,
                               halo_from, ext_attr(age), ext_attr(sex), edge_sets, dict(self.modes),
                               dict(self.venue_global[r]))
        self.local_sets = self.halo_lists = self.venue_global = None      # consumed
        return out

    def finish(self, age, sex, slice_agents: Optional[int] = None) -> Dict[int, "RankWorld"]:
        if any(isinstance(x[0], torch.Tensor) for hl in self.halo_lists.values() for x in hl) or any(
                isinstance(ls["agent_global"], torch.Tensor) for sets in self.local_sets.values() for ls in sets.values()):
            return self._finish_torch(age, sex, slice_agents)
        age, sex = np.asarray(age), np.asarray(sex)
        out = {}
        for r in self.ranks:
            a0, a1 = int(self.bounds[r]), int(self.bounds[r + 1])
            n_local = a1 - a0
            halo_global, owner = order_halo(self.halo_lists[r], self.bounds)
            halo_from = np.bincount(owner, minlength=self.world_size).astype(np.int64)   # grouped by owner
            sorter = np.argsort(halo_global, kind="stable")
            sa = slice_agents
            if sa is None:
                # slices sized for the OWNED agents: phase D runs one workgroup per owned slice, and a rank whose halo
                # is as large as its share (C3 at 8 ranks: 1.25 M owned, 1.4 M halo) would otherwise keep half of the
                # CUs idle there (measured, tools/rank_share.py: 0.126 -> 0.121 ms per step with the venue blocks sized
                # by the owned slices too, DistributedHotPath; GJ_RANK_SLICES=ext is the old rule)
                _, sa = TL.choose_slices(n_local + len(halo_global) if os.environ.get("GJ_RANK_SLICES", "owned") == "ext"
                                         else n_local)
            n_local_pad = -(-n_local // sa) * sa if len(halo_global) else n_local
            n_ext = n_local_pad + len(halo_global)
            n_slices = max(1, -(-n_ext // sa))
            edge_sets = {}
            for name, ls in self.local_sets[r].items():
                g = ls.pop("agent_global")
                ext = g - a0
                rem = (g < a0) | (g >= a1)
                if rem.any():
                    ext[rem] = n_local_pad + sorter[np.searchsorted(halo_global, g[rem], sorter=sorter)]
                edge_sets[name] = {"agent": ext, "venue": ls["venue"], "people": ls["people"]}
            ext_global = np.zeros(n_ext, dtype=np.int64)
            ext_global[:n_local] = np.arange(a0, a1)
            ext_global[n_local_pad:] = halo_global
            out[r] = RankWorld(r, self.world_size, self.bounds, n_local, sa, n_local_pad, n_ext, n_slices, halo_global,
                               halo_from, age[ext_global], sex[ext_global], edge_sets, dict(self.modes),
                               dict(self.venue_global[r]))
        self.local_sets = self.halo_lists = self.venue_global = None      # consumed
        return out




def build_rank_worlds(world: dict, world_size: int, ranks: Optional[Sequence[int]] = None,
                      modes: Optional[Dict[str, str]] = None, slice_agents: Optional[int] = None,
                      progress=None) -> Dict[int, RankWorld]:
    """The parts of ``ranks`` (default: all) of a world held in memory."""
    part = RankPartitioner(world["n_agents"], world_size, ranks, modes, networks=world.get("networks"),
                           n_sets=len(world["edge_sets"]), log=progress)
    for name, es in world["edge_sets"].items():
        mode = part.add_set(name, es["agent"], es["venue"], es["people"])
        if progress:
            progress(f"partitioned edge set {name} ({mode})")
    return part.finish(world["age"], world["sex"], slice_agents)


def build_rank_world(world: dict, rank: int, world_size: int, modes: Optional[Dict[str, str]] = None,
                     slice_agents: Optional[int] = None, progress=None) -> RankWorld:
    return build_rank_worlds(world, world_size, [rank], modes, slice_agents, progress)[rank]


def stream_rank_share(pieces, rank: int, world_size: int, reorder: Optional[str] = None,
                      modes: Optional[Dict[str, str]] = None, progress=None, slice_agents: Optional[int] = None):
    """One rank's share of a world that arrives piece by piece (``synthetic.iter_world``): every edge set is
    relabelled (``reorder``: the locality order of ``synthetic.reorder_agents``, defined by the FIRST set streamed),
    cut down to this rank's part and dropped before the next one is generated.  Returns (RankWorld, share) where
    share = {"networks", "state" (owned agents only), "n_agents", "total_edges", "sizes", "original_id" (owned)};
    identical to ``build_rank_world(reorder_agents(make_world(...)), rank, world_size)`` - tested."""
    from .synthetic import locality_permutation

    part = header = state = None
    order = new_of = None
    for piece in pieces:
        if piece[0] == "header":
            header = piece[1]
            part = RankPartitioner(header["n_agents"], world_size, [rank], modes, networks=header.get("networks"),
                                   n_sets=header.get("n_sets"), log=progress)
        elif piece[0] == "set":
            _, name, es = piece
            agent = es["agent"]
            if reorder is not None:
                if order is None:
                    if name != reorder:
                        raise ValueError(f"the locality order is defined by set {reorder!r}, which must be streamed first")
                    order, new_of = locality_permutation(header["n_agents"], es["agent"], es["venue"])
                agent = new_of[agent]
            mode = part.add_set(name, agent, es["venue"], es["people"])
            if progress:
                progress(f"rank {rank}: kept its part of edge set {name} ({mode})")
            del es, agent
        else:
            state = piece[1]
    a0, a1 = int(part.bounds[rank]), int(part.bounds[rank + 1])
    pick = (lambda v: v[a0:a1]) if order is None else (lambda v: v[order[a0:a1]])
    age = header["age"] if order is None else header["age"][order]
    sex = header["sex"] if order is None else header["sex"][order]
    total_edges, sizes, classes = part.total_edges, dict(part.sizes), dict(part.classes)
    rw = part.finish(age, sex, slice_agents)[rank]
    host = lambda v: v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    share = {"networks": header["networks"], "state": {k: np.ascontiguousarray(host(pick(v))) for k, v in state.items()},
             "n_agents": header["n_agents"], "total_edges": total_edges, "sizes": sizes, "classes": classes,
             "original_id": (np.arange(a0, a1) if order is None else host(order[a0:a1]))}
    return rw, share


def with_twins(names, set_of, twin_of):
    """Network names in accumulation order with the twins of split sets put in: the networks of one edge set stay
    adjacent (the launch groups them by set), so a run of networks on one set is followed by the run of their twins."""
    out, run, run_set = [], [], None

    def flush():
        out.extend(run)
        out.extend(t for t in (twin_of(n) for n in run) if t is not None)

    for n in names:
        s = set_of(n)
        if run and s != run_set:
            flush()
            run = []
        run.append(n)
        run_set = s
    flush()
    return out


def expand_split_networks(specs, networks, betas, edge_sets):
    """Networks of a world whose partition split some edge sets (``mode_of`` "split"): behind every network on a
    split set comes its twin on the set's partial-sum half - same beta, mask and table, so that the two together
    are the original network (sums over a set's venues are sums over both halves).  Returns (specs, network names
    in accumulation order, betas)."""
    from .plan import NetworkSpec

    twins = {s[: -len(SPLIT_SUFFIX)] for s in edge_sets if s.endswith(SPLIT_SUFFIX)}
    if not twins:
        return list(specs), list(networks), dict(betas)
    out_specs, by_name = [], {}
    for sp in specs:
        out_specs.append(sp)
        if sp.edge_set in twins:
            tw = NetworkSpec(sp.name + SPLIT_SUFFIX, sp.edge_set + SPLIT_SUFFIX, sp.mask_kind, sp.table)
            out_specs.append(tw)
            by_name[sp.name] = tw.name
    out_betas = dict(betas)
    for n, t in by_name.items():
        if n in betas:
            out_betas[t] = betas[n]
    base_set = {sp.name: sp.edge_set for sp in specs}
    out_names = with_twins(networks, base_set.get, lambda n: by_name.get(n))
    return out_specs, out_names, out_betas


class HaloExchange:
    """The once-per-step all-to-all of halo transmissions.

    Setup (collective, once): every rank tells each peer which of the peer's agents it needs.
    Per step: ``exchange(x)`` packs ``x[send_index]`` and receives straight into ``x[n_local_pad:]``.
    Works on any backend / device torch.distributed supports (RCCL on GPUs, gloo on CPU in tests).
    """

    def __init__(self, rw: RankWorld, device, group=None, pack=None, force_active: bool = False):
        import torch.distributed as dist

        self.rw, self.group, self.device = rw, group, torch.device(device)
        self.pack = pack
        R, r = rw.world_size, rw.rank
        # RCCL moves device buffers directly; gloo (tests: several ranks sharing one GPU) is staged through the host
        self.host_staged = self.device.type == "cuda" and dist.get_backend(group) == "gloo"
        data_device = self.device
        if self.host_staged:
            self.device = torch.device("cpu")
        self.recv_counts = [int(c) for c in rw.halo_from]
        need_counts = torch.tensor(self.recv_counts, dtype=torch.int64, device=self.device)
        send_counts = torch.empty(R, dtype=torch.int64, device=self.device)
        dist.all_to_all_single(send_counts, need_counts, group=group)
        self.send_counts = [int(c) for c in send_counts.cpu()]
        need_ids = torch.from_numpy(rw.halo_global).to(self.device)       # grouped by owner, ascending
        asked = torch.empty(sum(self.send_counts), dtype=torch.int64, device=self.device)
        dist.all_to_all_single(asked, need_ids, self.send_counts, self.recv_counts, group=group)
        local = asked - int(rw.bounds[r])
        if local.numel() and (int(local.min()) < 0 or int(local.max()) >= rw.n_local):
            raise RuntimeError("halo setup: a peer asked for an agent this rank does not own")
        self.send_index = local.to(device=data_device, dtype=torch.int32).contiguous()
        # one staging buffer per array exchanged in a step (0: transmission, 1: q * transmission): an asynchronous
        # all-to-all may still be reading the first while the second is packed
        self.send_bufs = [torch.empty(self.send_index.numel(), dtype=torch.float32, device=data_device) for _ in range(2)]
        # a collective is entered by every rank or by none: skip the per-step exchange only if NO rank has halo
        total = torch.tensor([self.send_index.numel() + rw.n_halo], dtype=torch.int64, device=self.device)
        dist.all_reduce(total, group=group)
        # force_active (single-rank diagnostics / tests): issue the (empty) all-to-all anyway, so that the asynchronous
        # collective + wait path of the production step runs
        self.active = int(total.item()) > 0 or force_active

    @property
    def bytes_per_step(self) -> int:
        return 4 * (self.send_index.numel() + sum(self.recv_counts))

    def exchange(self, x: torch.Tensor, async_op: bool = False, which: int = 0):
        """x: float32[n_ext]; fills x[n_local_pad:] with the owners' current values.  With async_op the
        collective runs on the backend's own stream and a work handle is returned (wait() before use)."""
        import torch.distributed as dist

        if not self.active:
            return None
        send_buf = self.send_bufs[which]
        if self.pack is not None:
            self.pack(self.send_index, x, send_buf)
        else:
            torch.index_select(x, 0, self.send_index.long(), out=send_buf)
        recv = x[self.rw.n_local_pad:self.rw.n_local_pad + self.rw.n_halo]
        if self.host_staged:
            tmp = torch.empty(self.rw.n_halo, dtype=torch.float32)
            dist.all_to_all_single(tmp, send_buf.cpu(), self.recv_counts, self.send_counts, group=self.group)
            recv.copy_(tmp)
            return None
        return dist.all_to_all_single(recv, send_buf, self.recv_counts, self.send_counts, group=self.group,
                                      async_op=async_op)


def emulate_exchange(rank_worlds: Sequence[RankWorld], xs: Sequence[np.ndarray]) -> None:
    """In-process stand-in for HaloExchange over all ranks at once (tests): fill every rank's halo
    slots from the owners' local values."""
    bounds = rank_worlds[0].bounds
    glob = np.concatenate([x[: rw.n_local] for rw, x in zip(rank_worlds, xs)])
    assert len(glob) == bounds[-1]
    for rw, x in zip(rank_worlds, xs):
        x[rw.n_local_pad:rw.n_local_pad + rw.n_halo] = glob[rw.halo_global]


class DistributedHotPath:
    """bench.py's stepping object for N > 1: compile this rank's part, step with the two collectives."""

    def __init__(self, world: dict, specs, betas: Dict[str, float], device, rank: int, world_size: int,
                 seed: int = 0, group=None, modes: Optional[Dict[str, str]] = None, collectives: bool = True,
                 progress=None, min_group_floats: int = 1 << 16, production_at_one_rank: bool = False,
                 quarantine_threshold: Optional[float] = None, rank_world: Optional[RankWorld] = None,
                 total_edges: Optional[int] = None, device_compile: bool = False, plan_kw: Optional[dict] = None):
        """``world``: the whole world (this rank's part is cut out of it), or - with ``rank_world`` - only
        {"networks": [...], "state": {name: this rank's OWNED agents' arrays}} next to the prebuilt part and the
        world's ``total_edges`` (what ``RankPartitioner`` hands over when the world is streamed)."""
        from . import _native as N
        from .benchrun import EventLog
        from .engine import AgentBuffers, InfectionEngine
        from .plan import DevicePlan, compile_plan

        self.device = torch.device(device)
        self.rank, self.world_size, self.group = rank, world_size, group
        rw = self.rw = rank_world if rank_world is not None else build_rank_world(world, rank, world_size, modes,
                                                                                  progress=progress)
        specs, networks, betas = expand_split_networks(specs, world["networks"], betas, rw.edge_sets)
        plan_kw = dict(plan_kw or {})
        if os.environ.get("GJ_RANK_SLICES", "owned") != "ext":     # venue blocks for the owned slices' tiles (see finish())
            plan_kw.setdefault("eb_target", TL.choose_block_edges(max(1, -(-rw.n_local // rw.slice_agents))))
        # A rank's share has few venue blocks (C3 / 8: ~200 for 256 CUs), and a block of a set that carries several
        # networks sums every slot once per network: such sets get blocks of a sixth of the edges while the share has
        # fewer than two blocks per CU (measured, tools/rank_share.py: C3 / 8 clustered 0.129 -> 0.111 ms, random 0.122 ->
        # 0.118, the june preset / 8 0.160 -> 0.121; on a whole GPU smaller blocks LOSE: profiles/r04_ab_multi_net_block_div_*)
        local_edges = sum(int(es["agent"].numel() if hasattr(es["agent"], "numel") else len(es["agent"]))
                          for es in rw.edge_sets.values())
        eb_rank = plan_kw.get("eb_target") or TL.choose_block_edges(max(1, -(-rw.n_ext // rw.slice_agents)))
        if "GJ_MULTI_NET_BLOCK_DIV" not in os.environ and local_edges < 512 * eb_rank:
            plan_kw.setdefault("multi_net_block_div", 6)
        leisure = tuple(s for s in rw.edge_sets if s.split(SPLIT_SUFFIX)[0] == "leisure")
        host = compile_plan(rw.n_local, rw.edge_sets, age=rw.age, sex=rw.sex, n_ext_agents=rw.n_ext,
                            layout="tiled", slices=(rw.n_slices, rw.slice_agents), progress=progress,
                            device=self.device if device_compile else None, leisure_sets=leisure or ("leisure",),
                            **plan_kw)
        self.set_of = {sp.name: sp.edge_set for sp in specs}
        # The production step scatters the partial-sum sets (phase A) while the halo all-to-all may still be writing the
        # halo part of the transmission array: safe because those sets keep only the rank's OWN agents' edges, i.e. no
        # chunk of theirs lies in a halo slice.  Checked here, once, on the compiled plan.
        first_halo_slice = rw.n_local_pad // rw.slice_agents          # (with a halo, n_local_pad is a slice boundary)
        for hs in host.sets:
            if rw.n_halo and rw.modes[hs.name] == "partial" and hs.tiled is not None and hs.tiled.n_blocks:
                cp = hs.tiled.chunk_ptr
                tail = cp[first_halo_slice:] if first_halo_slice < len(cp) else cp[-1:]
                if int(tail[-1]) != int(tail[0]):
                    raise RuntimeError(f"edge set {hs.name}: a partial-sum set holds edges of halo agents")
        nets_on = {}
        for sp in specs:
            nets_on[sp.edge_set] = nets_on.get(sp.edge_set, 0) + 1
        floats = {n: max(1, len(es["people"])) * max(1, nets_on.get(n, 1))
                  for n, es in rw.edge_sets.items() if rw.modes[n] == "partial"}
        # A second all-reduce costs ~45 us of host time per step (tools/host_overhead.py): worth it only while
        # the rank's kernels take several times that, i.e. for shares of >= ~2e7 set-edges (global count / ranks,
        # so every rank takes the same decision)
        if total_edges is None:
            total_edges = sum(len(es["agent"]) for es in world["edge_sets"].values())
        pipelined = total_edges / max(1, world_size) >= PIPELINE_MIN_EDGES or min_group_floats <= 1
        self.reduce_groups = reduce_groups(floats, min_group_floats if pipelined else 1 << 62)
        partial = [n for g in self.reduce_groups for n in g]
        self.engine = InfectionEngine(DevicePlan(host, specs, self.device, flat_cum_sets=partial))
        self.flat_cum = self.engine.plan.flat_cum
        self.group_cum = []                                            # one contiguous view per all-reduce
        for g in self.reduce_groups:
            lo = self.engine.plan.flat_offsets[g[0]][0]
            hi = sum(self.engine.plan.flat_offsets[g[-1]])
            self.group_cum.append(self.flat_cum[lo:hi])
        self.exchange_sets = [n for n in rw.edge_sets if rw.modes[n] != "partial"]   # halo or local
        self._params_cache = {}
        self.networks = networks
        self.betas, self.seed = betas, seed
        a0, a1 = int(rw.bounds[rank]), int(rw.bounds[rank + 1])
        own = (lambda v: v) if rank_world is not None else (lambda v: v[a0:a1])
        st = {k: torch.from_numpy(np.ascontiguousarray(own(v))).to(self.device) for k, v in world["state"].items()}
        if any(v.numel() != rw.n_local for v in st.values()):
            raise ValueError("state arrays must cover exactly this rank's owned agents")
        st["transmission"] = torch.zeros(rw.n_ext, dtype=torch.float32, device=self.device)
        self.q_thr = quarantine_threshold
        # quarantine (quarantine_policies.py:13-33): q * transmission is a second per-agent array of pass 1; its halo
        # part travels like the transmissions' when a MASKED edge set runs in halo mode (households read raw values)
        st["q_transmission"] = (torch.zeros(rw.n_ext, dtype=torch.float32, device=self.device)
                                if quarantine_threshold is not None else None)
        from . import _native as N_

        self.exchange_q = quarantine_threshold is not None and any(
            rw.modes[sp.edge_set] == "halo" and sp.mask_kind != N_.MASK_RAW for sp in specs if sp.edge_set in rw.modes)
        self.state = st
        self.new_infected = torch.empty(rw.n_local, dtype=torch.float32, device=self.device)
        self.bufs = AgentBuffers(self.engine.plan, max_infectiousness=st["max_infectiousness"], shape=st["shape"],
                                 rate=st["rate"], shift=st["shift"], infection_time=st["infection_time"],
                                 is_infected=st["is_infected"], susceptibility=st["susceptibility"],
                                 transmission=st["transmission"], current_stage=st["current_stage"],
                                 q_transmission=st["q_transmission"])
        self.io = self.engine.io(new_infected=self.new_infected)
        lib = N.load()

        def pack(index, src, out):
            N.check(lib.gj_pack_f32(index.numel(), N.ptr(index), N.ptr(src), N.ptr(out), N.current_stream()),
                    "gj_pack_f32")

        # production_at_one_rank: diagnostics - take the overlapped multi-rank step with a single rank
        self.halo = (HaloExchange(rw, self.device, group=group, pack=pack,
                                  force_active=production_at_one_rank and world_size == 1)
                     if (collectives and (world_size > 1 or production_at_one_rank)) else None)
        self.a0 = a0
        self.t = 0
        self.log = EventLog()
        self.graph = None
        self._clock_ptr = None
        self._venue_weights: Dict[str, torch.Tensor] = {}

    def params(self, only=None):
        """only: None = every network; "halo" / "partial" = the networks on sets of that exchange mode;
        a collection of edge-set names = the networks on those sets.  The struct is built once per
        selection and only its clock fields change from step to step (this sits on the launch path)."""
        edge_set_of = self.set_of.__getitem__
        key = only if (only is None or isinstance(only, str)) else tuple(only)
        p = self._params_cache.get(key)
        if p is None:
            nets = self.networks
            if isinstance(only, str):
                want = ("halo",) if only == "halo" else ("partial", "local")
                nets = [n for n in nets if self.rw.modes[edge_set_of(n)] in want]
            elif only is not None:
                nets = [n for n in nets if edge_set_of(n) in key]
            has_q = self.q_thr is not None
            p = self.engine.params(now=1.0, delta_time=1.0, day_type=0, active=nets, betas=self.betas,
                                   has_quarantine=has_q, q_threshold=self.q_thr if has_q else float("inf"),
                                   seed=self.seed, step=0, agent_offset=self.a0)
            p.clock = self._clock_ptr            # None: scalars from the launch arguments
            self._params_cache[key] = p
        p.now, p.step = 1.0 + self.t, self.t
        return p

    def _all_reduce(self, async_op: bool, buf: Optional[torch.Tensor] = None):
        import torch.distributed as dist

        buf = self.flat_cum if buf is None else buf
        if buf is None or not buf.numel() or not dist.is_initialized():
            return None
        if dist.get_backend(self.group) == "gloo":        # tests: host-staged, synchronous
            tmp = buf.cpu()
            dist.all_reduce(tmp, group=self.group)
            buf.copy_(tmp)
            return None
        return dist.all_reduce(buf, group=self.group, async_op=async_op)

    def step(self, timed: bool = False):
        """One step of the benchmark's fixed schedule (all networks, constant betas)."""
        if self.graph is not None and not timed:
            self.graph.replay()              # clock advance + every launch and collective of the production step
            self.t += 1
            return
        if self.graph is not None:           # an eager (event-bracketed) step of a captured runner: keep the device clock in step
            self.clock.advance(self._delta_now)
        self.run_step(self.bufs, self.io, self.params, timed=timed)
        self.t += 1

    def capture(self, delta_now: float = 1.0):
        """The production step - kernels AND the RCCL collectives, on the streams they run on - captured once in a
        hipGraph and replayed per timestep: one host call instead of ~8 launches + 2-3 collectives (at 8 ranks a
        rank's kernels take ~0.12 ms, the eager host path 0.09-0.14 ms).  ``now`` and the Philox stream id are read
        from device memory (engine.StepClock), which the graph's first node advances."""
        from .engine import StepClock

        self.clock = StepClock(self.device)
        self._delta_now = float(delta_now)
        self._params_cache.clear()
        self._clock_ptr = self.clock.ptr
        p_all = self.params(None)
        self.clock.set(p_all.now - delta_now, self.t - 1)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            self.clock.advance(delta_now)
            self.run_step(self.bufs, self.io, self.params, timed=False)
        self.graph = graph
        return graph

    def drop_graph(self) -> None:
        """Back to the eager production step (launch scalars from the arguments again)."""
        self.graph = None
        self._clock_ptr = None
        self._params_cache.clear()

    def suspend_graph(self):
        """Eager production steps for a while, the captured graph kept: returns the token ``resume_graph`` takes."""
        token = self.graph
        self.drop_graph()
        return token

    def resume_graph(self, token) -> None:
        """Replay the graph ``suspend_graph`` put aside again, from the runner's current timestep."""
        if token is None:
            return
        self.graph = token
        self._clock_ptr = self.clock.ptr
        self._params_cache.clear()
        self.sync_clock()

    def sync_clock(self) -> None:
        """After ``self.t`` was set by hand (a state restored): the device clock the graph's first node advances."""
        if self.graph is not None:
            self.clock.set((1.0 + self.t) - self._delta_now, self.t - 1)

    def run_step(self, bufs, io, params_of, timed: bool = False):
        """The multi-rank launch sequence for one step.  ``params_of(None)`` gives the launch parameters of every
        active network, ``params_of(edge set names)`` those of the networks on these sets (the phase entry points
        take any subset); ``bufs`` must carry this object's extended ``transmission`` / ``q_transmission`` arrays.

        Untimed (production) form overlaps the collectives with the phases that do not depend on them: the halo
        all-to-all runs under phases A+B of the partial-sum sets, the partial-sum all-reduces (one per group of
        ``reduce_groups``) under the later groups' A+B and the halo sets' A+B+C.  The timed form runs everything
        in sequence, with a single all-reduce, so that every launch and collective can be bracketed by events."""
        e = self.engine
        p_all = params_of(None)
        if timed or self.halo is None:
            mark = self.log.mark if timed else (lambda label: None)
            mark("begin")
            e.step_phase(bufs, p_all, io, 0)
            mark("transmission")
            if self.halo is not None:
                self.halo.exchange(self.state["transmission"])
                if self.exchange_q and p_all.has_quarantine:
                    self.halo.exchange(self.state["q_transmission"], which=1)
                mark("halo_all_to_all")
            e.step_phase(bufs, p_all, io, 1)
            mark("tile_scatter")
            e.step_phase(bufs, p_all, io, 5)
            mark("tile_venues_B")
            self._all_reduce(False)
            mark("partial_all_reduce")
            e.step_phase(bufs, p_all, io, 6)
            mark("tile_venues_C")
            e.step_phase(bufs, p_all, io, 3)
            mark("tile_agents")
            return
        # Production form.  Sets that need no partial sums ("halo" / "local") run A, B, C under the
        # all-reduces; the partial-sum sets go group by group (largest cum buffer first) so that the
        # big all-reduce is in flight while the remaining groups are still computing.
        e.step_phase(bufs, p_all, io, 0)                                  # transmission
        h = self.halo.exchange(self.state["transmission"], async_op=True)  # all-to-all on the comm stream
        hq = (self.halo.exchange(self.state["q_transmission"], async_op=True, which=1)
              if (self.exchange_q and p_all.has_quarantine) else None)
        pending = []
        for g, buf in zip(self.reduce_groups, self.group_cum):
            p_g = params_of(g)
            if not p_g.n_nets:      # no active network on these sets this step (the same decision on every rank)
                continue
            e.step_phase(bufs, p_g, io, 7)                                # A, B of this group's sets
            pending.append((p_g, self._all_reduce(True, buf)))            # all-reduce on the comm stream
        for work in (h, hq):
            if work is not None:
                work.wait()
        p_x = params_of(self.exchange_sets)
        if p_x.n_nets:
            e.step_phase(bufs, p_x, io, 8)                                # A, B, C of the halo / local sets
        for p_g, r in pending:
            if r is not None:
                r.wait()
            e.step_phase(bufs, p_g, io, 6)                                # C of the group
        e.step_phase(bufs, p_all, io, 3)                                  # D + epilogue: all sets

    # ---- what the differentiable step (autograd.DistributedHotPathStep) needs -------------------------------
    def sparse_passes(self, bufs, io, p, between=None) -> None:
        """The two sparse passes of one step on ``bufs``' extended transmission arrays, in sequence and WITHOUT the
        decision: halo all-to-all, phases A + B, all-reduce of the partial sums, ``between()`` (every set's complete
        per-venue sums are in ``plan.cum_of`` then), phases C + D (the per-agent sums go to ``io.trans_susc``).
        With ``p.transpose = 1`` this is the adjoint of the same operator - its communication pattern is the
        forward's, the cotangents of the halo agents travel like their transmissions."""
        e = self.engine
        if self.halo is not None:
            self.halo.exchange(bufs.tensors["transmission"])
            if self.exchange_q and p.has_quarantine:
                self.halo.exchange(bufs.tensors["q_transmission"], which=1)
        e.step_phase(bufs, p, io, 1)
        e.step_phase(bufs, p, io, 5)
        if self.halo is not None:
            self._all_reduce(False)
        if between is not None:
            between()
        e.step_phase(bufs, p, io, 6)
        e.step_phase(bufs, p, io, 4)

    def venue_weights(self, set_name: str) -> Optional[torch.Tensor]:
        """Weight of this rank's copy of every venue of a set in a sum over the WORLD's venues (the gradient of a
        network's beta is such a sum): a partition of unity over the ranks.  Partial-sum sets hold every venue
        complete on every rank after the all-reduce - rank 0 counts them; a halo set's venue lives on every rank
        that owns one of its attendees - the rank owning the attendee with the smallest id counts it.
        None = weight 1 everywhere (a single rank)."""
        if self.world_size == 1 or self.halo is None:
            return None
        w = self._venue_weights.get(set_name)
        if w is None:
            rw = self.rw
            V = len(rw.edge_sets[set_name]["people"])
            if rw.modes[set_name] == "partial":
                w = torch.full((V,), 1.0 if self.rank == 0 else 0.0, dtype=torch.float64, device=self.device)
            else:
                es = rw.edge_sets[set_name]
                host = lambda t: t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
                es = {"agent": host(es["agent"]), "venue": host(es["venue"])}
                ext = np.asarray(es["agent"], dtype=np.int64)
                gid = np.where(ext < rw.n_local, ext + self.a0,
                               rw.halo_global[np.clip(ext - rw.n_local_pad, 0, max(0, rw.n_halo - 1))] if rw.n_halo
                               else ext + self.a0)
                first = np.full(V, np.iinfo(np.int64).max, dtype=np.int64)
                np.minimum.at(first, np.asarray(es["venue"], dtype=np.int64), gid)
                mine = (first >= self.a0) & (first < self.a0 + rw.n_local)
                w = torch.from_numpy(mine.astype(np.float64)).to(self.device)
            self._venue_weights[set_name] = w
        return w

    def all_reduce_max(self, x: torch.Tensor) -> torch.Tensor:
        return self._reduce_small(x, "max")

    def all_reduce_sum(self, x: torch.Tensor) -> torch.Tensor:
        return self._reduce_small(x, "sum")

    def _reduce_small(self, x: torch.Tensor, op: str) -> torch.Tensor:
        import torch.distributed as dist

        if self.halo is None or self.world_size == 1 or not dist.is_initialized():
            return x
        rop = dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM
        if dist.get_backend(self.group) == "gloo":
            tmp = x.cpu()
            dist.all_reduce(tmp, op=rop, group=self.group)
            return tmp.to(x.device)
        x = x.clone()
        dist.all_reduce(x, op=rop, group=self.group)
        return x

    def reset_timers(self):
        self.log.clear()

    def kernel_ms(self) -> Dict[str, float]:
        return {k: float(np.mean(v)) for k, v in self.log.spans().items()}


class PartitionedHotPath:
    """Several agent partitions stepped one after the other on ONE GPU - for worlds whose tiles would
    become too small as a single partition (edges per tile fall like 1/size: 10^8 agents on one device
    leave ~25 edges per tile).  Each partition is exactly a rank of the multi-GPU path with its own
    compiled plan; the two collectives become device-side copies / sums between the partitions' buffers.
    Results are identical to the unpartitioned run (fixed-point sums, Philox keyed by global agent id)."""

    def __init__(self, world: dict, specs, betas: Dict[str, float], device, parts: int, seed: int = 0,
                 progress=None, device_compile: bool = False):
        from .benchrun import EventLog

        self.device = torch.device(device)
        self.parts = parts
        self.ranks: List[DistributedHotPath] = []
        rws = build_rank_worlds(world, parts, progress=progress)          # every set sorted once, not scanned per part
        total_edges = sum(len(es["agent"]) for es in world["edge_sets"].values())
        for r in range(parts):
            rw = rws.pop(r)
            a0, a1 = int(rw.bounds[r]), int(rw.bounds[r + 1])
            share = {"networks": world["networks"], "state": {k: v[a0:a1] for k, v in world["state"].items()}}
            self.ranks.append(DistributedHotPath(share, specs, betas, device, r, parts, seed=seed, collectives=False,
                                                 progress=None, rank_world=rw, total_edges=total_edges,
                                                 device_compile=device_compile))
            if progress:
                progress(f"compiled partition {r + 1}/{parts}")
        self.halo_index = [torch.from_numpy(rk.rw.halo_global).to(self.device) for rk in self.ranks]
        self.log = EventLog()
        self.new_infected = None
        self.probs = None

    @property
    def state(self):
        return {k: torch.cat([rk.state[k][: rk.rw.n_local] for rk in self.ranks]) for k in
                ("is_infected", "susceptibility", "infection_time")}

    def step(self, timed: bool = False):
        mark = self.log.mark if timed else (lambda label: None)
        ranks = self.ranks
        mark("begin")
        ps = [rk.params() for rk in ranks]
        for rk, p in zip(ranks, ps):
            rk.engine.step_phase(rk.bufs, p, rk.io, 0)
        mark("transmission")
        if any(len(h) for h in self.halo_index):
            glob = torch.cat([rk.state["transmission"][: rk.rw.n_local] for rk in ranks])
            for rk, idx in zip(ranks, self.halo_index):
                if len(idx):
                    rk.state["transmission"][rk.rw.n_local_pad:rk.rw.n_local_pad + rk.rw.n_halo] = glob[idx]
        mark("halo_copy")
        for rk, p in zip(ranks, ps):
            rk.engine.step_phase(rk.bufs, p, rk.io, 1)
        mark("tile_scatter")
        for rk, p in zip(ranks, ps):
            rk.engine.step_phase(rk.bufs, p, rk.io, 5)
        mark("tile_venues_B")
        if ranks[0].flat_cum is not None and ranks[0].flat_cum.numel():
            # the stand-in of the partial-sum all-reduce: fp32 adds in `reduce_order` (default: rank order).  RCCL sums
            # in an order of its own (ring / tree, per chunk), so a venue's cum - and what follows from it - agrees
            # across rank counts to fp32 rounding of R terms, not bit for bit (tests: the order reversed)
            order = getattr(self, "reduce_order", None) or range(len(ranks))
            order = list(order)
            total = ranks[order[0]].flat_cum.clone()
            for r in order[1:]:
                total += ranks[r].flat_cum
            for rk in ranks:
                rk.flat_cum.copy_(total)
        mark("partial_sum")
        for rk, p in zip(ranks, ps):
            rk.engine.step_phase(rk.bufs, p, rk.io, 6)
        mark("tile_venues_C")
        for rk, p in zip(ranks, ps):
            rk.engine.step_phase(rk.bufs, p, rk.io, 3)
            rk.t += 1
        mark("tile_agents")

    def reset_timers(self):
        self.log.clear()

    def kernel_ms(self) -> Dict[str, float]:
        return {k: float(np.mean(v)) for k, v in self.log.spans().items()}


def world_from_data(data, model=None, networks: Optional[Sequence[str]] = None) -> dict:
    """A world in the reference's graph format (``HeteroData`` as ``Runner.get_data`` leaves it) -> the neutral
    description ``build_rank_world`` / ``DistributedHotPath`` partition: ``n_agents``, ``age``, ``sex``, the COO
    edge sets with their ``people`` counts, the per-agent state, and ``networks`` = the infection networks of
    ``model`` (a ``GradJune``) or the given names, restricted to the edge sets the world has and listed in the
    activity-hierarchy order in which the reference accumulates them."""
    from .synthetic import edge_set_of

    ag = data["agent"]
    n = len(ag["id"])
    np_ = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    world = {"preset": "data", "n_agents": n, "age": np_(ag["age"]).astype(np.int64),
             "sex": (np_(ag["sex"]) if "sex" in ag else np.zeros(n)).astype(np.int64), "edge_sets": {}}
    for key, store in data.edge_items():
        src, rel, dst = key
        if src != "agent" or not rel.startswith("attends_") or "edge_index" not in store:
            continue
        ei = np_(store.edge_index).astype(np.int64)
        people = np_(data[dst]["people"]).astype(np.int64)
        world["edge_sets"][rel[len("attends_"):]] = {"agent": ei[0], "venue": ei[1], "people": people}
    if networks is None:
        networks = list(model.infection_networks.networks.keys()) if model is not None else list(world["edge_sets"])
    from .timer import activity_hierarchy

    # in the reference's accumulation order (timer.py:14-26), which also keeps the networks of one set adjacent
    world["networks"] = sorted((name for name in networks if edge_set_of(name) in world["edge_sets"]),
                               key=activity_hierarchy.index)
    f32 = lambda t: np_(t).astype(np.float32)
    ip = ag["infection_parameters"]
    world["state"] = {k: f32(ip[k]) for k in ("max_infectiousness", "shape", "rate", "shift")}
    for k in ("is_infected", "susceptibility", "infection_time"):
        world["state"][k] = f32(ag[k])
    world["state"]["current_stage"] = f32(ag["symptoms"]["current_stage"])
    return world
