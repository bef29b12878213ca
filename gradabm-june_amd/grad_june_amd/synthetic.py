"""Seeded synthetic contact worlds in the reference's graph format.

The reference ships one 769-agent world; its "London" world is not in the tree.  Benchmarks and
large-size tests therefore use worlds generated here with ``numpy.random.default_rng(seed)``:
unsorted COO edge lists per venue type, ``people[v]`` = degree, agent age/sex, the four
per-agent infection parameters drawn from the default distributions
(configs/default.yaml:100-116), 1 % of the agents infected at U[-10, 0] days.

Presets (BASELINE.json configs):
  "c2"  1 M agents, household/school/company, 5 memberships per agent and network (15 M edges)
  "c3"  10 M agents, household, care_home, company, school, university + pub/grocery/gym on one
        shared leisure set; every edge set 15 M edges (90 M set-edges, 120 M network-edges)
  "c5"  power-law venue degrees (Zipf alpha=2 truncated to [1, 50 000])
  "june"  a world with the MEMBERSHIP STRUCTURE of the reference's own graphs (june_world_loader: population/group_ids
        holds at most one household, one primary activity - school, company, university or care home - per person;
        leisure attaches everybody of the k nearest super areas to each super-area venue, leisure_loader.py:38-73) and
        the reference's default eleven networks (configs/default.yaml:12-43), six of them on the one leisure set.
        Always on a map (``geography`` is "clustered" whatever is asked): see ``JUNE_WORLD``.
All presets scale with ``n_agents`` (edges per agent are kept).

``geography="random"`` (default; SURVEY 8d's specification): agents are dealt to venue slots uniformly at random - no
locality at all.  ``geography="clustered"``: the same presets (venue-size distributions, memberships per agent, exact
``people[v]`` = degree) on a world that HAS a geography, as the reference's worlds do (june_world_loader: people live in
areas inside super areas, ``population/super_area``; venues belong to a super area; leisure attaches the people of the
k nearest super areas to each super-area venue, leisure_loader.py:38-73) - see ``GEOGRAPHY``.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

PRESETS = {
    # set -> (memberships per agent, size distribution)
    "c2": {
        "household": (5.0, ("poisson1", 1.5)),
        "school": (5.0, ("lognormal", 500.0, 0.5)),
        "company": (5.0, ("lognormal", 20.0, 1.0)),
    },
    "c3": {
        "household": (1.5, ("poisson1", 1.5)),
        "care_home": (1.5, ("lognormal", 50.0, 0.5)),
        "company": (1.5, ("lognormal", 20.0, 1.0)),
        "school": (1.5, ("lognormal", 500.0, 0.5)),
        "university": (1.5, ("lognormal", 2000.0, 0.5)),
        "leisure": (1.5, ("lognormal", 5000.0, 0.5)),
    },
    "c5": {
        s: (1.5, ("zipf", 2.0, 50_000))
        for s in ("household", "care_home", "company", "school", "university", "leisure")
    },
}

#: the "june" preset: who attends what (by age, as JUNE's activity assignment does; the shares are stated, not fitted),
#: venue sizes, and k of the leisure loader
JUNE_WORLD = {
    "household": ("poisson1", 1.4),                                       # every person lives in exactly one household
    "school": (5, 18, 1.0, ("lognormal", 500.0, 0.5)),                    # ages [5, 18): everybody
    "university": (18, 25, 0.3, ("lognormal", 2000.0, 0.5)),              # ages [18, 25): 30 %
    "company": (18, 65, 0.7, ("lognormal", 20.0, 1.0)),                   # ages [18, 65): 70 % of those not at university
    "care_home": (75, 100, 0.1, ("lognormal", 50.0, 0.5)),                # ages >= 75: 10 %
    "k_leisure": 3,                                                       # GraphLoader(k_leisure): the k nearest super areas
}

DEFAULT_AGENTS = {"c2": 1_000_000, "c3": 10_000_000, "c5": 100_000_000, "june": 10_000_000}

NETWORKS = {
    "june": ["school", "university", "company", "care_home", "pub", "gym", "grocery", "visit", "care_visit", "cinema",
             "household"],
    "c2": ["school", "company", "household"],
    "c3": ["school", "university", "company", "care_home", "pub", "gym", "grocery", "household"],
    "c5": ["school", "university", "company", "care_home", "pub", "gym", "grocery", "household"],
}


# ---- geography of the "clustered" worlds ---------------------------------------------------------------------------
# Agents are numbered along the geography: agent a lives in super area a // SUPER_AREA_AGENTS (JUNE's super areas are
# MSOAs of ~8 000 residents, its areas output areas of ~300), and the super areas are the cells of a square grid
# visited along a Hilbert curve, so that a contiguous range of agent ids - a rank of the multi-GPU partition - is a
# compact region of the map.  A membership of an agent in a set goes
#     with probability 1 - p_near - p_leak  to a venue of its own super area,
#     with probability p_near               to a venue of one of the <= 8 surrounding super areas (the "k nearest super
#                                           areas" of leisure_loader.py:38-73; catchment areas of schools, local commutes),
#     with probability p_leak               to a venue anywhere in the world (long commutes, universities, care visits),
# and the venues of a set are laid out along the same curve (a venue larger than a super area spans neighbouring ones).
# Households are formed from CONSECUTIVE agents (a household's members are neighbours in the id order): p = None.
# The probabilities are stated, not fitted: companies leak most (London's commuting is city-wide), schools least.
SUPER_AREA_AGENTS = 5000
GEOGRAPHY = {
    "household": None,
    "care_home": (0.20, 0.02),
    "company": (0.30, 0.30),
    "school": (0.15, 0.01),
    "university": (0.40, 0.10),
    "leisure": (0.40, 0.02),
}


def hilbert_cells(grid: int):
    """(x, y) of the cells of a ``grid`` x ``grid`` map in the order a Hilbert curve (of the enclosing power-of-two
    square) visits them: consecutive cells are neighbours or close, any contiguous run of cells is a compact region."""
    order = 1
    while order < grid:
        order *= 2
    d = np.arange(order * order, dtype=np.int64)
    x = np.zeros_like(d)
    y = np.zeros_like(d)
    t = d.copy()
    s = 1
    while s < order:
        rx = 1 & (t // 2)
        ry = 1 & (t ^ rx)
        flip = (ry == 0) & (rx == 1)          # rotate the quadrant
        x_f = np.where(flip, s - 1 - x, x)
        y_f = np.where(flip, s - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y_f, x_f), np.where(swap, x_f, y_f)
        x += s * rx
        y += s * ry
        t //= 4
        s *= 2
    keep = (x < grid) & (y < grid)
    return x[keep], y[keep]


def super_area_map(n_agents: int, sa_agents: int = SUPER_AREA_AGENTS):
    """The map of a clustered world: ``n_sa`` super areas of ``sa_agents`` consecutive agents each, their grid cells and
    ``nb[n_sa, 8]`` / ``nb_n[n_sa]`` = the surrounding super areas that exist (at least one for n_sa > 1)."""
    n_sa = max(1, -(-int(n_agents) // sa_agents))
    grid = int(np.ceil(np.sqrt(n_sa)))
    x, y = hilbert_cells(grid)
    x, y = x[:n_sa], y[:n_sa]
    cell = np.full((grid + 2, grid + 2), -1, dtype=np.int64)
    cell[x + 1, y + 1] = np.arange(n_sa)
    nb = np.stack([cell[x + 1 + dx, y + 1 + dy] for dx in (-1, 0, 1) for dy in (-1, 0, 1) if (dx, dy) != (0, 0)], axis=1)
    order = np.argsort(nb < 0, axis=1, kind="stable")            # existing neighbours first
    nb = np.take_along_axis(nb, order, axis=1)
    nb_n = (nb >= 0).sum(1)
    return {"n_sa": n_sa, "sa_agents": sa_agents, "x": x, "y": y, "nb": nb, "nb_n": nb_n}


def _membership_keys(rng, agent: np.ndarray, n_agents: int, geo, mix) -> np.ndarray:
    """Where on the curve (in units of agents) each membership looks for its venue."""
    if mix is None:                                   # households: neighbours in the id order
        return agent + rng.uniform(-1.5, 1.5, len(agent))
    p_near, p_leak = mix
    sa = geo["sa_agents"]
    home = agent // sa
    u = rng.random(len(agent))
    target = home.copy()
    near = (u >= 1.0 - p_near - p_leak) & (u < 1.0 - p_leak) & (geo["nb_n"][home] > 0)
    pick = (rng.random(int(near.sum())) * geo["nb_n"][home[near]]).astype(np.int64)
    target[near] = geo["nb"][home[near], pick]
    leak = u >= 1.0 - p_leak
    target[leak] = rng.integers(0, geo["n_sa"], int(leak.sum()))
    return (target + rng.random(len(agent))) * float(sa)


def _venue_sizes(rng, dist, n_edges: int) -> np.ndarray:
    """Draw venue sizes until they sum to ``n_edges`` (last venue takes the remainder)."""
    kind = dist[0]
    if kind == "poisson1":
        mean = 1.0 + dist[1]
        draw = lambda m: 1 + rng.poisson(dist[1], m)
    elif kind == "lognormal":
        mean_target, sigma = dist[1], dist[2]
        mu = np.log(mean_target) - 0.5 * sigma * sigma
        mean = mean_target
        draw = lambda m: np.maximum(1, rng.lognormal(mu, sigma, m)).astype(np.int64)
    elif kind == "zipf":
        alpha, cap = dist[1], dist[2]
        mean = 12.0
        draw = lambda m: np.minimum(rng.zipf(alpha, m), cap).astype(np.int64)
    else:
        raise ValueError(kind)
    sizes = np.zeros(0, dtype=np.int64)
    total = 0
    while total < n_edges:
        m = int((n_edges - total) / mean * 1.05) + 16
        sizes = np.concatenate([sizes, draw(m).astype(np.int64)])
        total = int(sizes.sum())
    cs = np.cumsum(sizes)
    last = int(np.searchsorted(cs, n_edges, side="left"))
    sizes = sizes[: last + 1].copy()
    sizes[last] -= cs[last] - n_edges
    return sizes[sizes > 0]


def _edge_set(rng, n_agents: int, n_edges: int, dist, geo=None, mix=None, members=None) -> dict:
    """``geo`` (a ``super_area_map``) + ``mix`` (the set's entry of GEOGRAPHY): the clustered form - memberships are
    dealt to the venue slots in the order of where they look for a venue, the slots lie along the curve.
    ``members``: the agent of every membership (the "june" preset: one per person who attends); default: every agent
    floor(E / A) memberships, a random subset one more."""
    if members is not None:
        n_edges = len(members)
    sizes = _venue_sizes(rng, dist, n_edges) if n_edges else np.zeros(0, dtype=np.int64)
    V = len(sizes)
    venue = np.repeat(np.arange(V, dtype=np.int64), sizes)
    if members is not None:
        agent = np.asarray(members, dtype=np.int64).copy()
    else:
        # every agent gets floor(E/A) memberships, a random subset one more
        base, extra = divmod(n_edges, n_agents)
        parts = [np.tile(np.arange(n_agents, dtype=np.int64), base)] if base else []
        if extra:
            parts.append(rng.choice(n_agents, extra, replace=False).astype(np.int64))
        agent = np.concatenate(parts)
    if n_edges == 0:
        return {"agent": agent, "venue": venue, "people": np.zeros(max(V, 1), dtype=np.int64)}
    if geo is None:
        rng.shuffle(agent)
    else:
        agent = agent[np.argsort(_membership_keys(rng, agent, n_agents, geo, mix), kind="stable")]
    # no duplicate (agent, venue) pair: move the few collisions to the next venue
    for _ in range(4):
        key = agent * V + venue
        order = np.argsort(key, kind="stable")
        dup = np.zeros(len(key), dtype=bool)
        dup[order[1:]] = key[order[1:]] == key[order[:-1]]
        if not dup.any():
            break
        venue[dup] = (venue[dup] + 1) % V
    perm = rng.permutation(n_edges)          # the reference format is unsorted COO
    agent, venue = agent[perm], venue[perm]
    people = np.bincount(venue, minlength=V).astype(np.int64)
    return {"agent": agent, "venue": venue, "people": people}


def iter_world(preset: str = "c3", n_agents: Optional[int] = None, seed: int = 1234,
               infected_fraction: float = 0.01, sets=None, edge_mult: float = 1.0, progress=None,
               geography: str = "random"):
    """The world of ``make_world`` piece by piece, in the order the generator draws them:
        ("header", {"preset", "n_agents", "age", "sex", "networks"})
        ("set", name, {"agent", "venue", "people"})     once per edge set
        ("state", {per-agent arrays})
    A consumer that keeps only its share of every set (distributed.RankPartitioner) never holds the whole COO."""
    if preset == "june":
        yield from _iter_june_world(10_000_000 if n_agents is None else n_agents, seed, infected_fraction, progress)
        return
    spec = PRESETS[preset]
    if n_agents is None:
        n_agents = DEFAULT_AGENTS[preset]
    if geography not in ("random", "clustered"):
        raise ValueError(f"geography {geography!r}: 'random' or 'clustered'")
    rng = np.random.default_rng(seed)
    A = int(n_agents)
    geo = super_area_map(A) if geography == "clustered" else None
    yield ("header", {"preset": preset, "n_agents": A, "age": rng.integers(0, 100, A, dtype=np.int64),
                      "sex": rng.integers(0, 2, A, dtype=np.int64), "networks": list(NETWORKS[preset]),
                      "n_sets": len([s for s in spec if sets is None or s in sets])})
    for name, (per_agent, dist) in spec.items():
        if sets is not None and name not in sets:
            continue
        es = _edge_set(rng, A, int(round(per_agent * edge_mult * A)), dist, geo, GEOGRAPHY.get(name, (0.3, 0.05)))
        if progress:
            progress(f"generated edge set {name}: {len(es['agent'])} edges")
        yield ("set", name, es)
        del es
    inf = (rng.random(A) < infected_fraction).astype(np.float32)
    yield ("state", {
        "max_infectiousness": rng.lognormal(0.0, 0.5, A).astype(np.float32),
        "shape": rng.normal(1.56, 0.08, A).astype(np.float32),
        "rate": rng.normal(0.53, 0.03, A).astype(np.float32),
        "shift": rng.normal(-2.12, 0.1, A).astype(np.float32),
        "is_infected": inf,
        "susceptibility": (1.0 - inf).astype(np.float32),
        "infection_time": (-10.0 * rng.random(A)).astype(np.float32) * inf,
        "current_stage": np.where(inf > 0, rng.integers(2, 6, A), 1).astype(np.float32),
    })


def _iter_june_world(n_agents: int, seed: int, infected_fraction: float, progress=None):
    """The "june" preset piece by piece (see ``JUNE_WORLD``)."""
    rng = np.random.default_rng(seed)
    A = int(n_agents)
    geo = super_area_map(A)
    age = rng.integers(0, 100, A, dtype=np.int64)
    sex = rng.integers(0, 2, A, dtype=np.int64)
    yield ("header", {"preset": "june", "n_agents": A, "age": age, "sex": sex, "networks": list(NETWORKS["june"]),
                      "n_sets": 6, "geography": "clustered"})
    everyone = np.arange(A, dtype=np.int64)
    es = _edge_set(rng, A, A, JUNE_WORLD["household"], geo, None, members=everyone)
    yield ("set", "household", es)
    taken = np.zeros(A, dtype=bool)                    # one primary activity per person (population/group_ids)
    for name in ("care_home", "school", "university", "company"):
        lo, hi, share, dist = JUNE_WORLD[name]
        who = (age >= lo) & (age < hi) & ~taken
        if share < 1.0:
            who &= rng.random(A) < share
        taken |= who
        es = _edge_set(rng, A, 0, dist, geo, GEOGRAPHY[name], members=everyone[who])
        if progress:
            progress(f"generated edge set {name}: {len(es['agent'])} edges")
        yield ("set", name, es)
        del es
    # leisure_loader.py:38-73: venue = super area, attended by everybody who lives in one of its k nearest super areas
    # (itself first); k nearest = itself and the first k - 1 surrounding cells that exist
    k, sa, n_sa = JUNE_WORLD["k_leisure"], geo["sa_agents"], geo["n_sa"]
    near = np.concatenate([np.arange(n_sa)[:, None], geo["nb"][:, :k - 1]], axis=1)            # [n_sa, k] (-1: none)
    home = everyone // sa
    agents, venues = [], []
    for c in range(k):
        src = near[:, c]                               # venue v is attended by the residents of super area near[v, c]
        ok = src >= 0
        v_of_src = np.flatnonzero(ok)
        # residents of super area s attend every venue v with near[v, c] == s
        order = np.argsort(src[ok], kind="stable")
        s_sorted, v_sorted = src[ok][order], v_of_src[order]
        first = np.searchsorted(s_sorted, home, side="left")
        last = np.searchsorted(s_sorted, home, side="right")
        cnt = last - first
        a_rep = np.repeat(everyone, cnt)
        offs = np.arange(len(a_rep)) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        agents.append(a_rep)
        venues.append(v_sorted[np.repeat(first, cnt) + offs])
    agent, venue = np.concatenate(agents), np.concatenate(venues)
    perm = rng.permutation(len(agent))
    agent, venue = agent[perm], venue[perm]
    yield ("set", "leisure", {"agent": agent, "venue": venue, "people": np.bincount(venue, minlength=n_sa).astype(np.int64)})
    del agent, venue
    inf = (rng.random(A) < infected_fraction).astype(np.float32)
    yield ("state", {
        "max_infectiousness": rng.lognormal(0.0, 0.5, A).astype(np.float32),
        "shape": rng.normal(1.56, 0.08, A).astype(np.float32),
        "rate": rng.normal(0.53, 0.03, A).astype(np.float32),
        "shift": rng.normal(-2.12, 0.1, A).astype(np.float32),
        "is_infected": inf,
        "susceptibility": (1.0 - inf).astype(np.float32),
        "infection_time": (-10.0 * rng.random(A)).astype(np.float32) * inf,
        "current_stage": np.where(inf > 0, rng.integers(2, 6, A), 1).astype(np.float32),
    })


def epidemic_state(n_agents: int, infected_fraction: float, seed: int) -> Dict:
    """Per-agent infection parameters and state at a given prevalence (a stream of its own: bench.py's second timed
    region re-seeds the SAME world at 30 % infected, where the transmission profile and the state updates cost most)."""
    rng = np.random.default_rng(seed)
    A = int(n_agents)
    inf = (rng.random(A) < infected_fraction).astype(np.float32)
    return {
        "max_infectiousness": rng.lognormal(0.0, 0.5, A).astype(np.float32),
        "shape": rng.normal(1.56, 0.08, A).astype(np.float32),
        "rate": rng.normal(0.53, 0.03, A).astype(np.float32),
        "shift": rng.normal(-2.12, 0.1, A).astype(np.float32),
        "is_infected": inf,
        "susceptibility": (1.0 - inf).astype(np.float32),
        "infection_time": (-10.0 * rng.random(A)).astype(np.float32) * inf,
        "current_stage": np.where(inf > 0, rng.integers(2, 6, A), 1).astype(np.float32),
    }


def make_world(preset: str = "c3", n_agents: Optional[int] = None, seed: int = 1234,
               infected_fraction: float = 0.01, sets=None, edge_mult: float = 1.0, progress=None,
               geography: str = "random") -> Dict:
    """Returns {"n_agents", "age", "sex", "edge_sets", "networks", "state"} as numpy arrays."""
    world: Dict = {"edge_sets": {}, "geography": geography}
    for piece in iter_world(preset, n_agents, seed, infected_fraction, sets, edge_mult, progress, geography):
        if piece[0] == "header":
            world.update(piece[1])
        elif piece[0] == "set":
            world["edge_sets"][piece[1]] = piece[2]
        else:
            world["state"] = piece[1]
    return world


def locality_permutation(n_agents: int, agent, venue):
    """(order, new_of) of ``reorder_agents`` from the edge set that defines the locality order (numpy arrays, or torch
    tensors on any device - the same permutation)."""
    if not isinstance(agent, np.ndarray) and hasattr(agent, "device"):
        import torch

        first = torch.full((n_agents,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=agent.device)
        first.scatter_reduce_(0, agent, venue, reduce="amin", include_self=True)
        order = torch.sort(first, stable=True)[1]
        new_of = torch.empty_like(order)
        new_of[order] = torch.arange(n_agents, device=agent.device)
        return order, new_of
    first = np.full(n_agents, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first, agent, venue)
    order = np.argsort(first, kind="stable")            # new position -> original id
    new_of = np.empty(n_agents, dtype=np.int64)
    new_of[order] = np.arange(n_agents)
    return order, new_of


def iter_world_torch(preset: str, n_agents: int, seed: int, device, infected_fraction: float = 0.01,
                     geography: str = "random", progress=None):
    """``iter_world`` drawn with torch's generator ON ``device``: the same pieces in the same order, every array a
    device tensor (int64 edge lists, ``people``; float32 state; int64 age / sex).  A 10^8-agent world streams in
    seconds where the numpy generator needs most of an hour, and nothing touches host memory - what the ranks of
    ``bench.py --gpus N`` consume (``distributed.stream_rank_share``: every rank draws the SAME seeded stream on its own
    GPU and keeps its share of each set before the next one is drawn; identical GPUs give identical streams).
    A different random world than ``iter_world(seed)``, with the same distributions; the rare duplicate
    (agent, venue) pairs are kept - the reference's format allows them."""
    import torch

    if geography not in ("random", "clustered"):
        raise ValueError(f"geography {geography!r}: 'random' or 'clustered'")
    if preset not in PRESETS:
        raise ValueError(f"preset {preset!r} is drawn by the numpy generator only (iter_world)")
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    A = int(n_agents)
    spec = PRESETS[preset]

    def sizes_for(dist, n_edges):
        kind = dist[0]
        if kind == "poisson1":
            mean = 1.0 + dist[1]
            draw = lambda m: 1 + torch.poisson(torch.full((m,), float(dist[1]), device=dev), generator=g).to(torch.int64)
        elif kind == "lognormal":
            mu, sigma, mean = float(np.log(dist[1]) - 0.5 * dist[2] ** 2), float(dist[2]), dist[1]
            draw = lambda m: torch.clamp(torch.exp(mu + sigma * torch.randn(m, device=dev, generator=g)), min=1.0).to(torch.int64)
        else:     # zipf(alpha) truncated to [1, cap]: inverse CDF over the finite support
            alpha, cap = dist[1], dist[2]
            pmf = torch.arange(1, cap + 1, device=dev, dtype=torch.float64) ** (-alpha)
            cdf = torch.cumsum(pmf / pmf.sum(), 0)
            mean = float((torch.arange(1, cap + 1, device=dev, dtype=torch.float64) * pmf).sum() / pmf.sum())
            draw = lambda m: 1 + torch.searchsorted(cdf, torch.rand(m, device=dev, dtype=torch.float64, generator=g)).clamp(max=cap - 1)
        sizes = torch.zeros(0, dtype=torch.int64, device=dev)
        while int(sizes.sum()) < n_edges:
            sizes = torch.cat([sizes, draw(int((n_edges - int(sizes.sum())) / mean * 1.05) + 16)])
        cs = torch.cumsum(sizes, 0)
        last = int(torch.searchsorted(cs, torch.tensor([n_edges], device=dev), right=False))
        sizes = sizes[: last + 1].clone()
        sizes[last] -= cs[last] - n_edges
        return sizes[sizes > 0]

    geo = None
    if geography == "clustered":
        m = super_area_map(A)
        geo = {"sa": m["sa_agents"], "n_sa": m["n_sa"], "nb": torch.from_numpy(m["nb"]).to(dev),
               "nb_n": torch.from_numpy(m["nb_n"]).to(dev)}

    def keys_of(agent, mix):        # synthetic._membership_keys on the device
        E = agent.numel()
        if mix is None:
            return agent.double() + (3.0 * torch.rand(E, device=dev, dtype=torch.float64, generator=g) - 1.5)
        p_near, p_leak = mix
        home = agent // geo["sa"]
        u = torch.rand(E, device=dev, generator=g)
        target = home.clone()
        near = (u >= 1.0 - p_near - p_leak) & (u < 1.0 - p_leak) & (geo["nb_n"][home] > 0)
        h = home[near]
        pick = (torch.rand(h.numel(), device=dev, generator=g) * geo["nb_n"][h]).to(torch.int64).clamp_(max=7)
        target[near] = geo["nb"][h, pick]
        leak = u >= 1.0 - p_leak
        target[leak] = torch.randint(0, geo["n_sa"], (int(leak.sum()),), device=dev, generator=g)
        return (target.double() + torch.rand(E, device=dev, dtype=torch.float64, generator=g)) * float(geo["sa"])

    yield ("header", {"preset": preset, "n_agents": A, "networks": list(NETWORKS[preset]), "n_sets": len(spec),
                      "geography": geography,
                      "age": torch.randint(0, 100, (A,), device=dev, generator=g),
                      "sex": torch.randint(0, 2, (A,), device=dev, generator=g)})
    for name, (per_agent, dist) in spec.items():
        E = int(round(per_agent * A))
        sizes = sizes_for(dist, E)
        V = sizes.numel()
        venue = torch.repeat_interleave(torch.arange(V, device=dev), sizes)
        base, extra = divmod(E, A)
        parts = [torch.arange(A, device=dev).repeat(base)] if base else []
        if extra:
            parts.append(torch.randperm(A, device=dev, generator=g)[:extra])
        agent = torch.cat(parts)
        del parts
        if geo is None:
            agent = agent[torch.randperm(E, device=dev, generator=g)]                # memberships dealt to venue slots
        else:
            agent = agent[torch.sort(keys_of(agent, GEOGRAPHY.get(name, (0.3, 0.05))), stable=True)[1]]
        perm = torch.randperm(E, device=dev, generator=g)                           # unsorted COO
        agent, venue = agent[perm].contiguous(), venue[perm].contiguous()
        del perm
        if progress:
            progress(f"drew edge set {name} on the device: {E} edges")
        yield ("set", name, {"agent": agent, "venue": venue, "people": sizes})
        del agent, venue
    f32 = dict(device=dev, dtype=torch.float32, generator=g)
    inf = (torch.rand(A, **f32) < infected_fraction).float()
    yield ("state", {
        "max_infectiousness": torch.exp(0.5 * torch.randn(A, **f32)),
        "shape": 1.56 + 0.08 * torch.randn(A, **f32),
        "rate": 0.53 + 0.03 * torch.randn(A, **f32),
        "shift": -2.12 + 0.1 * torch.randn(A, **f32),
        "is_infected": inf,
        "susceptibility": 1.0 - inf,
        "infection_time": -10.0 * torch.rand(A, **f32) * inf,
        "current_stage": torch.where(inf > 0, torch.randint(2, 6, (A,), device=dev, generator=g).float(),
                                     torch.ones(A, device=dev)),
    })


def make_world_torch(preset: str, n_agents: int, seed: int, device, infected_fraction: float = 0.01,
                     geography: str = "random") -> Dict:
    """A world of the same shape as ``make_world``'s, drawn with torch's generator ON ``device`` in seconds
    (``iter_world_torch`` assembled): the edge lists stay on the device as int64 tensors - what
    ``compile_plan(device=...)`` takes - and the per-agent arrays come back as numpy.  For the large-size property
    tests, which check the kernels against sums taken from the very same edge lists."""
    world: Dict = {"edge_sets": {}}
    for piece in iter_world_torch(preset, n_agents, seed, device, infected_fraction, geography):
        if piece[0] == "header":
            world.update({k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in piece[1].items()})
        elif piece[0] == "set":
            es = piece[2]
            world["edge_sets"][piece[1]] = {"agent": es["agent"], "venue": es["venue"], "people": es["people"].cpu().numpy()}
        else:
            world["state"] = {k: v.cpu().numpy() for k, v in piece[1].items()}
    return world


def reorder_agents(world: Dict, by: str = "household") -> Dict:
    """Renumber the agents so that the members of one venue of set ``by`` are consecutive (agents are
    ordered by their first venue in that set; agents without one keep their relative order at the
    end).  A graph-compile-time permutation for locality: tiles of that set concentrate on the
    slice/block diagonal, and in a multi-GPU partition most of its venues become rank-local.
    ``world["original_id"]`` maps new -> original agent id (results are reported through it)."""
    A = world["n_agents"]
    es = world["edge_sets"][by]
    order, new_of = locality_permutation(A, es["agent"], es["venue"])
    out = dict(world)
    out["age"], out["sex"] = world["age"][order], world["sex"][order]
    out["state"] = {k: v[order] for k, v in world["state"].items()}
    out["edge_sets"] = {k: {"agent": new_of[v["agent"]], "venue": v["venue"], "people": v["people"]}
                        for k, v in world["edge_sets"].items()}
    out["original_id"] = order if "original_id" not in world else world["original_id"][order]
    return out


def edge_set_of(network: str) -> str:
    return "leisure" if network in ("pub", "gym", "grocery", "visit", "cinema", "care_visit") else network


def algorithmic_bytes(world_or_sizes, networks) -> int:
    """B_step of SURVEY.md section 8(d) (fp32 values, int32 indices, no temporaries):
    8*sum_sets E_s + 8*sum_n E_n + 12*sum_n V_n + (8*N + 64)*A."""
    A = world_or_sizes["n_agents"]
    es = world_or_sizes["edge_sets"]
    E = {k: (len(v["agent"]) if "agent" in v else v["n_edges"]) for k, v in es.items()}
    V = {k: (len(v["people"]) if "people" in v else v["n_venues"]) for k, v in es.items()}
    sets = {edge_set_of(n) for n in networks}
    b = 8 * sum(E[s] for s in sets)
    b += 8 * sum(E[edge_set_of(n)] for n in networks)
    b += 12 * sum(V[edge_set_of(n)] for n in networks)
    b += (8 * len(networks) + 64) * A
    return int(b)


def network_edges(world_or_sizes, networks) -> int:
    es = world_or_sizes["edge_sets"]
    return int(sum((len(es[edge_set_of(n)]["agent"]) if "agent" in es[edge_set_of(n)] else es[edge_set_of(n)]["n_edges"])
                   for n in networks))
