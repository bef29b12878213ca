"""CPU: agent partition, halo / partial-sum exchange plans and the real torch.distributed collectives
(gloo, world_size 2) - the multi-GPU path minus the kernels, which are stood in by the tiled
layout's numpy emulation (tiling.emulate_pass1/2, itself pinned against bincount sums)."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from grad_june_amd.distributed import (HaloExchange, build_rank_world, choose_modes, emulate_exchange,
                                       partition_bounds)
from grad_june_amd.plan import compile_plan
from grad_june_amd.synthetic import make_world
from grad_june_amd.tiling import emulate_pass1, emulate_pass2


def small_world(n=6000, seed=5):
    w = make_world("c3", n_agents=n, seed=seed)
    w["state"]["transmission"] = np.random.default_rng(seed).random(n).astype(np.float32)
    return w


def reference_pass(world, x):
    """Global single-rank result per set: cum[v] (beta = 1) and per-agent sum of cum over its venues."""
    out = {}
    for name, es in world["edge_sets"].items():
        with np.errstate(divide="ignore"):
            pc = np.clip(1.0 / (es["people"].astype(np.float64) - 1), None, 1.0).clip(0.0)
        V = len(es["people"])
        cum = pc * np.bincount(es["venue"], weights=x[es["agent"]].astype(np.float64), minlength=V)
        acc = np.bincount(es["agent"], weights=cum[es["venue"]], minlength=world["n_agents"])
        out[name] = (cum, acc)
    return out


def rank_pass(rw, host, x_ext, all_reduce):
    """One rank's pass 1 + pass 2 on its compiled (tiled) plan, `all_reduce` combining partial sets."""
    cums = {}
    for s in host.sets:
        _, cums[s.name] = emulate_pass1(s.tiled, x_ext.copy() if len(x_ext) >= host.n_slices * host.slice_agents
                                        else np.pad(x_ext, (0, host.n_slices * host.slice_agents - len(x_ext))),
                                        host.slice_agents, beta=1.0)
    for name in cums:
        if rw.modes[name] == "partial":
            cums[name] = all_reduce(name, cums[name])
    accs = {}
    for s in host.sets:
        accs[s.name] = emulate_pass2(s.tiled, cums[s.name], host.n_slices * host.slice_agents, host.slice_agents)[: rw.n_local]
    return cums, accs


@pytest.mark.parametrize("R", [2, 3, 8])
def test_partition_matches_single_rank(R):
    world = small_world()
    x = world["state"]["transmission"]
    ref = reference_pass(world, x)
    modes = choose_modes(world, R)                 # the per-set rule (rounds 1-3), forced: every set whole
    assert set(modes.values()) == {"halo", "partial"}
    rws = [build_rank_world(world, r, R, modes, slice_agents=256) for r in range(R)]
    b = partition_bounds(world["n_agents"], R)
    xs = []
    for rw in rws:
        xe = np.zeros(rw.n_ext, dtype=np.float32)
        xe[: rw.n_local] = x[b[rw.rank]:b[rw.rank + 1]]
        xs.append(xe)
        assert rw.n_local_pad % rw.slice_agents == 0 or rw.n_halo == 0
        owner = np.searchsorted(b, rw.halo_global, side="right") - 1
        assert (owner != rw.rank).all() and np.array_equal(np.bincount(owner, minlength=R), rw.halo_from)
    emulate_exchange(rws, xs)
    hosts = [compile_plan(rw.n_local, rw.edge_sets, age=rw.age, sex=rw.sex, n_ext_agents=rw.n_ext, layout="tiled",
                          slices=(rw.n_slices, rw.slice_agents), sv_max=512, eb_target=4096) for rw in rws]
    # partial sets: emulate the all-reduce by summing every rank's partial cum
    partial_names = [n for n, m in modes.items() if m == "partial"]
    pre = {}
    for rw, host, xe in zip(rws, hosts, xs):
        for s in host.sets:
            if s.name in partial_names:
                pad = np.pad(xe, (0, host.n_slices * host.slice_agents - len(xe)))
                pre.setdefault(s.name, []).append(emulate_pass1(s.tiled, pad, host.slice_agents, beta=1.0)[1])
    total = {n: np.sum(v, axis=0) for n, v in pre.items()}
    for rw, host, xe in zip(rws, hosts, xs):
        cums, accs = rank_pass(rw, host, xe, lambda name, c: total[name])
        lo, hi = b[rw.rank], b[rw.rank + 1]
        for name in world["edge_sets"]:
            cum_ref, acc_ref = ref[name]
            if rw.modes[name] == "halo":     # local venue numbering: compare through venue_global
                assert np.allclose(cums[name], cum_ref[rw.venue_global[name]], rtol=1e-5, atol=1e-7), name
            else:
                assert np.allclose(cums[name], cum_ref, rtol=1e-5, atol=1e-7), name
            assert np.allclose(accs[name], acc_ref[lo:hi], rtol=1e-5, atol=1e-6), name


@pytest.mark.parametrize("R", [2, 5])
def test_heavy_tailed_sets_are_split(R):
    """BASELINE config 5's power-law venues: a set whose mean venue is small but whose edges lie in huge venues is cut
    into a local + halo half and a partial-sum half, venue by venue (classify_venues: n attendees on T ranks are a
    partial-sum venue when they are more than 8 and the halo form would move more floats, n (T - 1), than the venue costs
    in the all-reduce, 2 k (R - 1)); per agent the two halves add up to the single-rank pass, and no rank's halo is blown up by a big venue."""
    from grad_june_amd.distributed import SPLIT_SUFFIX, classify_venues, mode_of

    world = make_world("c5", n_agents=6000, seed=2)
    world["state"]["transmission"] = np.random.default_rng(1).random(6000).astype(np.float32)
    x = world["state"]["transmission"]
    ref = reference_pass(world, x)
    modes = choose_modes(world, R)
    assert set(modes.values()) == {"split"}           # (the per-set rule of rounds 1-3, GJ_EXCHANGE_RULE=set: cut at 8 attendees)
    assert mode_of(15, 10, R, np.array([1, 1, 1, 1, 1, 1, 1, 1, 1, 6])) == "halo"          # small mean, no tail
    assert mode_of(15_000_000, 6_000_000, R) == "halo" and mode_of(100, 5, R) == "partial" and mode_of(1, 1, 1) == "local"
    b = partition_bounds(world["n_agents"], R)
    rws = [build_rank_world(world, r, R, slice_agents=256) for r in range(R)]
    xs = []
    for rw in rws:
        assert set(rw.edge_sets) == {n + sfx for n in world["edge_sets"] for sfx in ("", SPLIT_SUFFIX)}
        for n in world["edge_sets"]:
            assert rw.modes[n] == "halo" and rw.modes[n + SPLIT_SUFFIX] == "partial"
            es = world["edge_sets"][n]
            k = 3 if n == "leisure" else 1            # pub, gym, grocery share the leisure set
            partial, n_att, T = classify_venues(es["agent"], es["venue"], len(es["people"]), b, k)
            assert np.array_equal(partial, (n_att > 8) & (n_att * (T - 1) > 2 * k * (R - 1))) and partial.any() and not partial.all()
            # the partial-sum half holds exactly those venues, whole; the other half only venues this rank touches
            assert np.array_equal(np.sort(rw.edge_sets[n + SPLIT_SUFFIX]["people"]), np.sort(es["people"][partial]))
            kept = rw.venue_global[n]                 # the set's ids of the venues this rank keeps of the other half
            assert not partial[kept].any() and ((n_att[kept] <= 8) | (n_att[kept] * (T[kept] - 1) <= 2 * k * (R - 1))).all()
            assert np.array_equal(rw.venue_global[n + SPLIT_SUFFIX], np.flatnonzero(partial))
        xe = np.zeros(rw.n_ext, dtype=np.float32)
        xe[: rw.n_local] = x[b[rw.rank]:b[rw.rank + 1]]
        xs.append(xe)
    emulate_exchange(rws, xs)
    hosts = [compile_plan(rw.n_local, rw.edge_sets, age=rw.age, sex=rw.sex, n_ext_agents=rw.n_ext, layout="tiled",
                          slices=(rw.n_slices, rw.slice_agents), sv_max=512, eb_target=4096) for rw in rws]
    pre = {}
    for rw, host, xe in zip(rws, hosts, xs):
        for s in host.sets:
            if rw.modes[s.name] == "partial":
                pad = np.pad(xe, (0, host.n_slices * host.slice_agents - len(xe)))
                pre.setdefault(s.name, []).append(emulate_pass1(s.tiled, pad, host.slice_agents, beta=1.0)[1])
    total = {n: np.sum(v, axis=0) for n, v in pre.items()}
    for rw, host, xe in zip(rws, hosts, xs):
        _, accs = rank_pass(rw, host, xe, lambda name, c: total[name])
        lo, hi = b[rw.rank], b[rw.rank + 1]
        for name in world["edge_sets"]:
            both = accs[name] + accs[name + SPLIT_SUFFIX]
            assert np.allclose(both, ref[name][1][lo:hi], rtol=1e-5, atol=1e-6), name
    # what the split buys: the halo of rank 0 against the same world with every set forced into halo mode
    forced = build_rank_world(world, 0, R, {n: "halo" for n in world["edge_sets"]}, slice_agents=256)
    assert rws[0].n_halo < 0.8 * forced.n_halo


def _ranks_equal_single_rank(world, R, slice_agents=256):
    """Every rank's pass 1 + pass 2 on its part (default exchange rule: per-venue classes) against the single-rank
    pass; returns the rank worlds and the partitioner's class counts."""
    from grad_june_amd.distributed import SPLIT_SUFFIX, RankPartitioner

    x = world["state"]["transmission"]
    ref = reference_pass(world, x)
    b = partition_bounds(world["n_agents"], R)
    part = RankPartitioner(world["n_agents"], R, networks=world["networks"], n_sets=len(world["edge_sets"]))
    for name, es in world["edge_sets"].items():
        part.add_set(name, es["agent"], es["venue"], es["people"])
    classes = dict(part.classes)
    by_rank = part.finish(world["age"], world["sex"], slice_agents)
    rws = [by_rank[r] for r in range(R)]
    xs = []
    for rw in rws:
        xe = np.zeros(rw.n_ext, dtype=np.float32)
        xe[: rw.n_local] = x[b[rw.rank]:b[rw.rank + 1]]
        xs.append(xe)
    emulate_exchange(rws, xs)
    hosts = [compile_plan(rw.n_local, rw.edge_sets, age=rw.age, sex=rw.sex, n_ext_agents=rw.n_ext, layout="tiled",
                          slices=(rw.n_slices, rw.slice_agents), sv_max=512, eb_target=4096) for rw in rws]
    pre = {}
    for rw, host, xe in zip(rws, hosts, xs):
        for s in host.sets:
            if rw.modes[s.name] == "partial":
                pad = np.pad(xe, (0, host.n_slices * host.slice_agents - len(xe)))
                pre.setdefault(s.name, []).append(emulate_pass1(s.tiled, pad, host.slice_agents, beta=1.0)[1])
    total = {n: np.sum(v, axis=0) for n, v in pre.items()}
    for rw, host, xe in zip(rws, hosts, xs):
        cums, accs = rank_pass(rw, host, xe, lambda name, c: total[name])
        lo, hi = b[rw.rank], b[rw.rank + 1]
        for name in world["edge_sets"]:
            both = sum(accs[n] for n in (name, name + SPLIT_SUFFIX) if n in accs)
            assert np.allclose(both, ref[name][1][lo:hi], rtol=1e-5, atol=1e-6), name
            for n in (name, name + SPLIT_SUFFIX):
                if n in cums:
                    vg = rw.venue_global[n]
                    assert np.allclose(cums[n], ref[name][0] if vg is None else ref[name][0][vg], rtol=1e-5, atol=1e-7), n
    return rws, classes


@pytest.mark.parametrize("R", [2, 4, 8])
def test_clustered_world_venues_are_classified_one_by_one(R):
    """A world with a geography (synthetic.GEOGRAPHY: households of neighbours, venues in the own / a neighbouring
    super area, a stated leak) under the household-major order: households are rank-local (no communication), only
    venues that reach across a rank boundary are communicated, and the halo is a small fraction of the random world's."""
    from grad_june_amd.synthetic import reorder_agents

    n = 120_000
    out = {}
    for geography in ("clustered", "random"):
        w = reorder_agents(make_world("c3", n_agents=n, seed=11, geography=geography), by="household")
        w["state"]["transmission"] = np.random.default_rng(2).random(n).astype(np.float32)
        out[geography] = _ranks_equal_single_rank(w, R)
    rws, classes = out["clustered"]
    rws_rnd, classes_rnd = out["random"]
    hh = classes["household"]
    assert hh["partial_sum"] == 0 and hh["local"] >= 0.995 * hh["venues"] and "household~big" not in rws[0].edge_sets
    assert all(rw.modes["household"] == "halo" for rw in rws)
    halo = max(rw.n_halo for rw in rws)
    assert halo < 0.35 * max(rw.n_halo for rw in rws_rnd)      # (a 15 000-agent rank is three super areas: mostly border)
    # small venues with every attendee on one rank never reach the all-reduce: the partial-sum buffer holds fewer
    # venues than the sets have (the per-set rule of rounds 1-3 all-reduced every care home and company)
    for s in ("care_home", "company"):
        c = classes[s]
        assert c["local"] + c["halo"] + c["partial_sum"] == c["venues"]
        if s + "~big" in rws[0].edge_sets:
            assert len(rws[0].edge_sets[s + "~big"]["people"]) == c["partial_sum"] < c["venues"]
    # the random world has no locality to use: only venues of one or two attendees are ever local there
    assert classes_rnd["company"]["local"] < classes["company"]["local"]
    assert classes_rnd["household"]["local"] < 0.9 * classes_rnd["household"]["venues"]


@pytest.mark.parametrize("geography", ["random", "clustered"])
@pytest.mark.parametrize("R", [2, 8])
def test_torch_streamed_share_equals_the_numpy_partition(geography, R):
    """``bench.py --gpus N`` draws the world with torch's generator on every rank's own GPU (iter_world_torch) and cuts
    the rank's part out with torch ops where the tensors live (here: CPU tensors, the same code path): array for array
    what the numpy partitioner makes of the assembled world - per-venue classes, split halves, halo lists, extended
    indices, the owned agents' state."""
    from grad_june_amd.distributed import stream_rank_share
    from grad_june_amd.synthetic import iter_world_torch, make_world_torch, reorder_agents

    n = 60_000
    w = make_world_torch("c3", n, 7, "cpu", geography=geography)
    wn = dict(w, edge_sets={k: {"agent": es["agent"].numpy(), "venue": es["venue"].numpy(), "people": es["people"]}
                            for k, es in w["edge_sets"].items()})
    wn = reorder_agents(wn, by="household")
    for r in (0, R - 1):
        ref = build_rank_world(wn, r, R)
        got, share = stream_rank_share(iter_world_torch("c3", n, 7, "cpu", geography=geography), r, R, reorder="household")
        assert (ref.n_local, ref.n_local_pad, ref.n_ext, ref.n_slices, ref.slice_agents) == \
               (got.n_local, got.n_local_pad, got.n_ext, got.n_slices, got.slice_agents)
        assert ref.modes == got.modes and list(ref.edge_sets) == list(got.edge_sets)
        assert np.array_equal(ref.halo_global, got.halo_global) and np.array_equal(ref.halo_from, got.halo_from)
        assert np.array_equal(ref.age, got.age) and np.array_equal(ref.sex, got.sex)
        for k in ref.edge_sets:
            for f in ("agent", "venue", "people"):
                b = got.edge_sets[k][f]
                assert np.array_equal(ref.edge_sets[k][f], b.numpy() if isinstance(b, torch.Tensor) else b), (k, f)
            v1, v2 = ref.venue_global[k], got.venue_global[k]
            assert (v1 is None and v2 is None) or np.array_equal(v1, v2), k
        a0, a1 = ref.bounds[r], ref.bounds[r + 1]
        for k, v in share["state"].items():
            assert np.array_equal(v, wn["state"][k][a0:a1]), k
        assert np.array_equal(share["original_id"], wn["original_id"][a0:a1])


def test_halo_agents_are_ordered_by_the_venue_that_needs_them():
    """The extended index range lists a rank's halo agents owner by owner (what the all-to-all delivers) and, inside an
    owner's group, by the halo set and venue that needs them: a halo slice then meets few venue blocks - its tiles are
    long instead of a handful of edges each (id order, rounds 1-3) - and the exchange still delivers the owners' values."""
    from grad_june_amd import distributed as D

    world = make_world("c5", n_agents=60_000, seed=4)
    x = np.random.default_rng(3).random(60_000).astype(np.float32)
    R, tiles = 4, {}
    for order in ("venue", "id"):
        D.HALO_ORDER = order
        try:
            rws = [build_rank_world(world, r, R, slice_agents=512) for r in range(R)]
        finally:
            D.HALO_ORDER = "venue"
        rw = rws[1]
        owner = np.searchsorted(rw.bounds, rw.halo_global, side="right") - 1
        assert (np.diff(owner) >= 0).all() and np.array_equal(np.bincount(owner, minlength=R), rw.halo_from)
        assert len(np.unique(rw.halo_global)) == rw.n_halo and (owner != 1).all()
        xs = []
        for w in rws:
            xe = np.zeros(w.n_ext, dtype=np.float32)
            xe[: w.n_local] = x[w.bounds[w.rank]:w.bounds[w.rank + 1]]
            xs.append(xe)
        emulate_exchange(rws, xs)
        assert np.array_equal(xs[1][rw.n_local_pad:], x[rw.halo_global])
        host = compile_plan(rw.n_local, rw.edge_sets, age=rw.age, sex=rw.sex, n_ext_agents=rw.n_ext, layout="tiled",
                            slices=(rw.n_slices, rw.slice_agents), sv_max=256, eb_target=1024)
        first_halo = rw.n_local_pad // rw.slice_agents
        n = 0
        for s in host.sets:
            if rw.modes[s.name] == "halo":
                J = s.tiled.n_blocks
                n += int((np.diff(s.tiled.tile_sptr[first_halo * J:]) > 0).sum())
        tiles[order] = n
        # every edge of a halo agent is still where the kernels look for it: pass 1 of the halo sets equals the global pass
        ref = reference_pass(world, x)
        pad = np.pad(xs[1], (0, host.n_slices * host.slice_agents - rw.n_ext))
        for s in host.sets:
            if rw.modes[s.name] == "halo":
                cum = emulate_pass1(s.tiled, pad, host.slice_agents, beta=1.0)[1]
                assert np.allclose(cum, ref[s.name.split("~")[0]][0][rw.venue_global[s.name]], rtol=1e-5, atol=1e-7), s.name
    assert tiles["venue"] < 0.6 * tiles["id"], tiles


def test_single_rank_is_all_local():
    world = small_world(2000)
    rw = build_rank_world(world, 0, 1)
    assert rw.n_halo == 0 and rw.n_ext == rw.n_local == 2000 and set(rw.modes.values()) == {"local"}
    for name, es in world["edge_sets"].items():
        assert np.array_equal(rw.edge_sets[name]["agent"], es["agent"])


def _gloo_worker(rank, R, port, ok):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=R)
    try:
        world = small_world(3000, seed=9)
        x = world["state"]["transmission"]
        b = partition_bounds(3000, R)
        rw = build_rank_world(world, rank, R, slice_agents=128)      # per-venue classes: most sets come in two halves
        assert any(n.endswith("~big") for n in rw.edge_sets) and {"halo", "partial"} <= set(rw.modes.values())
        xe = torch.zeros(rw.n_ext)
        xe[: rw.n_local] = torch.from_numpy(x[b[rank]:b[rank + 1]])
        halo = HaloExchange(rw, "cpu")
        halo.exchange(xe)                                   # the real all_to_all_single
        assert torch.equal(xe[rw.n_local_pad:], torch.from_numpy(x[rw.halo_global])), "halo values"
        assert sum(halo.send_counts) == halo.send_index.numel()
        # partial sums through the real all_reduce
        host = compile_plan(rw.n_local, rw.edge_sets, age=rw.age, sex=rw.sex, n_ext_agents=rw.n_ext, layout="tiled",
                            slices=(rw.n_slices, rw.slice_agents))
        ref = reference_pass(world, x)
        pad = np.pad(xe.numpy(), (0, host.n_slices * host.slice_agents - rw.n_ext))
        for s in host.sets:
            cum = emulate_pass1(s.tiled, pad, host.slice_agents, beta=1.0)[1]
            whole = ref[s.name.split("~")[0]][0]                     # the whole set's per-venue sums, by the set's venue ids
            vg = rw.venue_global[s.name]
            want = whole if vg is None else whole[vg]
            if rw.modes[s.name] == "partial":
                t = torch.from_numpy(cum.copy())
                dist.all_reduce(t)
                assert np.allclose(t.numpy(), want, rtol=1e-5, atol=1e-7), s.name
            else:
                assert np.allclose(cum, want, rtol=1e-5, atol=1e-7), s.name
        ok[rank] = 1
    finally:
        dist.destroy_process_group()


def test_gloo_world_size_2_collectives():
    R = 2
    ok = mp.get_context("spawn").Array("i", [0] * R)
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_gloo_worker, args=(R, port, ok), nprocs=R, join=True)
    assert list(ok) == [1] * R


def test_reorder_agents_is_a_pure_renumbering_and_shrinks_the_halo():
    from grad_june_amd.synthetic import reorder_agents

    world = small_world(20_000, seed=3)
    re = reorder_agents(world, by="household")
    oid = re["original_id"]
    assert np.array_equal(np.sort(oid), np.arange(20_000))
    for k, es in world["edge_sets"].items():       # same edge multiset through the id map
        a = np.sort(oid[re["edge_sets"][k]["agent"]] * 10**7 + re["edge_sets"][k]["venue"])
        b = np.sort(es["agent"] * 10**7 + es["venue"])
        assert np.array_equal(a, b), k
    for k in world["state"]:
        assert np.array_equal(re["state"][k], world["state"][k][oid])
    # members of a household are consecutive: the first-household key is non-decreasing
    first = np.full(20_000, 10**12)
    np.minimum.at(first, re["edge_sets"]["household"]["agent"], re["edge_sets"]["household"]["venue"])
    assert (np.diff(first[first < 10**12]) >= 0).all()
    h0 = build_rank_world(world, 1, 4).n_halo
    h1 = build_rank_world(re, 1, 4).n_halo
    assert h1 < 0.7 * h0


def test_reduce_groups_split_off_the_largest_partial_sum_set():
    from grad_june_amd.distributed import reduce_groups

    assert reduce_groups({}) == []
    assert reduce_groups({"a": 10}) == [["a"]]
    assert reduce_groups({"a": 10, "b": 20}) == [["b", "a"]]                          # too small to pipeline
    assert reduce_groups({"school": 30_000, "company": 750_000, "care_home": 300_000}) == \
        [["company"], ["care_home", "school"]]
    assert reduce_groups({"x": 5, "y": 5}, min_floats=1) == [["x"], ["y"]]               # ties: by name


@pytest.mark.parametrize("R,reorder", [(2, None), (3, "household"), (8, "household")])
def test_streamed_partition_equals_the_in_memory_one(R, reorder):
    """bench.py --gpus N: every rank generates the world set by set and keeps only its share
    (distributed.stream_rank_share) - the same RankWorld, array for array, as cutting the part out of the whole
    (re-ordered) world, and build_rank_worlds (one sort per set for all ranks) equals rank-by-rank scans."""
    from grad_june_amd.distributed import build_rank_world, build_rank_worlds, stream_rank_share
    from grad_june_amd.synthetic import iter_world, make_world, reorder_agents

    kw = dict(preset="c3", n_agents=6000, seed=11, infected_fraction=0.1)
    world = make_world(**kw)
    if reorder:
        world = reorder_agents(world, by=reorder)
    all_at_once = build_rank_worlds(world, R)
    for rank in range(R):
        ref = build_rank_world(world, rank, R)
        rw, share = stream_rank_share(iter_world(**kw), rank, R, reorder=reorder)
        for got in (rw, all_at_once[rank]):
            assert (got.n_local, got.n_local_pad, got.n_ext, got.n_slices, got.slice_agents) == \
                   (ref.n_local, ref.n_local_pad, ref.n_ext, ref.n_slices, ref.slice_agents)
            assert got.modes == ref.modes and list(got.edge_sets) == list(ref.edge_sets)
            for k in ("halo_global", "halo_from", "age", "sex", "bounds"):
                assert np.array_equal(getattr(got, k), getattr(ref, k)), k
            for name, es in ref.edge_sets.items():
                for k in ("agent", "venue", "people"):
                    assert np.array_equal(got.edge_sets[name][k], es[k]), (name, k)
                vg, vr = got.venue_global[name], ref.venue_global[name]
                assert (vg is None and vr is None) or np.array_equal(vg, vr)
        a0, a1 = int(ref.bounds[rank]), int(ref.bounds[rank + 1])
        for k, v in world["state"].items():
            assert np.array_equal(share["state"][k], v[a0:a1]), k
        assert share["total_edges"] == sum(len(es["agent"]) for es in world["edge_sets"].values())
        assert share["networks"] == world["networks"]
        if reorder:
            assert np.array_equal(share["original_id"], world["original_id"][a0:a1])
    with pytest.raises(ValueError):
        stream_rank_share(iter_world(**kw), 0, 2, reorder="school")


def test_twin_networks_of_split_sets():
    """expand_split_networks / with_twins: every network on a split set gets a twin on the set's partial-sum half (same
    beta, mask and table), the networks of one edge set stay adjacent (the launch groups networks by set), and sets
    that were not split are left alone."""
    from grad_june_amd import _native as N
    from grad_june_amd.distributed import SPLIT_SUFFIX, expand_split_networks, with_twins
    from grad_june_amd.plan import NetworkSpec

    tab = np.ones((2, 2, 100), np.float32)
    specs = [NetworkSpec("school", "school", N.MASK_Q, None), NetworkSpec("pub", "leisure", N.MASK_QL, tab),
             NetworkSpec("gym", "leisure", N.MASK_QL, 2 * tab), NetworkSpec("household", "household", N.MASK_RAW, None)]
    names = ["school", "pub", "gym", "household"]
    betas = {"school": 0.5, "pub": 0.1, "gym": 0.2, "household": 0.4}
    sets = {"school": 0, "school" + SPLIT_SUFFIX: 0, "leisure": 0, "leisure" + SPLIT_SUFFIX: 0, "household": 0}
    s2, n2, b2 = expand_split_networks(specs, names, betas, sets)
    assert n2 == ["school", "school~big", "pub", "gym", "pub~big", "gym~big", "household"]
    by = {sp.name: sp for sp in s2}
    assert set(by) == set(n2) and by["gym~big"].edge_set == "leisure~big" and by["gym~big"].mask_kind == N.MASK_QL
    assert by["gym~big"].table is by["gym"].table and b2["gym~big"] == 0.2 and b2["school~big"] == 0.5 and "household~big" not in b2
    seen = []
    for n in n2:                                   # adjacency: a set's networks form one run
        es = by[n].edge_set
        assert es not in seen[:-1] or seen[-1] == es
        if not seen or seen[-1] != es:
            seen.append(es)
    # nothing split: unchanged
    s3, n3, b3 = expand_split_networks(specs, names, betas, {"school": 0, "leisure": 0, "household": 0})
    assert n3 == names and len(s3) == len(specs) and b3 == betas
    assert with_twins(["a", "b"], {"a": "x", "b": "y"}.get, lambda n: None) == ["a", "b"]


def test_split_mode_stays_inside_the_library_limits():
    """A split set becomes two edge sets and every network on it a pair: six heavy-tailed sets carrying the reference's
    eleven networks would need 12 sets / 22 networks (GJ_MAX_SETS 12, GJ_MAX_NETS 16).  The partitioner splits while the
    limits hold and runs the remaining sets unsplit (partial sums or halo, from global sizes) - and says so - instead of
    failing later in compile_plan."""
    from grad_june_amd import _native as N
    from grad_june_amd.distributed import RankPartitioner, expand_split_networks
    from grad_june_amd.plan import NetworkSpec
    from grad_june_amd.synthetic import edge_set_of

    world = make_world("c5", n_agents=30_000, seed=3)
    networks = ["school", "university", "company", "care_home", "pub", "gym", "grocery", "visit", "care_visit", "cinema",
                "household"]
    said = []
    want = {s: "split" for s in world["edge_sets"]}      # (at 10^8 agents mode_of asks for this by itself)
    part = RankPartitioner(world["n_agents"], 4, [1], modes=want, networks=networks, n_sets=len(world["edge_sets"]),
                           log=said.append)
    for name, es in world["edge_sets"].items():
        part.add_set(name, es["agent"], es["venue"], es["people"])
    modes = dict(part.modes)
    rw = part.finish(world["age"], world["sex"])[1]
    split = [s for s in modes if s.endswith("~big")]
    assert len(split) == 5 and "leisure~big" not in modes and modes["leisure"] in ("partial", "halo")
    assert len(said) == 1 and "leisure" in said[0] and "not split" in said[0]
    specs = [NetworkSpec(n, edge_set_of(n), N.MASK_RAW if n == "household" else N.MASK_Q, None) for n in networks]
    out_specs, names, _ = expand_split_networks(specs, networks, {n: 1.0 for n in networks}, rw.edge_sets)
    assert len(rw.edge_sets) <= N.GJ_MAX_SETS and len(names) <= N.GJ_MAX_NETS
    per_set = {}
    for sp in out_specs:
        per_set[sp.edge_set] = per_set.get(sp.edge_set, 0) + 1
    assert max(per_set.values()) <= N.GJ_MAX_NETS_PER_SET
    # the benchmark's own configuration (8 networks) still splits every set, as before
    part = RankPartitioner(world["n_agents"], 4, [0], modes=want, networks=world["networks"],
                           n_sets=len(world["edge_sets"]))
    for name, es in world["edge_sets"].items():
        assert part.add_set(name, es["agent"], es["venue"], es["people"]) == "split"
