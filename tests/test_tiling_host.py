"""CPU: the tiled (propagation-blocked) layout - structure invariants and a numpy emulation of the
four streaming phases against plain bincount sums."""
import numpy as np
import pytest

from grad_june_amd.tiling import (build_tiled, choose_slices, emulate_pass1, emulate_pass2, venue_blocks)


def random_set(rng, A, V, E, big=None):
    venue = rng.integers(0, V, E)
    if big:
        venue[: big] = min(3, V - 1)           # one huge venue
    agent = rng.integers(0, A, E)
    return agent, venue


@pytest.mark.parametrize("A,V,E,sa,svmax,eb", [(5000, 700, 20000, 512, 128, 3000), (300, 5, 4000, 64, 65535, 1 << 30),
                                               (10000, 9000, 15000, 1024, 4096, 2000), (64, 1, 10, 64, 16, 16),
                                               (2000, 3000, 6000, 64, 16, 16), (3000, 600, 30000, 128, 32, 600)])
@pytest.mark.parametrize("wide", [False, True], ids=["narrow-desc", "wide-desc"])
def test_layout_and_emulation(A, V, E, sa, svmax, eb, wide):
    rng = np.random.default_rng(A + V)
    agent, venue = random_set(rng, A, V, E, big=E // 3)
    S = -(-A // sa)
    pc = rng.random(V).astype(np.float32)
    cls = rng.integers(0, 200, A).astype(np.uint8)
    t = build_tiled("x", agent, venue, V, pc, S, sa, agent_class=cls, sv_max=svmax, eb_target=eb, wide=wide)
    assert t.desc_wide == wide and t.chunk_desc.shape[1] == (8 if wide else 4)
    J = t.n_blocks
    assert t.blk_v0[0] == 0 and t.blk_v0[-1] == V and t.blk_e0[-1] == t.n_slots and t.tile_sptr[-1] == E
    assert (np.diff(t.blk_v0) <= svmax).all() and (np.diff(t.blk_v0) > 0).all()
    deg = np.bincount(venue, minlength=V)
    blk_edges = np.add.reduceat(deg, t.blk_v0[:-1])
    assert (t.blk_e0 % 8 == 0).all() and np.array_equal(np.diff(t.blk_e0), -(-blk_edges // 8) * 8)
    assert (t.e_lv != 0xFFFF).sum() == E
    # chunk descriptors: every slice-major edge resolves to the block-major slot the tile tables give
    seg = t.tile_sptr[0:S * J + 1:J]
    tile_of_pos = np.searchsorted(t.tile_sptr, np.arange(E), side="right") - 1
    slot_ref = t.tile_jpos[tile_of_pos] + (np.arange(E) - t.tile_sptr[tile_of_pos])
    n_multi = 0
    for sl in range(S):
        for c in range(t.chunk_ptr[sl], t.chunk_ptr[sl + 1]):
            i0 = seg[sl] + 64 * (c - t.chunk_ptr[sl])
            n = min(64, seg[sl + 1] - i0)
            if wide:
                d = [int(x) for x in t.chunk_desc[c]]
                starts = [0] + [(d[6] >> (8 * k)) & 0xFF for k in range(4)] + [d[7] & 0xFF]
                multi, j0 = (d[7] >> 8) & 1, (d[7] & 0xFFFFFFFF) >> 9
                if multi and t.slot_idx is None:   # the j0 field of a chunk the descriptor cannot express: its row of explicit slots
                    assert np.array_equal(t.multi_slots[j0][:n], slot_ref[i0:i0 + n]) and (t.multi_slots[j0][n:] == 0).all()
                elif not multi or t.slot_idx is not None:
                    assert t.tile_sptr[sl * J + j0] <= i0 < t.tile_sptr[sl * J + j0 + 1]
                assert starts == sorted(starts) and all(0 < st <= 64 for st in starts[1:])
                lane = np.arange(n)
                seg_of = sum((lane >= st).astype(int) for st in starts[1:])
                got = np.array(d[:6])[seg_of] + lane
                n_tiles = len(np.unique(tile_of_pos[i0:i0 + n]))
                assert multi == (n_tiles > 6)
                if multi:
                    n_multi += 1
                    ok = lane < starts[5]            # the first five segments are still exact
                    assert np.array_equal(got[ok], slot_ref[i0:i0 + n][ok])
                else:
                    assert np.array_equal(got, slot_ref[i0:i0 + n])
                continue
            slot0, slot1, sm, j0 = (int(x) for x in t.chunk_desc[c])
            split, multi = sm & 0xFFFF, sm >> 16
            assert 1 <= split <= n
            assert np.array_equal(slot_ref[i0:i0 + split], slot0 + np.arange(split))
            if multi:
                n_multi += 1
                assert np.array_equal(t.multi_slots[j0][:n], slot_ref[i0:i0 + n]) and (t.multi_slots[j0][n:] == 0).all()
            else:
                assert t.tile_sptr[sl * J + j0] <= i0 < t.tile_sptr[sl * J + j0 + 1]
                assert np.array_equal(slot_ref[i0 + split:i0 + n], slot1 + np.arange(n - split))
    if t.slot_idx is None:         # (a set in the explicit-slot form carries every edge's slot instead)
        assert (t.multi_slots is None and n_multi == 0) or len(t.multi_slots) == n_multi     # one row per such chunk, chunk order
    # every tile is contiguous in both orders and holds the same multiset of edges
    lens = np.diff(t.tile_sptr).reshape(S, J)
    assert lens.sum() == E
    x = rng.random(S * sa).astype(np.float32)
    val, cum = emulate_pass1(t, x, sa, beta=0.7)
    ref = np.float32(0.7) * pc * np.bincount(venue, weights=x[agent].astype(np.float64), minlength=V).astype(np.float32)
    assert np.allclose(cum, ref, rtol=1e-5, atol=1e-7)
    # val in block-major order is x[agent] of the edge stored there
    assert np.allclose(np.sort(val[t.e_lv != 0xFFFF]), np.sort(x[agent]))
    acc = emulate_pass2(t, cum, A, sa)
    ref2 = np.bincount(agent, weights=cum[venue].astype(np.float64), minlength=A)
    assert np.allclose(acc, ref2, rtol=1e-5, atol=1e-6)
    # leisure-style tables ride on the per-edge class
    tab = rng.random(200).astype(np.float32)
    _, cum_l = emulate_pass1(t, x, sa, beta=1.0, table=tab)
    ref_l = pc * np.bincount(venue, weights=(tab[cls[agent]] * x[agent]).astype(np.float64), minlength=V).astype(np.float32)
    assert np.allclose(cum_l, ref_l, rtol=1e-5, atol=1e-7)
    acc_l = emulate_pass2(t, cum_l, A, sa, weight_table=tab)
    ref_l2 = np.bincount(agent, weights=(tab[cls[agent]] * cum_l[venue]).astype(np.float64), minlength=A)
    assert np.allclose(acc_l, ref_l2, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("A,V,E,sa,svmax,eb", [(5000, 40, 60000, 512, 16, 20000), (3000, 6, 30000, 128, 65535, 1 << 30),
                                               (2000, 3000, 6000, 64, 16, 16)])
@pytest.mark.parametrize("explicit", [None, True], ids=["descriptors", "explicit-slots"])
def test_tiles_padded_to_whole_sectors(A, V, E, sa, svmax, eb, explicit):
    """build_tiled(tile_pad=16), the experiment of DESIGN section 8: every tile's run is a multiple of 16 positions in both
    orders (pads: local agent 0 / local venue 0xFFFF), so every piece of a 64-edge chunk covers whole 64-byte sectors
    of `val`; both passes give what the unpadded layout gives; a set of tiny tiles is left alone."""
    rng = np.random.default_rng(A + E)
    agent, venue = random_set(rng, A, V, E)
    S = -(-A // sa)
    pc = rng.random(V).astype(np.float32)
    cls = rng.integers(0, 200, A).astype(np.uint8)
    kw = dict(agent_class=cls, sv_max=svmax, eb_target=eb, explicit=explicit, wide=True if explicit else None)
    t0 = build_tiled("x", agent, venue, V, pc, S, sa, **kw)
    t = build_tiled("x", agent, venue, V, pc, S, sa, tile_pad=16, **kw)
    J = t.n_blocks
    lens = np.diff(t.tile_sptr)
    tiny = E < 64 * (np.diff(t0.tile_sptr) > 0).sum()
    if tiny:                                      # mean tile below 64 edges: unpadded, the same arrays
        assert np.array_equal(t.a_la, t0.a_la) and np.array_equal(t.e_lv, t0.e_lv)
        return
    assert (lens % 16 == 0).all() and (t.tile_jpos % 16 == 0).all() and (t.tile_sptr % 16 == 0).all()
    assert (t.e_lv != 0xFFFF).sum() == E and t.n_edges == t.tile_sptr[-1] == len(t.a_la) and t.n_edges < E + 16 * S * J
    # a chunk's pieces: starts and lengths are multiples of 16 (whole sectors of 4-byte values)
    if t.slot_idx is not None:
        assert len(t.slot_idx) == t.n_edges
        brk = np.flatnonzero(np.diff(t.slot_idx) != 1) + 1
        assert (brk % 16 == 0).all() and (t.slot_idx[np.concatenate([[0], brk])] % 16 == 0).all()
    x = rng.random(S * sa).astype(np.float32)
    tab = rng.random(200).astype(np.float32)
    for table in (None, tab):
        _, cum0 = emulate_pass1(t0, x, sa, beta=0.7, table=table)
        _, cum = emulate_pass1(t, x, sa, beta=0.7, table=table)
        assert np.array_equal(cum, cum0)
        assert np.allclose(emulate_pass2(t, cum, A, sa, weight_table=table), emulate_pass2(t0, cum0, A, sa, weight_table=table),
                           rtol=1e-6, atol=1e-7)


def test_empty_set_and_slice_choice():
    t = build_tiled("e", np.zeros(0, np.int64), np.zeros(0, np.int64), 0, np.zeros(0, np.float32), 4, 64)
    assert t.n_blocks == 0 and t.n_edges == 0
    t = build_tiled("e", np.zeros(0, np.int64), np.zeros(0, np.int64), 10, np.ones(10, np.float32), 4, 64)
    assert t.n_blocks == 1 and t.n_edges == 0 and t.blk_v0.tolist() == [0, 10]
    for n in (1, 100, 769, 10_000, 1_000_000, 10_000_000, 25_000_000):
        S, SA = choose_slices(n)
        assert S * SA >= n and (S - 1) * SA < n and SA <= 20480
    assert choose_slices(10_000_000)[0] in (511, 512)
    b = venue_blocks(np.array([5, 5, 100000, 5, 5]), sv_max=2, eb_target=50)
    assert b.tolist() == [0, 2, 3, 5]


@pytest.mark.parametrize("A,V,E,sa,owned", [(5000, 700, 9000, 512, 5000), (300, 5, 400, 64, 300), (1000, 65534, 1500, 128, 700),
                                            (64, 1, 64, 64, 64)])
def test_ell_form_of_pass2(A, V, E, sa, owned):
    """tiling.build_ell + the emulation of phase D's direct form against bincount sums:
    agents beyond ``owned`` are halo agents and get no row; groups of venues give the same result."""
    from grad_june_amd.tiling import build_ell, direct_columns, direct_eligible, ell_rows, emulate_direct_pass2

    rng = np.random.default_rng(A + E)
    agent = np.concatenate([np.arange(min(A, E)), rng.integers(0, A, max(0, E - A))])    # degree 1 or 2, some 3+
    rng.shuffle(agent)
    venue = rng.integers(0, V, len(agent))
    S_owned = -(-owned // sa)
    ell3, K = build_ell(agent, venue, owned, S_owned, sa)
    deg = np.bincount(agent[agent < owned], minlength=owned)
    assert K == direct_columns(int(deg.max())) and ell3.dtype == np.uint16
    assert ell3.shape == (max(1, K // 2), S_owned * sa, min(K, 2)) and ell3.flags["C_CONTIGUOUS"]
    ell = ell_rows(ell3)                       # [rows, K]: plane p holds columns 2p, 2p + 1
    assert ((ell != 0xFFFF).sum(1)[:owned] == deg).all() and (ell[owned:] == 0xFFFF).all()
    for a in (0, owned // 2, owned - 1):       # COO order inside a row
        assert np.array_equal(ell[a][: deg[a]], venue[agent == a])
    cum = rng.random(V).astype(np.float32)
    ref = np.bincount(agent[agent < owned], weights=cum[venue[agent < owned]].astype(np.float64), minlength=owned)
    for gv in (None, max(1, V // 3)):
        acc = emulate_direct_pass2(ell3 if gv else ell, cum, owned, group_venues=gv)
        assert np.allclose(acc, ref, rtol=1e-5, atol=1e-6)
    # per-class weights over several networks of one set (leisure)
    cls = rng.integers(0, 200, A).astype(np.uint8)
    w = rng.random((3, 200)).astype(np.float32)
    cum3 = rng.random((V, 3)).astype(np.float32)
    own = agent < owned
    ref3 = np.bincount(agent[own], weights=(w[:, cls[agent[own]]].T.astype(np.float64) * cum3[venue[own]]).sum(1),
                       minlength=owned)
    assert np.allclose(emulate_direct_pass2(ell, cum3, owned, weights=w, agent_class=cls), ref3, rtol=1e-5, atol=1e-6)
    assert direct_eligible(V, int(own.sum()), owned, int(deg.max()), 1, sa) == (int(deg.max()) <= 8 and K * owned <= 4 * own.sum())
    assert not direct_eligible(65535, 10, 10, 1, 1, 64) and not direct_eligible(100, 0, 10, 0, 1, 64)
    assert not direct_eligible(60000, 10**6, 10**6, 2, 6, 20480)      # 360 k table floats: more than four groups
    assert direct_eligible(30000, 10**6, 10**6, 2, 1, 20480)


def test_compile_plan_chooses_the_direct_form():
    """compile_plan(direct=None) marks the sets whose sizes allow the direct form (few venues, bounded agent degree);
    direct=False disables it; naming an ineligible set is an error; the plan round-trips through save / load."""
    import os
    import tempfile

    from grad_june_amd.plan import compile_plan, load_plan, save_plan
    from grad_june_amd.synthetic import make_world

    w = make_world("c3", n_agents=30_000, seed=5)
    host = compile_plan(w["n_agents"], w["edge_sets"], age=w["age"], sex=w["sex"], layout="tiled")
    got = {s.name: s.tiled.ell_k for s in host.sets}
    assert got["household"] in (1, 2, 4, 8) and all(got[n] == 2 for n in ("school", "university", "leisure"))   # tiny world: all fit
    off = compile_plan(w["n_agents"], w["edge_sets"], age=w["age"], sex=w["sex"], layout="tiled", direct=False)
    assert all(s.tiled.ell_k == 0 and s.tiled.ell is None for s in off.sets)
    only = compile_plan(w["n_agents"], w["edge_sets"], age=w["age"], sex=w["sex"], layout="tiled", direct=("school",))
    assert {s.name for s in only.sets if s.tiled.ell_k} == {"school"}
    big = dict(w["edge_sets"])
    big["household"] = dict(big["household"], people=np.ones(70_000, dtype=np.int64))     # > 65534 venues
    with pytest.raises(ValueError):
        compile_plan(w["n_agents"], big, age=w["age"], sex=w["sex"], layout="tiled", direct=("household",))
    auto = compile_plan(w["n_agents"], big, age=w["age"], sex=w["sex"], layout="tiled")
    assert {s.name: s.tiled.ell_k for s in auto.sets}["household"] == 0
    with tempfile.TemporaryDirectory() as d:
        save_plan(host, os.path.join(d, "p.npz"))
        back = load_plan(os.path.join(d, "p.npz"))
    for a, b in zip(host.sets, back.sets):
        assert a.tiled.ell_k == b.tiled.ell_k and np.array_equal(a.tiled.ell, b.tiled.ell)


def test_local_venue_index_never_collides_with_the_pad_marker():
    """e_lv is 16-bit and 0xFFFF marks a pad slot: a block of 65536 venues would give its last venue that index and
    silently drop its edges - refused at compile time (tiling.py and the C ABI's gj_compile_* alike)."""
    with pytest.raises(ValueError, match="16-bit"):
        build_tiled("x", np.zeros(4, np.int64), np.arange(4), 70000, np.ones(70000, np.float32), 1, 64, sv_max=65536)
    t = build_tiled("x", np.arange(64) % 8, np.arange(64) * 1000, 70000, np.ones(70000, np.float32), 1, 64,
                       sv_max=65535, eb_target=1 << 30)
    assert int(np.diff(t.blk_v0).max()) <= 65535 and int(t.e_lv[t.e_lv != 0xFFFF].max()) <= 65534


def _household_major(A, V, mem, rng):
    """A set in the household-major order: every agent gets `mem` memberships (mem = 1: the reference's worlds, one
    household per person), agents renumbered by their smallest venue; unsorted COO."""
    agent = np.repeat(np.arange(A), mem) if isinstance(mem, int) else np.repeat(np.arange(A), mem)
    venue = rng.integers(0, V, len(agent))
    first = np.full(A, np.iinfo(np.int64).max)
    np.minimum.at(first, agent, venue)
    order = np.argsort(first, kind="stable")
    new_of = np.empty(A, dtype=np.int64)
    new_of[order] = np.arange(A)
    perm = rng.permutation(len(agent))
    return new_of[agent][perm], venue[perm]


@pytest.mark.parametrize("A,V,mem,sa,halo", [(5000, 1800, 2, 512, 0), (5000, 2500, 1, 512, 0), (3000, 900, 3, 256, 700),
                                             (700, 5000, 1, 64, 0)])
def test_run_form_of_the_set_that_orders_the_agents(A, V, mem, sa, halo):
    """tiling.split_primary_runs + finish_run_form: one edge per owned agent leaves the tiled arrays and both passes
    (emulated) still give the sums of the whole set - also when EVERY edge is primary (mem = 1: the tiled arrays are
    empty) and with halo agents, whose edges all stay tiled."""
    from grad_june_amd import tiling as TL

    rng = np.random.default_rng(A + V + mem)
    agent, venue = _household_major(A, V, mem, rng)
    n_own_slices = -(-A // sa)
    if halo:        # halo agents sit behind the owned slices and attend random venues
        ha = n_own_slices * sa + rng.integers(0, halo, 2 * halo)
        agent, venue = np.concatenate([agent, ha]), np.concatenate([venue, rng.integers(0, V, 2 * halo)])
        perm = rng.permutation(len(agent))
        agent, venue = agent[perm], venue[perm]
    n_ext = n_own_slices * sa + halo if halo else A
    S = -(-n_ext // sa)
    rf = TL.split_primary_runs(agent, venue, A, V, sa)
    assert rf is not None and rf.n_primary == A and int((~rf.keep).sum()) == A
    # the primary edge of an agent: the first COO edge to its smallest venue
    for a in rng.integers(0, A, 50):
        e = np.flatnonzero(agent == a)
        assert venue[e].min() == rf.vmin[a]
        assert not rf.keep[e[venue[e] == rf.vmin[a]][0]] and rf.keep[e].sum() == len(e) - 1
    pc = rng.random(V).astype(np.float32)
    t = build_tiled("hh", agent[rf.keep], venue[rf.keep], V, pc, S, sa, sv_max=300, eb_target=2000)
    t.runs = TL.finish_run_form(rf, t.blk_v0, A, sa)
    assert t.n_edges == len(agent) - A and (t.n_edges == 0) == (mem == 1 and not halo)
    assert t.runs.blk_r0[0] == 0 and t.runs.blk_r0[-1] == A and (np.diff(t.runs.blk_r0) >= 0).all()
    x = rng.random(S * sa).astype(np.float32)
    x[A:n_own_slices * sa] = 0
    _, cum = emulate_pass1(t, x, sa, 0.7)
    ref = np.float32(0.7) * pc * np.bincount(venue, weights=x[agent].astype(np.float64), minlength=V).astype(np.float32)
    assert np.allclose(cum, ref, rtol=1e-5, atol=1e-7)
    acc = emulate_pass2(t, cum, A, sa)
    own = agent < A
    ref2 = np.bincount(agent[own], weights=cum[venue[own]].astype(np.float64), minlength=A)
    assert np.allclose(acc, ref2, rtol=1e-5, atol=1e-7)
    # a world that is NOT in that order has no run form; neither has one whose windows would not fit
    shuffled = rng.permutation(A)[agent[agent < A]]
    assert TL.split_primary_runs(shuffled, venue[agent < A], A, V, sa) is None
    if V >= 5000:
        assert t.runs.max_window <= TL.RUN_MAX_WINDOW
        assert TL.split_primary_runs(agent, venue * 100, A, V * 100, sa) is None


def test_compile_plan_run_form_choice_and_roundtrip(tmp_path):
    from grad_june_amd.plan import compile_plan, load_plan, save_plan
    from grad_june_amd.synthetic import make_world, reorder_agents

    w = reorder_agents(make_world("c3", n_agents=30_000, seed=4), by="household")
    # at this size every set is small enough for the direct form: the run form must be asked for
    host = compile_plan(w["n_agents"], w["edge_sets"], age=w["age"], sex=w["sex"], layout="tiled", runs=("household",))
    by = {s.name: s.tiled for s in host.sets}
    assert by["household"].runs is not None and by["household"].ell_k == 0
    assert by["household"].runs.n_primary == w["n_agents"]
    assert by["household"].n_edges == len(w["edge_sets"]["household"]["agent"]) - w["n_agents"]
    assert all(by[s].runs is None for s in by if s != "household")
    with pytest.raises(ValueError, match="ordered by their smallest venue"):
        compile_plan(w["n_agents"], w["edge_sets"], age=w["age"], sex=w["sex"], layout="tiled", runs=("company",))
    none = compile_plan(w["n_agents"], w["edge_sets"], age=w["age"], sex=w["sex"], layout="tiled", runs=False)
    assert all(s.tiled.runs is None for s in none.sets)
    path = tmp_path / "plan.npz"
    save_plan(host, path)
    back = load_plan(path)
    rb = {s.name: s.tiled for s in back.sets}["household"]
    for k in ("pv_blk", "pv_win", "blk_r0", "win_lo", "win_n"):
        assert np.array_equal(np.asarray(getattr(rb.runs, k)), np.asarray(getattr(by["household"].runs, k))), k
    assert rb.n_edges == by["household"].n_edges and rb.runs.max_window == by["household"].runs.max_window


def test_explicit_slots_replace_descriptors_for_tiny_tiles():
    """Tiles of a few edges: more than six per 64-edge chunk is beyond the wide descriptor and every lane would walk
    the tile tables; build_tiled then carries slot_idx, the block-major slot of every slice-major edge."""
    rng = np.random.default_rng(3)
    A, V, E, sa = 4000, 3000, 9000, 64
    agent, venue = rng.integers(0, A, E), rng.integers(0, V, E)
    pc = np.ones(V, np.float32)
    t = build_tiled("x", agent, venue, V, pc, -(-A // sa), sa, sv_max=16, eb_target=16)
    assert t.desc_wide and t.slot_idx is not None                       # chosen by itself: ~2 edges per tile
    # slot_idx agrees with the tile tables (what the descriptors encode)
    S, J = t.n_slices, t.n_blocks
    for s_ in range(0, S, 7):
        for j in range(0, J, 11):
            a, b = t.tile_sptr[s_ * J + j], t.tile_sptr[s_ * J + j + 1]
            assert np.array_equal(t.slot_idx[a:b], t.tile_jpos[s_ * J + j] + np.arange(b - a))
    # every edge has a slot of its own, none is a pad slot
    assert len(np.unique(t.slot_idx)) == E and (t.e_lv[t.slot_idx] != 0xFFFF).all()
    big = build_tiled("x", agent, venue, V, pc, -(-A // sa), sa, sv_max=4096, eb_target=1 << 20)
    assert big.slot_idx is None                                         # ordinary tiles keep their descriptors
    forced = build_tiled("x", agent, venue, V, pc, -(-A // sa), sa, sv_max=4096, eb_target=1 << 20, explicit=True)
    assert forced.slot_idx is not None and len(forced.slot_idx) == E


@pytest.mark.parametrize("sv_max,eb", [(16, 16), (64, 256), (256, 2048), (4096, 1 << 20)])
def test_no_compiled_set_walks_the_tile_tables(sv_max, eb):
    """The fence of the geometry cliff (round 2: one tuner candidate at 7.03 ms against 0.108; round 3 reproduced it at
    40.9 ms against 1.13 on a C5 world with tiles of a few edges): whatever the geometry, at most
    tiling.EXPLICIT_MIN_SHARE of a compiled set's chunks resolve their slots by walking the tile tables."""
    from grad_june_amd import tiling as TL

    rng = np.random.default_rng(sv_max)
    A, V, E, sa = 6000, 5000, 20000, 64
    agent, venue = rng.integers(0, A, E), (rng.zipf(1.5, E) % V)
    t = build_tiled("x", agent, venue, V, np.ones(V, np.float32), -(-A // sa), sa, sv_max=sv_max, eb_target=eb)
    # round 4: NO chunk walks - what a descriptor cannot express has a row of explicit slots (multi_slots)
    assert TL.walk_share(t) == 0.0
    if t.slot_idx is None:
        forced = build_tiled("x", agent, venue, V, np.ones(V, np.float32), -(-A // sa), sa, sv_max=sv_max, eb_target=eb,
                             explicit=False)
        assert TL.walk_share(forced) == 0.0
        # the descriptors themselves (round 3's fence) still express all but EXPLICIT_MIN_SHARE of the chunks
        bare = build_tiled("x", agent, venue, V, np.ones(V, np.float32), -(-A // sa), sa, sv_max=sv_max, eb_target=eb,
                           multi_rows=False)
        assert bare.multi_slots is None and TL.walk_share(bare) <= TL.EXPLICIT_MIN_SHARE
