"""CPU: graph compile (COO -> CSR) and the pass-1 workgroup schedule."""
import numpy as np
import pytest
import torch

import gj_testlib as L
from grad_june_amd import _native as N
from grad_june_amd.plan import build_schedule, compile_edge_set, compile_plan, p_contact


def emulate_schedule(v_rowptr, v_agent, blocks, long_rows, x, n_slots):
    """What the kernels compute with a schedule: every venue summed exactly once."""
    V = len(v_rowptr) - 1
    cum = np.full(V, np.nan)
    partial = np.zeros(max(1, n_slots))
    for (s, kind, v0, v1, e0, e1, slot, lanes) in blocks:
        if kind == 0:
            assert e1 - e0 <= N.GJ_STREAM_EDGES and lanes in (1, 4, 16, 64)
            assert e0 == v_rowptr[v0] and e1 == v_rowptr[v1]
            for v in range(v0, v1):
                assert np.isnan(cum[v])
                cum[v] = x[v_agent[v_rowptr[v]:v_rowptr[v + 1]]].sum()
        else:
            assert v1 == v0 + 1 and v_rowptr[v0] <= e0 < e1 <= v_rowptr[v0 + 1]
            partial[slot] = x[v_agent[e0:e1]].sum()
    for (s, v, s0, s1) in long_rows:
        assert np.isnan(cum[v])
        cum[v] = partial[s0:s1].sum()
    return cum


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_csr_and_schedule_cover_every_venue_once(seed):
    rng = np.random.default_rng(seed)
    A, V = 5000, 300
    deg = rng.integers(0, 40, V)
    deg[5] = 0
    deg[17] = 5000          # long venue (> 2048)
    deg[18] = 20000         # several LONG chunks
    deg[299] = 2049
    venue = np.repeat(np.arange(V), deg)
    agent = rng.integers(0, A, len(venue))
    perm = rng.permutation(len(venue))
    es = compile_edge_set("x", agent[perm], venue[perm], deg, A)
    assert es.v_rowptr[-1] == len(venue) and es.a_rowptr[-1] == len(venue)
    assert np.array_equal(np.diff(es.v_rowptr), deg)
    # stable: inside a venue row agents keep COO order
    for v in (0, 17, 100):
        sel = venue[perm] == v
        assert np.array_equal(es.v_agent[es.v_rowptr[v]:es.v_rowptr[v + 1]], agent[perm][sel])
    for a in (0, 1, 4999):
        sel = agent[perm] == a
        assert np.array_equal(es.a_venue[es.a_rowptr[a]:es.a_rowptr[a + 1]], venue[perm][sel])
    blocks, long_rows, n_slots = build_schedule(es.v_rowptr, 0)
    x = rng.random(A)
    cum = emulate_schedule(es.v_rowptr, es.v_agent, blocks, long_rows, x, n_slots)
    ref = np.bincount(venue, weights=x[agent], minlength=V)
    assert not np.isnan(cum).any()
    assert np.allclose(cum, ref)
    assert {17, 18, 299} == set(long_rows[:, 1].tolist())


def test_empty_and_ragged_inputs():
    es = compile_edge_set("e", np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0), 10)
    assert es.n_edges == 0 and es.n_venues == 0 and len(es.a_rowptr) == 11
    b, lr, ns = build_schedule(es.v_rowptr, 0)
    assert len(b) == 0 and len(lr) == 0 and ns == 0
    # venues but no edges: all covered by stream blocks of empty rows
    es = compile_edge_set("e", np.zeros(0, np.int64), np.zeros(0, np.int64), np.ones(5000), 10)
    b, lr, ns = build_schedule(es.v_rowptr, 0)
    assert b[:, 3].max() == 5000 and (b[:, 5] - b[:, 4]).sum() == 0
    with pytest.raises(ValueError):
        compile_edge_set("bad", np.array([11]), np.array([0]), np.ones(1), 10)
    with pytest.raises(ValueError):
        compile_edge_set("bad", np.array([1]), np.array([3]), np.ones(1), 10)


def test_p_contact_matches_reference_formula():
    people = torch.tensor([0, 1, 2, 3, 26, 1000])
    ref = torch.maximum(torch.minimum(1.0 / (people - 1), torch.tensor(1.0)), torch.tensor(0.0)).numpy()
    assert np.array_equal(p_contact(people), ref)
    assert np.array_equal(p_contact(25 * torch.ones(4)), np.full(4, np.float32(1) / np.float32(24)))


def test_plan_of_reference_world():
    """world769 (the reference's data.pkl): counts from SURVEY section 4 / test_june_world_loader.py."""
    npz = L.load_npz("world769.npz")
    w = L.world_from(npz, prefix="")
    es = {k: {kk: vv.numpy() for kk, vv in v.items()} for k, v in w["edge_sets"].items()}
    plan = compile_plan(w["n_agents"], es, age=w["age"].numpy(), sex=w["sex"].numpy())
    assert plan.n_agents == 769
    got = {s.name: (s.n_venues, s.n_edges) for s in plan.sets}
    assert got == {"household": (355, 745), "company": (1980, 333), "school": (1, 78), "university": (39, 43),
                   "care_home": (1, 27), "leisure": (3, 769)} or got["household"] == (355, 745)
    assert plan.n_edges == 1995
    assert plan.agent_class.max() < 200
    # every venue of every set is covered exactly once
    for sid, s in enumerate(plan.sets):
        b = plan.blocks[plan.blocks[:, 0] == sid]
        covered = np.zeros(s.n_venues, dtype=int)
        for row in b[b[:, 1] == 0]:
            covered[row[2]:row[3]] += 1
        for row in plan.long_rows[plan.long_rows[:, 0] == sid]:
            covered[row[1]] += 1
        assert (covered == 1).all()


def test_plan_save_load_roundtrip(tmp_path):
    from grad_june_amd.plan import load_plan, save_plan

    npz = L.load_npz("world769.npz")
    w = L.world_from(npz, prefix="")
    es = {k: {kk: vv.numpy() for kk, vv in v.items()} for k, v in w["edge_sets"].items()}
    for layout in ("csr", "tiled", "both"):
        plan = compile_plan(w["n_agents"], es, age=w["age"].numpy(), sex=w["sex"].numpy(), layout=layout)
        save_plan(plan, tmp_path / f"{layout}.npz")
        back = load_plan(tmp_path / f"{layout}.npz")
        assert (back.n_agents, back.layout, back.n_slices, back.slice_agents) == (plan.n_agents, layout, plan.n_slices, plan.slice_agents)
        assert np.array_equal(back.agent_class, plan.agent_class) and np.array_equal(back.blocks, plan.blocks)
        for a, b in zip(plan.sets, back.sets):
            assert a.name == b.name and a.n_edges == b.n_edges
            for k in ("v_rowptr", "v_agent", "a_rowptr", "a_venue", "v_pcontact"):
                x, y = getattr(a, k), getattr(b, k)
                assert (x is None and y is None) or np.array_equal(x, y), (layout, k)
            if a.tiled is not None:
                for k in ("blk_v0", "blk_e0", "e_lv", "e_cls", "a_la", "tile_sptr", "tile_jpos", "chunk_ptr", "chunk_desc"):
                    x, y = getattr(a.tiled, k), getattr(b.tiled, k)
                    assert (x is None and y is None) or np.array_equal(x, y), (layout, k)


def test_device_compile_needs_a_hip_device():
    """compile_plan(device=...) runs the library's compile kernels: a CPU device is refused, nothing falls back."""
    import torch

    rng = np.random.default_rng(5)
    venue = rng.integers(0, 9, 400)
    sets = {"school": {"agent": rng.integers(0, 300, 400), "venue": venue, "people": np.bincount(venue, minlength=9)}}
    with pytest.raises(RuntimeError, match="HIP device"):
        compile_plan(300, sets, layout="tiled", device=torch.device("cpu"))