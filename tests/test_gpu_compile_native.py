"""The library's graph-compile kernels (csrc/gj_compile.hip, C ABI gj_compile_*) against
the numpy specification of the tiled layout (grad_june_amd/tiling.py): every array bit for bit, on random edge sets
that cover the edge cases of the layout (one huge venue, tiles of a few edges, duplicated edges, halo agents, an
empty edge list, venues without edges, both chunk-descriptor formats), on the reference's own 769-agent world and
on a 2 M-agent benchmark world."""
import numpy as np
import pytest
import torch

import gj_testlib as L
from grad_june_amd import tiling as TL
from grad_june_amd.plan import _host

pytestmark = pytest.mark.gpu


def random_set(rng, A, V, E, big=None, dup=0):
    venue = rng.integers(0, V, E)
    if big:
        venue[:big] = min(3, V - 1)             # one huge venue
    agent = rng.integers(0, A, E)
    if dup:                                     # duplicated (agent, venue) pairs: the reference's format allows them
        src = rng.integers(0, E, dup)
        dst = rng.integers(0, E, dup)
        agent[dst], venue[dst] = agent[src], venue[src]
    return agent.astype(np.int64), venue.astype(np.int64)


def assert_same_tiled(got, ref, what=""):
    assert (got.n_blocks, got.n_slots, got.n_edges, got.desc_wide) == (ref.n_blocks, ref.n_slots, ref.n_edges, ref.desc_wide), what
    for k in ("blk_v0", "blk_e0", "tile_sptr", "tile_jpos", "chunk_ptr", "chunk_desc", "e_cls", "v_pcontact"):
        x, y = getattr(got, k), getattr(ref, k)
        assert (x is None) == (y is None), (what, k)
        if x is not None:
            assert x.device.type == "cuda", (what, k)
            assert np.array_equal(_host(x).reshape(np.asarray(y).shape), y), (what, k)
    for k in ("e_lv", "a_la"):
        assert np.array_equal(_host(getattr(got, k)).view(np.uint16), getattr(ref, k)), (what, k)
    assert (got.slot_idx is None) == (ref.slot_idx is None), what
    if ref.slot_idx is not None:
        assert np.array_equal(_host(got.slot_idx), ref.slot_idx), (what, "slot_idx")
    assert (got.multi_slots is None) == (ref.multi_slots is None), (what, "multi_slots")
    if ref.multi_slots is not None:
        assert np.array_equal(_host(got.multi_slots), ref.multi_slots), (what, "multi_slots")


CASES = [  # A, V, E, sa, svmax, eb, wide, dup
    (5000, 700, 20000, 512, 128, 3000, None, 0),
    (300, 5, 4000, 64, 65535, 1 << 30, None, 50),
    (2000, 3000, 6000, 64, 16, 16, True, 0),
    (3000, 600, 30000, 128, 32, 600, None, 300),
    (64, 1, 10, 64, 16, 16, False, 0),
    (10000, 9000, 15000, 1024, 4096, 2000, True, 0),
    (70000, 40000, 200000, 19840, 8192, 32768, None, 0),
    (1000, 5000, 300, 256, 64, 100, None, 0),          # most venues without an edge
]


@pytest.mark.parametrize("A,V,E,sa,svmax,eb,wide,dup", CASES)
def test_native_build_equals_numpy_build(device, A, V, E, sa, svmax, eb, wide, dup):
    from grad_june_amd.tiling_native import build_tiled_native

    rng = np.random.default_rng(A * 7 + V)
    agent, venue = random_set(rng, A, V, E, big=E // 3, dup=dup)
    S = -(-A // sa)
    pc = rng.random(V).astype(np.float32)
    cls = rng.integers(0, 200, A).astype(np.uint8)
    for use_cls in (cls, None):
        ref = TL.build_tiled("x", agent, venue, V, pc, S, sa, agent_class=use_cls, sv_max=svmax, eb_target=eb, wide=wide)
        got = build_tiled_native("x", torch.from_numpy(agent).to(device), torch.from_numpy(venue).to(device), V, pc, S, sa,
                                 agent_class=use_cls, sv_max=svmax, eb_target=eb, wide=wide, device=device,
                                 n_ext_agents=A)
        assert_same_tiled(got, ref, f"cls={use_cls is not None}")


def test_native_build_edge_cases(device):
    from grad_june_amd.tiling_native import build_tiled_native

    z = torch.zeros(0, dtype=torch.int64, device=device)
    e = build_tiled_native("e", z, z, 0, np.zeros(0, np.float32), 4, 64, device=device)        # no venues at all
    assert e.n_blocks == 0 and e.n_edges == 0
    pc = np.ones(10, np.float32)
    ref = TL.build_tiled("v", np.zeros(0, np.int64), np.zeros(0, np.int64), 10, pc, 3, 64)      # venues, no edges
    got = build_tiled_native("v", z, z, 10, pc, 3, 64, device=device)
    assert_same_tiled(got, ref, "no edges")
    # the index range checks of compile_edge_set, made by the kernels
    a = torch.tensor([0, 5, 64 * 3], dtype=torch.int64, device=device)
    v = torch.tensor([0, 1, 2], dtype=torch.int64, device=device)
    with pytest.raises(ValueError, match="agent index out of range"):
        build_tiled_native("bad", a, v, 10, pc, 3, 64, device=device)
    with pytest.raises(ValueError, match="agent index out of range"):
        build_tiled_native("bad", torch.tensor([0, 100], device=device), v[:2], 10, pc, 3, 64, device=device, n_ext_agents=100)
    with pytest.raises(ValueError, match="venue index out of range"):
        build_tiled_native("bad", torch.tensor([0, 1, 2], device=device), torch.tensor([0, 10, 2], device=device), 10, pc, 3,
                           64, device=device)
    with pytest.raises(ValueError, match="agent index out of range"):
        build_tiled_native("bad", torch.tensor([0, -1, 2], device=device), v, 10, pc, 3, 64, device=device)


@pytest.mark.parametrize("A,V,E,sa,owned", [(5000, 700, 9000, 512, 5000), (300, 5, 400, 64, 300), (1000, 65534, 1500, 128, 700),
                                            (64, 1, 64, 64, 64), (3000, 50, 200, 256, 3000)])
def test_native_ell_equals_numpy_build(device, A, V, E, sa, owned):
    from grad_june_amd.tiling_native import EllBuilder

    rng = np.random.default_rng(A + E)
    agent = np.concatenate([np.arange(min(A, E)), rng.integers(0, A, max(0, E - A))]).astype(np.int64)
    rng.shuffle(agent)
    venue = rng.integers(0, V, len(agent)).astype(np.int64)
    S_owned = -(-owned // sa)
    ell3, K = TL.build_ell(agent, venue, owned, S_owned, sa)
    deg = np.bincount(agent[agent < owned], minlength=owned)
    b = EllBuilder(torch.from_numpy(agent).to(device), torch.from_numpy(venue).to(device), V, owned, sa, device)
    assert b.degrees() == (int((agent < owned).sum()), int(deg.max()))
    got, Kd = b.build(int(deg.max()))
    assert Kd == K and got.device.type == "cuda"
    assert np.array_equal(got.cpu().numpy().view(np.uint16), ell3)


def test_native_compile_of_the_reference_world(device):
    """The reference's own 769-agent world (tests/golden/june769.npz: the edge lists its loader pickled), every edge
    set, in the geometry the plan compiler picks for it."""
    from grad_june_amd.plan import compile_plan

    npz = L.load_npz("june769.npz")
    world = L.world_from(npz)
    sets = {k: {"agent": np.asarray(v["agent"]), "venue": np.asarray(v["venue"]), "people": np.asarray(v["people"])}
            for k, v in world["edge_sets"].items()}
    host = compile_plan(world["n_agents"], sets, age=world["age"], sex=world["sex"], layout="tiled")
    dev = compile_plan(world["n_agents"], sets, age=world["age"], sex=world["sex"], layout="tiled", device=device)
    assert np.array_equal(dev.work, host.work)
    for a, b in zip(dev.sets, host.sets):
        assert_same_tiled(a.tiled, b.tiled, a.name)
        assert a.tiled.ell_k == b.tiled.ell_k
        if b.tiled.ell is not None:
            assert np.array_equal(_host(a.tiled.ell).view(np.uint16), b.tiled.ell), a.name


def test_native_compile_benchmark_world(device):
    """2 M agents of the C3 benchmark world: six sets of 3 M edges, ~100 slices, the leisure set with classes."""
    from grad_june_amd.plan import compile_plan
    from grad_june_amd.synthetic import make_world

    world = make_world("c3", n_agents=2_000_000, seed=3, infected_fraction=0.01)
    host = compile_plan(world["n_agents"], world["edge_sets"], age=world["age"], sex=world["sex"], layout="tiled")
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    dev = compile_plan(world["n_agents"], world["edge_sets"], age=world["age"], sex=world["sex"], layout="tiled", device=device)
    torch.cuda.synchronize()
    print(f"native compile of 18 M edges incl. upload: {time.perf_counter() - t0:.2f} s")
    assert np.array_equal(dev.work, host.work)
    for a, b in zip(dev.sets, host.sets):
        assert_same_tiled(a.tiled, b.tiled, a.name)
        assert a.tiled.ell_k == b.tiled.ell_k
        if b.tiled.ell is not None:
            assert np.array_equal(_host(a.tiled.ell).view(np.uint16), b.tiled.ell), a.name


def test_device_compile_equals_host_compile(device):
    """compile_plan(device=...) yields the plan compile_plan builds with numpy (custom geometry, three sets)."""
    from grad_june_amd.plan import compile_plan

    rng = np.random.default_rng(5)
    A = 3000
    sets = {}
    for name, V, E in (("household", 1200, 4500), ("school", 9, 4000), ("leisure", 40, 5000)):
        venue = rng.integers(0, V, E)
        sets[name] = {"agent": rng.integers(0, A, E), "venue": venue, "people": np.bincount(venue, minlength=V)}
    age, sex = rng.integers(0, 100, A), rng.integers(0, 2, A)
    kw = dict(age=age, sex=sex, layout="tiled", sv_max=256, eb_target=1024, slices=(-(-A // 128), 128))
    ref = compile_plan(A, sets, **kw)
    got = compile_plan(A, {k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in sets.items()},
                       device=device, **kw)
    assert np.array_equal(got.work, ref.work) and got.n_slices == ref.n_slices
    for a, b in zip(got.sets, ref.sets):
        assert (a.name, a.n_venues, a.n_edges) == (b.name, b.n_venues, b.n_edges)
        assert np.array_equal(a.v_pcontact, b.v_pcontact)
        assert_same_tiled(a.tiled, b.tiled, a.name)
