"""Tiled ("propagation-blocked") layout of an edge set - the fast path of the two passes.

Why: in CSR form every edge costs a random 4-byte gather (``transmission[agent]`` in pass 1,
``cum[venue]`` in pass 2).  On MI355X such a gather pulls a whole 128-byte line from the Infinity
Cache or L2 - measured 60 G gathers/s from a 40 MB table, 175 G/s when the table is L2-resident -
which bounds the CSR kernels at 6-12 % of the HBM roofline.  LDS serves random 4-byte reads two
orders of magnitude faster, so the tiled layout arranges for every random access to hit LDS:

* agents are cut into ``S`` *slices* of ``SA`` consecutive agents (a slice's values fit in LDS),
* the venues of a set are cut into *blocks* of consecutive venues (a block's sums fit in LDS),
* edge (a, v) belongs to *tile* (slice(a), block(v)).  Edges are stored twice, as 16-bit local
  indices: ``a_la`` in slice-major tile order (s, j) and ``e_lv`` in block-major tile order (j, s);
  a tile is contiguous in both, and inside a tile both use the same order (by venue, then agent).

Per step and set, four streaming phases move 24 bytes per edge through HBM, all coalesced:
  A  (one workgroup per slice)   x[slice] -> LDS;  val[blockmajor pos] = x[a_la]
  B  (one workgroup per block)   sums[e_lv] += val   (LDS 64-bit fixed-point atomics), cum = beta*p_contact*sums
  C  (same workgroup)            val[i] = cum[e_lv[i]]                (in place)
  D  (one workgroup per slice)   acc[a_la] += val[blockmajor pos]     (LDS fixed-point atomics), epilogue
This module builds the static arrays (numpy, host) and is exercised on the CPU by an emulation of
the four phases (tests/test_tiling_host.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

SA_MAX = 19840        # agents per slice: 64-bit fixed-point sums + two flag bits per agent in 160 KiB of LDS (phase D)
MAX_AGENT_EDGES = 4096  # per owned agent, all sets: 4 096 terms of up to 262 144 stay inside phase D's 64-bit sums
SV_MAX = 16384        # venues per block: 128 KiB of 64-bit sums in phase B, one block per CU.  (Rounds 1-2: 8192, so that
                      # two blocks shared a CU's LDS - worth 25 % then; with the direct form of pass 2 and the exact run
                      # merging the launch runs one workgroup per CU and larger blocks win: round 3, tools/ab.py on C3,
                      # 8192 / 65536 edges per block 538 us per step, 16384 / 131072 528.  Local venue index is 16-bit.)
EB_TARGET = 131072    # edges per block aimed for (work per workgroup of phases B/C)
N_CU = 256
PAD = 8               # block-major arrays: every block starts on a multiple of PAD slots
CHUNK = 64            # slice-major traversal granule (one wave-instruction)


def choose_slices(n_agents: int, sa_max: int = SA_MAX, n_cu: int = N_CU):
    """(S, SA): S a multiple of the CU count once the world is large, SA <= sa_max."""
    if n_agents <= 0:
        return 1, 64
    if n_agents <= n_cu * 1024:
        sa = 1024 if n_agents > 1024 * 8 else max(64, -(-n_agents // 8))
        sa = -(-sa // 64) * 64
        s = -(-n_agents // sa)
        return s, sa
    m = -(-n_agents // (n_cu * sa_max))
    s = n_cu * m
    sa = -(-n_agents // s)
    sa = -(-sa // 64) * 64
    s = -(-n_agents // sa)
    return s, sa


def choose_block_edges(n_slices: int, eb_max: int = EB_TARGET) -> int:
    """Edges per venue block.  A tile holds about eb / S edges: below ~128 the slice-major phases (A, D)
    slow down (chunks spanning several tiles), while small blocks give phases B/C more workgroups -
    what a small world (one rank's share of a strong-scaled run) needs to fill 256 CUs.  Measured on
    MI355X (gpurun_out sweeps, c3 preset): 1.25M agents 0.190 -> 0.131 ms/step with 32768 instead of
    131072; 10M agents: 131072 (with blocks of up to 16384 venues) 2 % faster than 65536 in round 3."""
    return int(min(eb_max, max(32768, 256 * n_slices)))


def venue_blocks(degree: np.ndarray, sv_max: int = SV_MAX, eb_target: int = EB_TARGET) -> np.ndarray:
    """Block boundaries ``blk_v0`` [J+1] over consecutive venues: <= sv_max venues, about eb_target
    edges (a venue above eb_target is a block of its own)."""
    V = len(degree)
    if V == 0:
        return np.zeros(1, dtype=np.int64)
    rp = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(degree, out=rp[1:])
    bounds = [0]
    v = 0
    while v < V:
        end = int(np.searchsorted(rp, rp[v] + eb_target, side="right")) - 1
        end = max(end, v + 1)
        end = min(end, V, v + sv_max)
        bounds.append(end)
        v = end
    return np.asarray(bounds, dtype=np.int64)


@dataclass
class TiledEdgeSet:
    name: str
    n_venues: int
    n_edges: int
    n_slices: int
    n_blocks: int
    blk_v0: np.ndarray      # int32 [J+1]  venue range of block j
    blk_e0: np.ndarray      # int32 [J+1]  block-major edge range of block j
    e_lv: np.ndarray        # uint16 [E]   local venue index, block-major tile order (j, s)
    e_cls: Optional[np.ndarray]  # uint8 [E] agent class of the edge's agent, block-major (leisure sets)
    a_la: np.ndarray        # uint16 [E]   local agent index, slice-major tile order (s, j)
    tile_sptr: np.ndarray   # int32 [S*J+1] slice-major prefix: tile (s, j) = [sptr[s*J+j], sptr[s*J+j+1])
    tile_jpos: np.ndarray   # int32 [S*J]   block-major start of tile (s, j)
    v_pcontact: np.ndarray  # float32 [V]
    n_slots: int = 0        # length of the block-major arrays: every block padded to a multiple of 8
    chunk_ptr: Optional[np.ndarray] = None   # int32 [S+1] first 64-edge chunk of slice s (slice-major)
    chunk_desc: Optional[np.ndarray] = None  # int32 [n_chunks, 4]: slot0, slot1, split | multi << 16, j0
    desc_wide: bool = False                  # chunk_desc is int32 [n_chunks, 8], see wide_descriptors()
    slot_idx: Optional[np.ndarray] = None    # int32 [E]: the block-major slot of every slice-major edge - replaces the
                                             # descriptors of a set with tiles of a few edges (EXPLICIT_MIN_SHARE)
    multi_slots: Optional[np.ndarray] = None  # int32 [n_multi, 64]: the slots of the 64 edges of every chunk its descriptor
                                              # cannot express ("multi"); the descriptor's j0 field then holds the row
    ell: Optional[np.ndarray] = None         # uint16 [owned agents padded to slices, ell_k]: "direct" pass 2, see build_ell()
    ell_k: int = 0                           # 0: pass 2 runs through phases C + D like pass 1
    runs: Optional["RunForm"] = None         # the set's primary edges in the run form (see split_primary_runs)
    presum: bool = False                     # pass 1 in the direct form too (ELL rows + per-workgroup LDS tables): a set
                                             # in the direct form whose edges all belong to owned agents
    # A chunk = 64 consecutive slice-major edges.  Its first `split` edges lie in one tile and map to
    # block-major slots slot0, slot0+1, ...; the rest lie in the next non-empty tile and map to slot1,
    # slot1+1, ...  `multi` flags the rare chunk that spans more than two tiles (tiny tiles): its
    # lanes take their slots from row j0 of `multi_slots` (round 4; before: they walked the tile tables from block j0 -
    # dependent loads, ~50x a normal chunk: a world with a geography keeps most of a slice's edges in a few long tiles
    # and scatters the rest - 1 % of the school edges - over every block, and that 1 % doubled phase A).


WIDE_SEGMENTS = 6      # tiles a 64-edge chunk may span in the wide descriptor format
WIDE_MIN_SHARE = 0.05  # a set whose chunks span > 2 tiles more often than this gets wide descriptors (round 4: 0.01 until such
                       # chunks had rows of explicit slots; clustered C3: 513 -> 508 us per step, profiles/r04_ab_wide_threshold_clustered.txt)
EXPLICIT_MIN_SHARE = 0.02   # a set whose chunks span > 6 tiles (every lane walks the tile tables: a 5x cliff in phases A
                            # and D) more often than this carries the slot of every edge explicitly instead (slot_idx)


def wide_descriptors(sptr: np.ndarray, jpos_flat: np.ndarray, seg: np.ndarray, chunk_ptr: np.ndarray,
                     first_edge: np.ndarray, chunk_end: np.ndarray, t0: np.ndarray, J: int):
    """Descriptors for sets with small tiles: int32 [n_chunks, 8] per 64-edge chunk of the slice-major order,
        base_0 .. base_5,  start_1 | start_2 << 8 | start_3 << 16 | start_4 << 24,  start_5 | multi << 8 | j0 << 9
    Segment k = the chunk's lanes [start_k, start_k+1) (start_0 = 0, unused segments start at 64) lies in one
    tile and maps to block-major slots base_k + lane.  ``multi``: more than 6 tiles - the lanes walk the tile
    tables from block j0.  Returns (desc, segments per chunk)."""
    n_chunks = len(first_edge)
    E = int(sptr[-1])
    tile_len = np.diff(sptr)
    starts_all = sptr[:-1][tile_len > 0]                     # slice-major position where each non-empty tile begins
    tile_ids = np.flatnonzero(tile_len > 0)
    # tile beginnings strictly inside a chunk are its segment boundaries
    lo = np.searchsorted(starts_all, first_edge, side="right")      # first boundary > first_edge
    hi = np.searchsorted(starts_all, chunk_end, side="left")        # boundaries < chunk_end
    nseg = 1 + (hi - lo)
    desc = np.zeros((n_chunks, 8), dtype=np.int64)
    desc[:, 0] = jpos_flat[t0] + (first_edge - sptr[t0])
    starts = np.full((n_chunks, WIDE_SEGMENTS - 1), 64, dtype=np.int64)
    for k in range(1, WIDE_SEGMENTS):
        has = nseg > k
        b = np.minimum(lo + (k - 1), len(starts_all) - 1) if len(starts_all) else np.zeros(n_chunks, np.int64)
        pos = starts_all[b] if len(starts_all) else np.zeros(n_chunks, np.int64)
        st = np.where(has, pos - first_edge, 64)
        base = np.where(has, jpos_flat[tile_ids[b]] - st, 0) if len(starts_all) else np.zeros(n_chunks, np.int64)
        desc[:, k] = base
        starts[:, k - 1] = st
    multi = nseg > WIDE_SEGMENTS
    j0 = t0 - np.repeat(np.arange(len(chunk_ptr) - 1) * J, np.diff(chunk_ptr))
    desc[:, 6] = starts[:, 0] | (starts[:, 1] << 8) | (starts[:, 2] << 16) | (starts[:, 3] << 24)
    desc[:, 7] = starts[:, 4] | (multi.astype(np.int64) << 8) | (j0 << 9)
    if len(j0) and int(j0.max()) >= (1 << 22):
        raise ValueError("too many venue blocks for the wide descriptor's j0 field")
    return desc.astype(np.uint32).view(np.int32), nseg


def multi_flags(chunk_desc, wide: bool) -> np.ndarray:
    """bool [n_chunks]: the chunks whose descriptor cannot express them (more tiles than it has segments)."""
    d = np.asarray(chunk_desc.cpu() if hasattr(chunk_desc, "cpu") else chunk_desc).reshape(-1, 8 if wide else 4)
    return (((d[:, 7].view(np.uint32) >> 8) & 1) if wide else (d[:, 2].view(np.uint32) >> 16)) != 0


def attach_multi_slots(chunk_desc: np.ndarray, wide: bool, first_edge: np.ndarray, chunk_end: np.ndarray,
                       sptr: np.ndarray, jpos_flat: np.ndarray):
    """The slots of the multi chunks, one row of 64 per chunk in chunk order (lanes past the chunk's end: 0), and the
    descriptors with the row number in their j0 field.  Returns (chunk_desc, multi_slots or None)."""
    flags = multi_flags(chunk_desc, wide)
    rows = np.flatnonzero(flags)
    if len(rows) == 0:
        return chunk_desc, None
    pos = first_edge[rows][:, None] + np.arange(CHUNK, dtype=np.int64)[None, :]
    ok = pos < chunk_end[rows][:, None]
    pos = np.where(ok, pos, first_edge[rows][:, None])
    t_of = np.searchsorted(sptr, pos.reshape(-1), side="right") - 1
    slots = (jpos_flat[t_of] + (pos.reshape(-1) - sptr[t_of])).reshape(-1, CHUNK)
    slots = np.where(ok, slots, 0).astype(np.int32)
    d = chunk_desc.copy()
    m = np.arange(len(rows), dtype=np.int64)
    if wide:
        w7 = d[rows, 7].view(np.uint32).astype(np.int64)
        d[rows, 7] = ((w7 & 0x1FF) | (m << 9)).astype(np.uint32).view(np.int32)
        if len(m) >= (1 << 22):
            raise ValueError("too many multi chunks for the wide descriptor's row field")
    else:
        d[rows, 3] = m.astype(np.int32)
    return d, slots


def walk_share(t: "TiledEdgeSet") -> float:
    """Share of a set's 64-edge chunks whose lanes have to walk the tile tables in phases A and D (more tiles in the
    chunk than its descriptor expresses, and no row of explicit slots for it).  0 for a set with explicit slots or with
    ``multi_slots`` - i.e. for everything build_tiled / the native compile produce since round 4."""
    if t.slot_idx is not None or t.multi_slots is not None or t.chunk_desc is None or len(t.chunk_desc) == 0:
        return 0.0
    d = np.asarray(t.chunk_desc.cpu() if hasattr(t.chunk_desc, "cpu") else t.chunk_desc).reshape(-1, 8 if t.desc_wide else 4)
    flag = ((d[:, 7].view(np.uint32) >> 8) & 1) if t.desc_wide else (d[:, 2].view(np.uint32) >> 16)
    return float((flag != 0).mean())


def build_tiled(name: str, agent_index, venue_index, n_venues: int, v_pcontact: np.ndarray,
                n_slices: int, slice_agents: int, agent_class: Optional[np.ndarray] = None,
                sv_max: int = SV_MAX, eb_target: int = EB_TARGET, wide: Optional[bool] = None,
                explicit: Optional[bool] = None, tile_pad: int = 1, multi_rows: bool = True) -> TiledEdgeSet:
    """``tile_pad`` (experiment, host compile only - DESIGN.md section 8, "the next lever"): every tile's run is padded
    to a multiple of ``tile_pad`` positions in BOTH orders, so that every piece of a 64-edge chunk starts and ends on
    a 64-byte boundary of ``val`` (tile_pad = 16).  Pad positions: local agent 0 in ``a_la``, local venue 0xFFFF in
    ``e_lv`` - phase B skips them, phase C writes 0 into them, phase D adds that 0 to agent 0 of the slice.  Sets
    whose tiles hold fewer than ``TILE_PAD_MIN_MEAN`` edges on average are left unpadded."""
    agent = np.asarray(agent_index, dtype=np.int64).ravel()
    venue = np.asarray(venue_index, dtype=np.int64).ravel()
    E = len(agent)
    if slice_agents > 65536 or sv_max > 65535:
        raise ValueError("local indices are 16-bit (and local venue 0xFFFF marks a pad slot)")
    degree = np.bincount(venue, minlength=n_venues) if E else np.zeros(n_venues, dtype=np.int64)
    blk_v0 = venue_blocks(degree, sv_max, eb_target)
    J = len(blk_v0) - 1
    S = n_slices
    if J == 0:
        z32 = np.zeros(1, dtype=np.int32)
        return TiledEdgeSet(name, n_venues, 0, S, 0, z32, z32, np.zeros(0, np.uint16), None,
                            np.zeros(0, np.uint16), np.zeros(1, np.int32), np.zeros(0, np.int32),
                            np.asarray(v_pcontact, dtype=np.float32), 0, np.zeros(S + 1, np.int32),
                            np.zeros((0, 4), np.int32))
    vblk = (np.searchsorted(blk_v0, np.arange(n_venues), side="right") - 1).astype(np.int64)
    j = vblk[venue]
    lv = venue - blk_v0[j]
    s = agent // slice_agents
    la = agent - s * slice_agents
    if E and (s.max() >= S):
        raise ValueError("agent index beyond the last slice")
    # block-major tile order: (j, s, lv, la); ties (duplicate edges) keep COO order
    key = ((j * S + s) * 65536 + lv) * 65536 + la
    order = np.argsort(key, kind="stable")
    tile_of = (j * S + s)[order]                       # block-major tile id per (unpadded) position
    real_len_js = np.bincount(tile_of, minlength=J * S).reshape(J, S)
    rpos_js = np.zeros(J * S + 1, dtype=np.int64)      # prefix of the real tile lengths: position in the sorted order
    np.cumsum(real_len_js.reshape(-1), out=rpos_js[1:])
    if tile_pad > 1 and E < TILE_PAD_MIN_MEAN * max(1, int((real_len_js > 0).sum())):
        tile_pad = 1                                   # tiles of a few edges: padding would multiply the set
    tile_len_js = -(-real_len_js // tile_pad) * tile_pad     # a tile's run in both orders (== real_len_js unpadded)
    upos_js = np.zeros(J * S + 1, dtype=np.int64)      # block-major prefix of the tiles' runs, blocks unpadded
    np.cumsum(tile_len_js.reshape(-1), out=upos_js[1:])
    # every block occupies a multiple of PAD slots so that phases B/C can use 16-byte accesses;
    # the pad slots at a block's end carry the sentinel local venue 0xFFFF
    blk_len = tile_len_js.sum(1)
    blk_slots = -(-blk_len // PAD) * PAD
    blk_start = np.zeros(J + 1, dtype=np.int64)
    np.cumsum(blk_slots, out=blk_start[1:])
    blk_ustart = np.concatenate([upos_js[0:J * S:S], [upos_js[-1]]])
    shift = np.repeat(blk_start[:-1] - blk_ustart[:-1], S)          # per tile (j, s): padded - unpadded
    jpos_js = np.concatenate([upos_js[:-1] + shift, [blk_start[-1]]])
    n_slots = int(blk_start[-1])
    tile_len_sj = tile_len_js.T.copy()                 # [S, J]
    sptr = np.zeros(S * J + 1, dtype=np.int64)
    np.cumsum(tile_len_sj.reshape(-1), out=sptr[1:])
    jpos_sj = jpos_js[:-1].reshape(J, S).T.copy()      # block-major start of tile (s, j)
    # slice-major position of every block-major position
    tj = tile_of // S
    ts = tile_of - tj * S
    within = np.arange(E, dtype=np.int64) - rpos_js[tile_of]
    pos_sm = sptr[ts * J + tj] + within
    pos_bm = jpos_js[tile_of] + within                              # padded block-major slot
    la_bm = la[order]
    E_real, E = E, int(sptr[-1])                       # from here on E counts slice-major POSITIONS (pads included)
    a_la = np.zeros(E, dtype=np.uint16)                # (pad positions: local agent 0)
    a_la[pos_sm] = la_bm.astype(np.uint16)
    e_lv = np.full(n_slots, 0xFFFF, dtype=np.uint16)
    e_lv[pos_bm] = lv[order].astype(np.uint16)
    e_cls = None
    if agent_class is not None:
        e_cls = np.zeros(n_slots, dtype=np.uint8)
        e_cls[pos_bm] = np.asarray(agent_class, dtype=np.uint8)[agent[order]]
    # 64-edge chunks of each slice's slice-major segment, and the block of each chunk's first edge
    seg = sptr[0:S * J + 1:J]                                        # [S+1] slice segment bounds
    n_chunks = -(-(np.diff(seg)) // CHUNK)
    chunk_ptr = np.zeros(S + 1, dtype=np.int64)
    np.cumsum(n_chunks, out=chunk_ptr[1:])
    first_edge = np.repeat(seg[:-1], n_chunks) + CHUNK * (np.arange(chunk_ptr[-1]) - np.repeat(chunk_ptr[:-1], n_chunks))
    # tile (slice-major index) containing an edge position: last tile whose start <= position, skipping empties
    seg_end = np.repeat(seg[1:], n_chunks)
    chunk_end = np.minimum(first_edge + CHUNK, seg_end)
    t0 = np.searchsorted(sptr, first_edge, side="right") - 1
    end0 = sptr[t0 + 1]
    split = np.minimum(end0, chunk_end) - first_edge
    slot0 = jpos_sj.reshape(-1)[t0] + (first_edge - sptr[t0])
    two = end0 < chunk_end                                             # chunk continues in another tile
    t1 = np.where(two, np.searchsorted(sptr, np.minimum(end0, E - 1), side="right") - 1, t0)
    slot1 = np.where(two, jpos_sj.reshape(-1)[t1] + (end0 - sptr[t1]), 0)
    multi = two & (sptr[t1 + 1] < chunk_end)
    chunk_desc = np.stack([slot0, slot1, split + (multi.astype(np.int64) << 16),
                           t0 - np.repeat(np.arange(S) * J, n_chunks)], axis=1).astype(np.int32)
    if wide is None:
        wide = bool(len(multi)) and float(multi.mean()) > WIDE_MIN_SHARE
    slot_idx = None
    if wide:
        chunk_desc, nseg = wide_descriptors(sptr, jpos_sj.reshape(-1), seg, chunk_ptr, first_edge, chunk_end, t0, J)
        if explicit is None:
            explicit = bool(len(nseg)) and float((nseg > WIDE_SEGMENTS).mean()) > EXPLICIT_MIN_SHARE
    multi_slots = None
    if explicit and E > 0:          # (a set without edges carries no slots, in either compile)
        # (every slice-major position, pads included: position p of tile t sits at tile_jpos[t] + (p - sptr[t]))
        t_of = np.searchsorted(sptr, np.arange(E, dtype=np.int64), side="right") - 1
        slot_idx = (jpos_sj.reshape(-1)[t_of] + (np.arange(E, dtype=np.int64) - sptr[t_of])).astype(np.int32)
    elif multi_rows and not explicit:
        chunk_desc, multi_slots = attach_multi_slots(np.ascontiguousarray(chunk_desc), bool(wide), first_edge, chunk_end,
                                                     sptr, jpos_sj.reshape(-1))
    return TiledEdgeSet(
        name=name, n_venues=n_venues, n_edges=E, n_slices=S, n_blocks=J,
        blk_v0=blk_v0.astype(np.int32), blk_e0=blk_start.astype(np.int32),
        e_lv=e_lv, e_cls=e_cls, a_la=a_la,
        tile_sptr=sptr.astype(np.int32), tile_jpos=jpos_sj.reshape(-1).astype(np.int32),
        v_pcontact=np.asarray(v_pcontact, dtype=np.float32), n_slots=n_slots,
        chunk_ptr=chunk_ptr.astype(np.int32), chunk_desc=np.ascontiguousarray(chunk_desc), desc_wide=bool(wide),
        slot_idx=slot_idx, multi_slots=multi_slots)


# ------------------------------------------------------------------------------------------------
# "run form": the edge set that defines the agent order needs no index arrays for one edge per agent
# ------------------------------------------------------------------------------------------------
# Under the household-major agent order (graph.locality_order / synthetic.reorder_agents: agents sorted by their
# first - smallest - venue of that set) the agents whose first venue is v are CONSECUTIVE.  One edge per agent, its
# "primary" edge (a, vmin(a)), is then implied by the agent's position:
#   pass 1: the block-major stream of these edges IS the transmission array itself - phase B reads x[a] (coalesced)
#           next to a 2-byte block-local venue index per agent and adds it to the venue's sum like any other slot;
#           phase A has nothing to scatter for them;
#   pass 2: an agent's primary venue lies in a narrow window of consecutive venues per slice: phase D stages that
#           window of `cum` through LDS (the direct form's machinery) and reads it through a 2-byte window-relative
#           index per agent; phase C has nothing to write for them.
# Per primary edge ~10 bytes per step instead of ~25.  The remaining edges of the set (an agent's other venues, edges
# of halo agents) stay in the tiled arrays.  In a world of the reference's kind - every person lives in exactly one
# household - the household set leaves the tiled path entirely.
TILE_PAD_MIN_MEAN = 64     # build_tiled(tile_pad > 1): mean edges per non-empty tile below which a set stays unpadded
RUN_MIN_SHARE = 0.25        # primary edges / owned edges below which the run form is not worth its two extra arrays
RUN_MAX_WINDOW = 32768      # venues a slice's window may span (LDS floats of phase D's table region; ids are 16-bit)


@dataclass
class RunForm:
    n_primary: int
    keep: Optional[np.ndarray]     # bool [E]: edges that stay in the tiled arrays (numpy build only)
    vmin: np.ndarray               # int32 [owned agents] smallest venue of the agent, 0x7FFFFFFF = no edge
    pv_blk: Optional[np.ndarray] = None   # uint16 [rows] primary venue, relative to its venue block's first venue
    pv_win: Optional[np.ndarray] = None   # uint16 [rows] primary venue, relative to the slice's window; 0xFFFF = none
    blk_r0: Optional[np.ndarray] = None   # int32 [J+1] agents [blk_r0[j], blk_r0[j+1]) have their primary venue in block j
    win_lo: Optional[np.ndarray] = None   # int32 [owned slices] first venue of the slice's window
    win_n: Optional[np.ndarray] = None    # int32 [owned slices] venues in the window
    max_window: int = 0


NO_VENUE = 0x7FFFFFFF


def split_primary_runs(agent_index, venue_index, n_agents: int, n_venues: int, slice_agents: int,
                       min_share: float = RUN_MIN_SHARE) -> Optional[RunForm]:
    """Decide whether a set takes the run form and, if so, pick every owned agent's primary edge: the FIRST edge in
    COO order to the agent's smallest venue.  None when the owned agents are not ordered by their smallest venue
    (agents without an edge last), when a slice's window is too wide or when too few edges would leave."""
    agent = np.asarray(agent_index, dtype=np.int64).ravel()
    venue = np.asarray(venue_index, dtype=np.int64).ravel()
    if n_agents <= 0 or len(agent) == 0:
        return None
    own = agent < n_agents
    vmin = np.full(n_agents, NO_VENUE, dtype=np.int64)
    np.minimum.at(vmin, agent[own], venue[own])
    if (np.diff(vmin) < 0).any():
        return None
    e_idx = np.flatnonzero(own & (venue == vmin[np.minimum(agent, n_agents - 1)]))
    pick = np.full(n_agents, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(pick, agent[e_idx], e_idx)
    has = vmin != NO_VENUE
    n_primary = int(has.sum())
    if n_primary < min_share * int(own.sum()):
        return None
    n_own_slices = max(1, -(-n_agents // slice_agents))
    win_lo = np.zeros(n_own_slices, dtype=np.int64)
    win_n = np.zeros(n_own_slices, dtype=np.int64)
    for s in range(n_own_slices):          # (vectorised below would need reduceat on ragged ends; slices are few)
        seg = vmin[s * slice_agents:min(n_agents, (s + 1) * slice_agents)]
        seg = seg[seg != NO_VENUE]
        if len(seg):
            win_lo[s] = seg[0]
            win_n[s] = seg[-1] - seg[0] + 1
    if int(win_n.max()) > RUN_MAX_WINDOW:
        return None
    keep = np.ones(len(agent), dtype=bool)
    keep[pick[has]] = False
    return RunForm(n_primary=n_primary, keep=keep, vmin=vmin.astype(np.int32), win_lo=win_lo.astype(np.int32),
                   win_n=win_n.astype(np.int32), max_window=int(win_n.max()))


def finish_run_form(rf: RunForm, blk_v0: np.ndarray, n_agents: int, slice_agents: int) -> RunForm:
    """The per-agent indices once the venue blocks of the set's tiled part are known."""
    vmin = rf.vmin.astype(np.int64)
    blk_v0 = np.asarray(blk_v0, dtype=np.int64)
    n_own_slices = len(rf.win_lo)
    rows = n_own_slices * slice_agents
    has = vmin != NO_VENUE
    pv_blk = np.full(rows, 0xFFFF, dtype=np.uint16)
    pv_win = np.full(rows, 0xFFFF, dtype=np.uint16)
    j = np.searchsorted(blk_v0, vmin[has], side="right") - 1
    pv_blk[:n_agents][has] = (vmin[has] - blk_v0[j]).astype(np.uint16)
    sl = np.arange(n_agents)[has] // slice_agents
    pv_win[:n_agents][has] = (vmin[has] - rf.win_lo.astype(np.int64)[sl]).astype(np.uint16)
    # agents of block j: vmin in [blk_v0[j], blk_v0[j+1]) - vmin is sorted, agents without an edge (NO_VENUE) come last
    rf.blk_r0 = np.searchsorted(vmin, blk_v0, side="left").astype(np.int32)
    rf.pv_blk, rf.pv_win = pv_blk, pv_win
    return rf


# ------------------------------------------------------------------------------------------------
# "direct" pass 2 for sets with few venues
# ------------------------------------------------------------------------------------------------
# When a set has so few venues that its whole cum vector fits in LDS (schools, universities, leisure venues:
# 10^3-10^4 venues against 10^7 agents), pass 2 needs no per-edge workspace at all: phase C is skipped and
# phase D, after the tiled sets are accumulated, loads the set's cum into LDS and every lane adds up the
# venues of ITS agents from an ELL table (K 16-bit venue ids per agent, 0xFFFF = none).  Per edge that is
# ~2*K/deg bytes of indices instead of 12.5 (C: e_lv + val write, D: a_la + val read + descriptors), no LDS
# atomics, and the sum of an agent is taken in its COO edge order.
DIRECT_MAX_VENUES = 65534     # venue ids are 16-bit, 0xFFFF = pad
DIRECT_MAX_K = 8              # ELL columns (a power of two: one 2/4/8/16-byte load per agent)
DIRECT_TABLE_GROUPS = 4       # the venues' cum may be staged through LDS in at most this many groups
DIRECT_MAX_PAD = 4.0          # ELL entries per edge above which the tiled form is cheaper
LDS_BYTES = 160 * 1024
DIRECT_WEIGHT_FLOATS = 8 * 200   # per-class weights of up to GJ_MAX_NETS_PER_SET leisure networks


def direct_table_floats(slice_agents: int = 0) -> int:
    """Floats of LDS for a group of venue values in phase D's direct form: by then the slice's sums are in
    registers, so all of the LDS but two leisure class-weight buffers and the slack of the two staging regions."""
    return LDS_BYTES // 4 - 2 * DIRECT_WEIGHT_FLOATS - 2 * 64


def direct_columns(degree_max: int) -> int:
    """ELL columns: a power of two >= 2 (the kernel takes an agent's entries in pairs)."""
    k = 2
    while k < degree_max:
        k *= 2
    return k


def direct_eligible(n_venues: int, n_edges_owned: int, n_agents: int, degree_max: int, nets: int,
                    slice_agents: int) -> bool:
    """Does pass 2 of this set run in the direct form?  A function of the set's sizes only."""
    if n_edges_owned == 0 or n_venues > DIRECT_MAX_VENUES or degree_max > DIRECT_MAX_K:
        return False
    if n_venues * max(1, nets) > DIRECT_TABLE_GROUPS * direct_table_floats(slice_agents):
        return False
    return direct_columns(degree_max) * n_agents <= DIRECT_MAX_PAD * n_edges_owned


def ell_planes(ell_rows_by_k, K: int):
    """[rows, K] -> the device form [planes, rows, 2]: columns in pairs, one plane per pair, so that one kernel body
    (a pair of entries per agent) serves every K."""
    if K <= 2:
        return ell_rows_by_k.reshape(1, ell_rows_by_k.shape[0], K)
    rows = ell_rows_by_k.shape[0]
    return ell_rows_by_k.reshape(rows, K // 2, 2).transpose(1, 0, 2) if isinstance(ell_rows_by_k, np.ndarray) \
        else ell_rows_by_k.reshape(rows, K // 2, 2).permute(1, 0, 2)


def build_ell(agent_index, venue_index, n_agents: int, n_slices_owned: int, slice_agents: int):
    """ELL form of the OWNED agents' edges: uint16 [planes, n_slices_owned * slice_agents, min(K, 2)]; entry
    (c // 2, a, c % 2) = the venue of agent a's c-th edge in COO order, 0xFFFF = no edge.  Returns (ell, K)."""
    agent = np.asarray(agent_index, dtype=np.int64).ravel()
    venue = np.asarray(venue_index, dtype=np.int64).ravel()
    owned = agent < n_agents
    a, v = agent[owned], venue[owned]
    order = np.argsort(a, kind="stable")
    a, v = a[order], v[order]
    deg = np.bincount(a, minlength=n_agents)
    K = direct_columns(int(deg.max()) if len(deg) else 1)
    rowptr = np.zeros(n_agents + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    col = np.arange(len(a), dtype=np.int64) - rowptr[a]
    ell = np.full((n_slices_owned * slice_agents, K), 0xFFFF, dtype=np.uint16)
    ell[a, col] = v.astype(np.uint16)
    return np.ascontiguousarray(ell_planes(ell, K)), K


def ell_rows(ell) -> np.ndarray:
    """The device form [planes, rows, kp] back to [rows, K] (tests, emulation)."""
    planes, rows, kp = ell.shape
    return np.ascontiguousarray(np.asarray(ell).transpose(1, 0, 2).reshape(rows, planes * kp))


def emulate_direct_pass2(ell: np.ndarray, cum: np.ndarray, n_agents: int, weights: Optional[np.ndarray] = None,
                         agent_class: Optional[np.ndarray] = None, group_venues: Optional[int] = None):
    """Phase D's direct form: acc[a] = sum over the agent's ELL entries of cum[v] (x weights[k][class[a]] summed
    over the set's networks k when ``cum`` is [V, nk]), venue groups of ``group_venues`` in turn, fp32."""
    if ell.ndim == 3:
        ell = ell_rows(ell)
    cum = np.asarray(cum, dtype=np.float32)
    cum2 = cum.reshape(len(cum), -1)
    V, nk = cum2.shape
    gv = group_venues or max(1, V)
    acc = np.zeros(n_agents, dtype=np.float32)
    for v0 in range(0, max(V, 1), gv):
        s = np.zeros(n_agents, dtype=np.float32)              # one lane's sum over its agent's entries in this group
        for c in range(ell.shape[1]):
            lv = ell[:n_agents, c].astype(np.int64)
            ok = (lv != 0xFFFF) & (lv >= v0) & (lv < v0 + gv)
            idx = np.where(ok, lv, 0)
            if weights is None:
                term = cum2[idx, 0]
            else:
                term = np.zeros(n_agents, dtype=np.float32)
                for k in range(nk):
                    term = term + weights[k][agent_class[:n_agents]] * cum2[idx, k]
            s = s + np.where(ok, term, np.float32(0)).astype(np.float32)
        acc = acc + s
    return acc


# ------------------------------------------------------------------------------------------------
# numpy emulation of the four phases (the specification the kernels are tested against on the CPU)
# ------------------------------------------------------------------------------------------------
def emulate_pass1(t: TiledEdgeSet, x: np.ndarray, slice_agents: int, beta: float,
                  table: Optional[np.ndarray] = None):
    """Phases A+B: returns (val [E] block-major, cum [V])."""
    S, J = t.n_slices, t.n_blocks
    val = np.zeros(t.n_slots, dtype=np.float32)
    for s in range(S):
        xs = x[s * slice_agents:(s + 1) * slice_agents]
        for j in range(J):
            a, b = t.tile_sptr[s * J + j], t.tile_sptr[s * J + j + 1]
            p = t.tile_jpos[s * J + j]
            val[p:p + (b - a)] = xs[t.a_la[a:b]]
    cum = np.zeros(t.n_venues, dtype=np.float32)
    for j in range(J):
        e0, e1 = t.blk_e0[j], t.blk_e0[j + 1]
        v0, v1 = t.blk_v0[j], t.blk_v0[j + 1]
        real = t.e_lv[e0:e1] != 0xFFFF
        xv = val[e0:e1][real].astype(np.float64)
        if table is not None:
            xv = table[t.e_cls[e0:e1][real]].astype(np.float64) * xv
        sums = np.bincount(t.e_lv[e0:e1][real], weights=xv, minlength=v1 - v0)
        if t.runs is not None:      # run form: the block's primary edges, value x[a], local venue pv_blk[a]
            r0, r1 = int(t.runs.blk_r0[j]), int(t.runs.blk_r0[j + 1])
            lv = t.runs.pv_blk[r0:r1].astype(np.int64)
            assert (lv != 0xFFFF).all() and (lv < v1 - v0).all()
            sums = sums + np.bincount(lv, weights=x[r0:r1].astype(np.float64), minlength=v1 - v0)
        cum[v0:v1] = (np.float32(beta) * t.v_pcontact[v0:v1]) * sums.astype(np.float32)
    return val, cum


def emulate_pass2(t: TiledEdgeSet, cum: np.ndarray, n_agents: int, slice_agents: int,
                  weight_table: Optional[np.ndarray] = None):
    """Phases C+D: returns acc [A] = sum over the agent's edges of cum[venue] (x table[cls])."""
    S, J = t.n_slices, t.n_blocks
    cval = np.zeros(t.n_slots, dtype=np.float32)
    for j in range(J):
        e0, e1 = t.blk_e0[j], t.blk_e0[j + 1]
        real = t.e_lv[e0:e1] != 0xFFFF
        c = np.zeros(e1 - e0, dtype=np.float32)
        c[real] = cum[t.blk_v0[j] + t.e_lv[e0:e1][real].astype(np.int64)]
        if weight_table is not None:
            c = weight_table[t.e_cls[e0:e1]] * c
        cval[e0:e1] = c
    acc = np.zeros(S * slice_agents, dtype=np.float64)
    for s in range(S):
        for j in range(J):
            a, b = t.tile_sptr[s * J + j], t.tile_sptr[s * J + j + 1]
            p = t.tile_jpos[s * J + j]
            np.add.at(acc, s * slice_agents + t.a_la[a:b].astype(np.int64), cval[p:p + (b - a)])
    if t.runs is not None:          # run form: every owned agent reads its primary venue from its slice's window of cum
        rf = t.runs
        for s in range(len(rf.win_lo)):
            lo, n = int(rf.win_lo[s]), int(rf.win_n[s])
            window = cum[lo:lo + n]
            pv = rf.pv_win[s * slice_agents:(s + 1) * slice_agents].astype(np.int64)
            ok = pv != 0xFFFF
            assert (pv[ok] < n).all()
            acc[s * slice_agents:(s + 1) * slice_agents][ok] += window[pv[ok]]
    return acc[:n_agents].astype(np.float32)
