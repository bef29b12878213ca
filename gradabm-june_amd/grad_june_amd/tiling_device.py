"""Device-side build of the tiled layout (row f4 of SURVEY section 8: "native graph compile").

``build_tiled_device`` is ``tiling.build_tiled`` with every O(E) step (two stable sorts' worth of keys, prefix
sums, the scatter of the 16-bit local indices, the chunk descriptors) as torch ops on the tensors' device - on
an MI355X a 15 M-edge set compiles in a fraction of a second instead of ~2 s of numpy, and the arrays are born
in HBM (no host round trip).  Only the block boundaries (a greedy walk over a few thousand blocks) are found on
the host, from the per-venue degrees.  The result is bit-identical to the numpy build
(tests/test_tiling_host.py runs both on the CPU; tests/test_gpu_fullsize_properties.py on the device).

16-bit fields are returned as int16 tensors holding the uint16 bit patterns (torch has no uint16 arithmetic),
which is also how the numpy build's arrays are uploaded.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import tiling as TL


def _u16(x: torch.Tensor) -> torch.Tensor:
    """int64 values in [0, 65535] -> int16 with the same low 16 bits."""
    return torch.where(x >= 32768, x - 65536, x).to(torch.int16)


def _wide_descriptors(sptr, jpos_flat, chunk_ptr, first_edge, chunk_end, t0, J):
    """tiling.wide_descriptors on torch tensors."""
    n_chunks = first_edge.numel()
    dev = sptr.device
    tile_len = sptr[1:] - sptr[:-1]
    nonempty = tile_len > 0
    starts_all = sptr[:-1][nonempty]
    tile_ids = torch.nonzero(nonempty).squeeze(1)
    lo = torch.searchsorted(starts_all, first_edge, right=True)
    hi = torch.searchsorted(starts_all, chunk_end, right=False)
    nseg = 1 + (hi - lo)
    desc = torch.zeros((n_chunks, 8), dtype=torch.int64, device=dev)
    desc[:, 0] = jpos_flat[t0] + (first_edge - sptr[t0])
    starts = torch.full((n_chunks, TL.WIDE_SEGMENTS - 1), 64, dtype=torch.int64, device=dev)
    n_starts = starts_all.numel()
    for k in range(1, TL.WIDE_SEGMENTS):
        has = nseg > k
        if n_starts:
            b = torch.clamp(lo + (k - 1), max=n_starts - 1)
            pos = starts_all[b]
            st = torch.where(has, pos - first_edge, torch.full_like(pos, 64))
            base = torch.where(has, jpos_flat[tile_ids[b]] - st, torch.zeros_like(st))
        else:
            st = torch.full((n_chunks,), 64, dtype=torch.int64, device=dev)
            base = torch.zeros(n_chunks, dtype=torch.int64, device=dev)
        desc[:, k] = base
        starts[:, k - 1] = st
    multi = (nseg > TL.WIDE_SEGMENTS).to(torch.int64)
    S = chunk_ptr.numel() - 1
    j0 = t0 - torch.repeat_interleave(torch.arange(S, device=dev) * J, chunk_ptr[1:] - chunk_ptr[:-1])
    if j0.numel() and int(j0.max()) >= (1 << 22):
        raise ValueError("too many venue blocks for the wide descriptor's j0 field")
    desc[:, 6] = starts[:, 0] | (starts[:, 1] << 8) | (starts[:, 2] << 16) | (starts[:, 3] << 24)
    desc[:, 7] = starts[:, 4] | (multi << 8) | (j0 << 9)
    # low 32 bits as int32 (bases may be negative, word 6 may exceed 2^31)
    desc = desc & 0xFFFFFFFF
    return torch.where(desc >= (1 << 31), desc - (1 << 32), desc).to(torch.int32)


def build_tiled_device(name: str, agent_index: torch.Tensor, venue_index: torch.Tensor, n_venues: int,
                       v_pcontact, n_slices: int, slice_agents: int, agent_class: Optional[torch.Tensor] = None,
                       sv_max: int = TL.SV_MAX, eb_target: int = TL.EB_TARGET,
                       wide: Optional[bool] = None) -> TL.TiledEdgeSet:
    dev = agent_index.device
    agent = agent_index.reshape(-1).to(torch.int64)
    venue = venue_index.reshape(-1).to(torch.int64)
    E = agent.numel()
    if slice_agents > 65536 or sv_max > 65536:
        raise ValueError("local indices are 16-bit")
    S = n_slices
    i32 = lambda t: t.to(torch.int32)
    v_pc = torch.as_tensor(np.asarray(v_pcontact, dtype=np.float32)) if not isinstance(v_pcontact, torch.Tensor) \
        else v_pcontact.to(torch.float32)
    degree = torch.bincount(venue, minlength=n_venues) if E else torch.zeros(n_venues, dtype=torch.int64, device=dev)
    blk_v0_np = TL.venue_blocks(degree.cpu().numpy(), sv_max, eb_target)      # a few thousand blocks: host walk
    J = len(blk_v0_np) - 1
    if J == 0:
        z = torch.zeros(1, dtype=torch.int32, device=dev)
        return TL.TiledEdgeSet(name, n_venues, 0, S, 0, z, z, torch.zeros(0, dtype=torch.int16, device=dev), None,
                               torch.zeros(0, dtype=torch.int16, device=dev), z.clone(),
                               torch.zeros(0, dtype=torch.int32, device=dev), v_pc.to(dev), 0,
                               torch.zeros(S + 1, dtype=torch.int32, device=dev),
                               torch.zeros((0, 4), dtype=torch.int32, device=dev))
    blk_v0 = torch.from_numpy(blk_v0_np).to(dev)
    vblk = torch.searchsorted(blk_v0, torch.arange(n_venues, device=dev), right=True) - 1
    j = vblk[venue]
    lv = venue - blk_v0[j]
    s = torch.div(agent, slice_agents, rounding_mode="floor")
    la = agent - s * slice_agents
    if E and int(s.max()) >= S:
        raise ValueError("agent index beyond the last slice")
    tile = j * S + s
    key = (tile * 65536 + lv) * 65536 + la
    order = torch.argsort(key, stable=True)
    del key
    tile_of = tile[order]
    tile_len_js = torch.bincount(tile_of, minlength=J * S).reshape(J, S)
    upos_js = torch.zeros(J * S + 1, dtype=torch.int64, device=dev)
    upos_js[1:] = torch.cumsum(tile_len_js.reshape(-1), 0)
    blk_len = tile_len_js.sum(1)
    blk_slots = torch.div(blk_len + (TL.PAD - 1), TL.PAD, rounding_mode="floor") * TL.PAD
    blk_start = torch.zeros(J + 1, dtype=torch.int64, device=dev)
    blk_start[1:] = torch.cumsum(blk_slots, 0)
    blk_ustart = upos_js[0:J * S:S]
    shift = torch.repeat_interleave(blk_start[:-1] - blk_ustart, S)
    jpos_js = upos_js[:-1] + shift                                              # [J*S]
    n_slots = int(blk_start[-1])
    tile_len_sj = tile_len_js.t().contiguous()
    sptr = torch.zeros(S * J + 1, dtype=torch.int64, device=dev)
    sptr[1:] = torch.cumsum(tile_len_sj.reshape(-1), 0)
    jpos_sj = jpos_js.reshape(J, S).t().contiguous().reshape(-1)                # block-major start of tile (s, j)
    tj = torch.div(tile_of, S, rounding_mode="floor")
    ts = tile_of - tj * S
    within = torch.arange(E, dtype=torch.int64, device=dev) - upos_js[tile_of]
    pos_sm = sptr[ts * J + tj] + within
    pos_bm = jpos_js[tile_of] + within
    a_la = torch.empty(E, dtype=torch.int16, device=dev)
    a_la[pos_sm] = _u16(la[order])
    e_lv = torch.full((n_slots,), -1, dtype=torch.int16, device=dev)            # 0xFFFF = pad
    e_lv[pos_bm] = _u16(lv[order])
    e_cls = None
    if agent_class is not None:
        e_cls = torch.zeros(n_slots, dtype=torch.uint8, device=dev)
        e_cls[pos_bm] = agent_class.to(device=dev, dtype=torch.uint8)[agent[order]]
    del pos_sm, pos_bm, within, tj, ts, order, tile_of
    seg = sptr[0:S * J + 1:J]
    n_chunks = torch.div((seg[1:] - seg[:-1]) + (TL.CHUNK - 1), TL.CHUNK, rounding_mode="floor")
    chunk_ptr = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    chunk_ptr[1:] = torch.cumsum(n_chunks, 0)
    total_chunks = int(chunk_ptr[-1])
    first_edge = (torch.repeat_interleave(seg[:-1], n_chunks)
                  + TL.CHUNK * (torch.arange(total_chunks, device=dev) - torch.repeat_interleave(chunk_ptr[:-1], n_chunks)))
    seg_end = torch.repeat_interleave(seg[1:], n_chunks)
    chunk_end = torch.minimum(first_edge + TL.CHUNK, seg_end)
    t0 = torch.searchsorted(sptr, first_edge, right=True) - 1
    end0 = sptr[t0 + 1]
    split = torch.minimum(end0, chunk_end) - first_edge
    slot0 = jpos_sj[t0] + (first_edge - sptr[t0])
    two = end0 < chunk_end
    t1 = torch.where(two, torch.searchsorted(sptr, torch.clamp(end0, max=max(E - 1, 0)), right=True) - 1, t0)
    slot1 = torch.where(two, jpos_sj[t1] + (end0 - sptr[t1]), torch.zeros_like(t1))
    multi = two & (sptr[t1 + 1] < chunk_end)
    j0 = t0 - torch.repeat_interleave(torch.arange(S, device=dev) * J, n_chunks)
    chunk_desc = i32(torch.stack([slot0, slot1, split + (multi.to(torch.int64) << 16), j0], dim=1))
    if wide is None:
        wide = bool(multi.numel()) and float(multi.to(torch.float32).mean()) > TL.WIDE_MIN_SHARE
    if wide:
        chunk_desc = _wide_descriptors(sptr, jpos_sj, chunk_ptr, first_edge, chunk_end, t0, J)
    return TL.TiledEdgeSet(
        name=name, n_venues=n_venues, n_edges=E, n_slices=S, n_blocks=J,
        blk_v0=i32(blk_v0), blk_e0=i32(blk_start), e_lv=e_lv, e_cls=e_cls, a_la=a_la,
        tile_sptr=i32(sptr), tile_jpos=i32(jpos_sj), v_pcontact=v_pc.to(dev), n_slots=n_slots,
        chunk_ptr=i32(chunk_ptr), chunk_desc=chunk_desc.contiguous(), desc_wide=bool(wide))


def ell_degree_max(agent_index: torch.Tensor, n_agents: int):
    """(edges of owned agents, their maximum degree) - what tiling.direct_eligible needs."""
    agent = agent_index.reshape(-1).to(torch.int64)
    a = agent[agent < n_agents]
    if not a.numel():
        return 0, 0
    return int(a.numel()), int(torch.bincount(a, minlength=n_agents).max())


def build_ell_device(agent_index: torch.Tensor, venue_index: torch.Tensor, n_agents: int, n_slices_owned: int,
                     slice_agents: int):
    """tiling.build_ell with torch ops on the tensors' device: (int16 [planes, rows, min(K, 2)] holding uint16 bit patterns, K)."""
    dev = agent_index.device
    agent = agent_index.reshape(-1).to(torch.int64)
    venue = venue_index.reshape(-1).to(torch.int64)
    owned = agent < n_agents
    a, v = agent[owned], venue[owned]
    order = torch.argsort(a, stable=True)
    a, v = a[order], v[order]
    deg = torch.bincount(a, minlength=n_agents)
    K = TL.direct_columns(int(deg.max()) if deg.numel() else 1)
    rowptr = torch.zeros(n_agents + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(deg, 0)
    col = torch.arange(a.numel(), dtype=torch.int64, device=dev) - rowptr[a]
    ell = torch.full((n_slices_owned * slice_agents, K), -1, dtype=torch.int16, device=dev)   # 0xFFFF = none
    ell[a, col] = _u16(v)
    return TL.ell_planes(ell, K).contiguous(), K
