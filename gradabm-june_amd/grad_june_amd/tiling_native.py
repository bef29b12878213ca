"""The tiled layout of an edge set built ON THE DEVICE by the library's compile kernels (row f4 of SURVEY section 8:
"native graph compile"; ``csrc/gj_compile.hip``, C ABI ``gj_compile_*``).

``build_tiled_native`` is ``tiling.build_tiled`` and ``build_ell_native`` is ``tiling.build_ell`` with every O(E)
step a HIP kernel (key construction, rocPRIM radix sort and scans, scatter of the 16-bit local indices, chunk
descriptors, the greedy venue-block walk): the reference's unsorted COO ``edge_index`` goes in as it lies in HBM, the
arrays of ``gj_tiled_set`` come out in HBM, bit-identical to the numpy specification
(tests/test_gpu_compile_native.py).  A 15 M-edge set compiles in ~20 ms instead of ~2 s of numpy.  torch is used for
what the tier allows it: device memory (the output buffers, the workspace) and the stream.

16-bit fields are returned as int16 tensors holding the uint16 bit patterns (torch has no uint16 arithmetic),
which is also how the numpy build's arrays are uploaded.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _native as N
from . import tiling as TL

_ERRORS = {1: "agent index out of range", 2: "venue index out of range", 3: "more venue blocks than expected",
           4: "too many venue blocks for the wide descriptor's j0 field",
           5: "an owned agent has more edges than the ELL table has columns"}


def _i64(t: torch.Tensor, device) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(t, dtype=np.int64)))
    return t.reshape(-1).to(device=device, dtype=torch.int64).contiguous()


class _Session:
    """One edge set's compile: the ctypes argument block, the counts and the workspace."""

    def __init__(self, agent, venue, n_venues, n_agents, n_ext, n_slices, slice_agents, sv_max, eb_target, agent_class,
                 device):
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("the native graph compile runs on a HIP device (host build: tiling.build_tiled)")
        self.lib = N.load()
        self.agent, self.venue = _i64(agent, self.dev), _i64(venue, self.dev)
        if self.agent.numel() != self.venue.numel():
            raise ValueError("agent / venue index lengths differ")
        self.cls = None
        if agent_class is not None:
            c = agent_class if isinstance(agent_class, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(agent_class))
            self.cls = c.to(device=self.dev, dtype=torch.uint8).contiguous()
        self.set = N.CompileSet(N.ptr(self.agent), N.ptr(self.venue), N.ptr(self.cls), self.agent.numel(), int(n_agents),
                                int(n_ext), int(n_venues), int(n_slices), int(slice_agents), int(sv_max), int(eb_target), 0)
        self.counts = torch.zeros(N.GJ_COMPILE_COUNTS, dtype=torch.int32, device=self.dev)
        self.ws = None

    def workspace(self):
        need = C.c_int64(0)
        N.check(self.lib.gj_compile_workspace_bytes(C.byref(self.set), C.byref(need)), "gj_compile_workspace_bytes")
        if self.ws is None or self.ws.numel() < need.value:
            self.ws = None
            self.ws = torch.empty(need.value, dtype=torch.uint8, device=self.dev)
        return self.ws

    def read_counts(self, name: str):
        c = self.counts.cpu().numpy()
        if c[N.GJ_CC_ERROR]:
            raise ValueError(f"{name}: {_ERRORS.get(int(c[N.GJ_CC_ERROR]), 'compile error')}")
        return c


def build_tiled_native(name: str, agent_index, venue_index, n_venues: int, v_pcontact, n_slices: int, slice_agents: int,
                       agent_class=None, sv_max: int = TL.SV_MAX, eb_target: int = TL.EB_TARGET,
                       wide: Optional[bool] = None, device=None, n_ext_agents: Optional[int] = None,
                       explicit: Optional[bool] = None, multi_rows: bool = True) -> TL.TiledEdgeSet:
    if slice_agents > 65536 or sv_max > 65535:
        raise ValueError("local indices are 16-bit")
    dev = torch.device(device if device is not None else agent_index.device)
    v_pc = (torch.as_tensor(np.asarray(v_pcontact, dtype=np.float32)) if not isinstance(v_pcontact, torch.Tensor)
            else v_pcontact.to(torch.float32)).to(dev)
    S = int(n_slices)
    n_ext = S * int(slice_agents) if n_ext_agents is None else int(n_ext_agents)
    se = _Session(agent_index, venue_index, n_venues, 0, n_ext, S, slice_agents, sv_max, eb_target, agent_class, dev)
    E = se.agent.numel()
    i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        st = N.current_stream()
        lib = se.lib
        blk_cap, slots_cap, chunks_cap = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        N.check(lib.gj_compile_capacity(C.byref(se.set), C.byref(blk_cap), None, None), "gj_compile_capacity")
        blk_v0 = i32(blk_cap.value + 1)
        ws = se.workspace()
        N.check(lib.gj_compile_blocks(C.byref(se.set), N.ptr(blk_v0), blk_cap.value, N.ptr(se.counts), N.ptr(ws),
                                      ws.numel(), st), "gj_compile_blocks")
        c = se.read_counts(name)
        J = int(c[N.GJ_CC_BLOCKS])
        if J == 0:
            z = torch.zeros(1, dtype=torch.int32, device=dev)
            return TL.TiledEdgeSet(name, n_venues, 0, S, 0, z, z, torch.zeros(0, dtype=torch.int16, device=dev), None,
                                   torch.zeros(0, dtype=torch.int16, device=dev), z.clone(),
                                   torch.zeros(0, dtype=torch.int32, device=dev), v_pc, 0,
                                   torch.zeros(S + 1, dtype=torch.int32, device=dev),
                                   torch.zeros((0, 4), dtype=torch.int32, device=dev))
        se.set.n_blocks = J
        N.check(lib.gj_compile_capacity(C.byref(se.set), None, C.byref(slots_cap), C.byref(chunks_cap)),
                "gj_compile_capacity")
        blk_e0, sptr, jpos, chunk_ptr = i32(J + 1), i32(S * J + 1), i32(S * J), i32(S + 1)
        e_lv = torch.empty(slots_cap.value, dtype=torch.int16, device=dev)
        e_cls = torch.empty(slots_cap.value, dtype=torch.uint8, device=dev) if se.cls is not None else None
        a_la = torch.empty(max(E, 1), dtype=torch.int16, device=dev)
        desc = i32(chunks_cap.value * 4)
        out = N.CompileOut(N.ptr(blk_e0), N.ptr(e_lv), N.ptr(e_cls), N.ptr(a_la), N.ptr(sptr), N.ptr(jpos),
                           N.ptr(chunk_ptr), N.ptr(desc), slots_cap.value, chunks_cap.value)
        ws = se.workspace()
        N.check(lib.gj_compile_tiles(C.byref(se.set), N.ptr(blk_v0), C.byref(out), N.ptr(se.counts), N.ptr(ws),
                                     ws.numel(), st), "gj_compile_tiles")
        c = se.read_counts(name)
        n_slots, n_chunks, n_multi = int(c[N.GJ_CC_SLOTS]), int(c[N.GJ_CC_CHUNKS]), int(c[N.GJ_CC_MULTI])
        if wide is None:
            wide = n_chunks > 0 and (n_multi / n_chunks) > TL.WIDE_MIN_SHARE
        slot_idx = None
        if wide:
            desc = i32(n_chunks * 8)
            se.counts[N.GJ_CC_WIDE_MULTI] = 0
            N.check(lib.gj_compile_wide_descriptors(C.byref(se.set), C.byref(out), n_chunks, N.ptr(desc),
                                                    N.ptr(se.counts), st), "gj_compile_wide_descriptors")
            c = se.read_counts(name)
            chunk_desc = desc[: n_chunks * 8].reshape(n_chunks, 8)
            if explicit is None:
                explicit = n_chunks > 0 and int(c[N.GJ_CC_WIDE_MULTI]) / n_chunks > TL.EXPLICIT_MIN_SHARE
        else:
            chunk_desc = desc[: n_chunks * 4].reshape(n_chunks, 4).clone()      # (drops the upper-bound tail)
        multi_slots = None
        if explicit and E > 0:
            slot_idx = torch.empty(E, dtype=torch.int32, device=dev)
            N.check(lib.gj_compile_explicit_slots(C.byref(se.set), C.byref(out), N.ptr(slot_idx), st),
                    "gj_compile_explicit_slots")
            torch.cuda.current_stream().synchronize()
        elif multi_rows and n_chunks > 0:
            # the chunks the descriptors cannot express get a row of explicit slots each (tiling.attach_multi_slots)
            n_m = int(c[N.GJ_CC_WIDE_MULTI]) if wide else n_multi
            if n_m > 0:
                chunk_desc = chunk_desc.contiguous()
                multi_slots = torch.empty((n_m, 64), dtype=torch.int32, device=dev)
                ws2 = torch.empty(2 * (4 * (n_chunks + 1) + 256) + (1 << 20), dtype=torch.uint8, device=dev)
                N.check(lib.gj_compile_multi_slots(C.byref(se.set), C.byref(out), n_chunks, 1 if wide else 0,
                                                   N.ptr(chunk_desc), n_m, N.ptr(multi_slots), N.ptr(se.counts), N.ptr(ws2),
                                                   ws2.numel(), st), "gj_compile_multi_slots")
                se.read_counts(name)
        se.ws = None
    return TL.TiledEdgeSet(
        name=name, n_venues=n_venues, n_edges=E, n_slices=S, n_blocks=J,
        blk_v0=blk_v0[: J + 1].clone(), blk_e0=blk_e0, e_lv=e_lv[:n_slots].clone(),
        e_cls=None if e_cls is None else e_cls[:n_slots].clone(), a_la=a_la[:E],
        tile_sptr=sptr, tile_jpos=jpos, v_pcontact=v_pc, n_slots=n_slots,
        chunk_ptr=chunk_ptr, chunk_desc=chunk_desc.contiguous(), desc_wide=bool(wide), slot_idx=slot_idx,
        multi_slots=multi_slots)


class EllBuilder:
    """The ELL rows of the direct form of pass 2, in two steps: ``degrees()`` (what tiling.direct_eligible needs),
    then ``build()`` if the set qualifies."""

    def __init__(self, agent_index, venue_index, n_venues: int, n_agents: int, slice_agents: int, device):
        n_owned_slices = max(1, -(-int(n_agents) // int(slice_agents)))
        self.rows = n_owned_slices * int(slice_agents)
        self.n_agents = int(n_agents)
        # (sv_max / eb_target / n_slices do not matter for the ELL stages; n_slices bounds the agent ids)
        self.se = _Session(agent_index, venue_index, n_venues, n_agents, n_agents, n_owned_slices, slice_agents,
                           TL.SV_MAX, TL.EB_TARGET, None, device)
        self.degree = torch.empty(self.n_agents + 1, dtype=torch.int32, device=self.se.dev)

    def degrees(self):
        """(edges of owned agents, their maximum degree)."""
        se = self.se
        with torch.cuda.device(se.dev):
            N.check(se.lib.gj_compile_ell_degrees(C.byref(se.set), N.ptr(self.degree), N.ptr(se.counts),
                                                  N.current_stream()), "gj_compile_ell_degrees")
        c = se.counts.cpu().numpy()
        return int(c[N.GJ_CC_OWNED_EDGES]), int(c[N.GJ_CC_MAX_DEGREE])

    def build(self, degree_max: int):
        """(int16 [planes, rows, 2] holding uint16 bit patterns, K) - after degrees()."""
        se = self.se
        K = TL.direct_columns(max(1, degree_max))
        ell = torch.empty((K // 2, self.rows, 2), dtype=torch.int16, device=se.dev)
        with torch.cuda.device(se.dev):
            ws = se.workspace()
            N.check(se.lib.gj_compile_ell(C.byref(se.set), K, self.rows, N.ptr(self.degree), N.ptr(ell), N.ptr(ws),
                                          ws.numel(), N.ptr(se.counts), N.current_stream()), "gj_compile_ell")
            if int(se.counts[N.GJ_CC_ERROR].item()):          # (synchronises)
                raise ValueError(f"ELL table of {K} columns does not hold every owned agent's edges")
        se.ws = None
        return ell, K


def split_primary_runs_native(agent_index, venue_index, n_agents: int, n_venues: int, slice_agents: int, device,
                              min_share: float = TL.RUN_MIN_SHARE):
    """``tiling.split_primary_runs`` on the device (gj_compile_runs_pick / _rest): returns (RunForm or None, {"agent",
    "venue"}: the edges that stay in the tiled arrays, int64 tensors in COO order).  ``RunForm.vmin`` / windows are
    device tensors; ``keep`` is not kept."""
    n_own_slices = max(1, -(-int(n_agents) // int(slice_agents)))
    if int(n_agents) <= 0:
        return None, None
    se = _Session(agent_index, venue_index, n_venues, n_agents, max(int(n_agents), 1), n_own_slices, slice_agents,
                  TL.SV_MAX, TL.EB_TARGET, None, device)
    # (n_ext of this session = the owned agents: halo agents' edges are never primary, the range check of their
    # indices is gj_compile_blocks' on the remaining edges)
    se.set.n_ext_agents = n_own_slices * int(slice_agents)
    dev, E = se.dev, se.agent.numel()
    if E == 0:
        return None, None
    i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
    vmin, pick, win_lo, win_n = i32(n_agents), i32(n_agents), i32(n_own_slices), i32(n_own_slices)
    keep = torch.empty(E, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = N.current_stream()
        N.check(se.lib.gj_compile_runs_pick(C.byref(se.set), N.ptr(vmin), N.ptr(pick), N.ptr(keep), N.ptr(win_lo),
                                            N.ptr(win_n), N.ptr(se.counts), st), "gj_compile_runs_pick")
        c = se.counts.cpu().numpy()
        n_primary, owned = int(c[N.GJ_CC_RUN_PRIMARY]), int(c[N.GJ_CC_OWNED_EDGES])
        if c[N.GJ_CC_RUN_UNSORTED] or n_primary == 0 or int(c[N.GJ_CC_RUN_WINDOW]) > TL.RUN_MAX_WINDOW \
                or n_primary < min_share * owned:
            return None, None
        rest_a = torch.empty(max(E - n_primary, 1), dtype=torch.int64, device=dev)
        rest_v = torch.empty(max(E - n_primary, 1), dtype=torch.int64, device=dev)
        ws = se.workspace()
        N.check(se.lib.gj_compile_runs_rest(C.byref(se.set), N.ptr(keep), N.ptr(rest_a), N.ptr(rest_v), N.ptr(ws),
                                            ws.numel(), N.ptr(se.counts), st), "gj_compile_runs_rest")
        torch.cuda.current_stream().synchronize()
    rf = TL.RunForm(n_primary=n_primary, keep=None, vmin=vmin[:n_agents], win_lo=win_lo, win_n=win_n,
                    max_window=int(c[N.GJ_CC_RUN_WINDOW]))
    return rf, {"agent": rest_a[: E - n_primary], "venue": rest_v[: E - n_primary]}


def finish_run_form_native(rf, blk_v0, n_agents: int, slice_agents: int, device):
    """``tiling.finish_run_form`` on the device (gj_compile_runs_index)."""
    dev = torch.device(device)
    n_own_slices = max(1, -(-int(n_agents) // int(slice_agents)))
    rows = n_own_slices * int(slice_agents)
    blk_v0 = blk_v0.to(device=dev, dtype=torch.int32).contiguous()
    J = blk_v0.numel() - 1
    lib = N.load()
    cset = N.CompileSet(None, None, None, 0, int(n_agents), rows, 0, n_own_slices, int(slice_agents), TL.SV_MAX,
                        TL.EB_TARGET, 0)
    pv_blk = torch.empty(rows, dtype=torch.int16, device=dev)
    pv_win = torch.empty(rows, dtype=torch.int16, device=dev)
    blk_r0 = torch.empty(J + 1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        N.check(lib.gj_compile_runs_index(C.byref(cset), N.ptr(rf.vmin), N.ptr(blk_v0), J, rows, N.ptr(rf.win_lo),
                                          N.ptr(pv_blk), N.ptr(pv_win), N.ptr(blk_r0), N.current_stream()),
                "gj_compile_runs_index")
        torch.cuda.current_stream().synchronize()
    rf.pv_blk, rf.pv_win, rf.blk_r0 = pv_blk, pv_win, blk_r0
    return rf
