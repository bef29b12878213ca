"""ctypes binding of libgradjune_hip.so (C ABI: include/gradjune_hip.h).

There is NO fallback: if the library has not been built (``python __graft_entry__.py`` or
``make -C gradabm-june_amd/csrc``) every compute entry point raises.  PyTorch is used only to
own device memory and to name the HIP stream the kernels are launched on.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

GJ_ABI_VERSION = 6
GJ_MAX_SETS = 12
GJ_MAX_NETS = 16
GJ_MAX_NETS_PER_SET = 8
GJ_TABLE_SIZE = 400
GJ_ADJ_BETA_BLOCKS = 256
GJ_STREAM_EDGES = 2048

MASK_RAW, MASK_Q, MASK_QL, MASK_QL_AGE75 = 0, 1, 2, 3

_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
# GJ_LIB_PATH: another build of the same library (kernel experiments, tools/ab.py); never a different implementation
LIB_PATH = os.environ.get("GJ_LIB_PATH") or os.path.join(_LIB_DIR, "libgradjune_hip.so")

_vp = C.c_void_p


class EdgeSet(C.Structure):
    _fields_ = [
        ("n_venues", C.c_int64),
        ("n_edges", C.c_int64),
        ("v_rowptr", _vp),
        ("v_agent", _vp),
        ("v_pcontact", _vp),
        ("a_rowptr", _vp),
        ("a_venue", _vp),
        ("cum", _vp),
        ("cum_stride", C.c_int32),
        ("_pad", C.c_int32),
    ]


class TiledSet(C.Structure):
    _fields_ = [
        ("n_blocks", C.c_int32),
        ("max_block_venues", C.c_int32),
        ("desc_wide", C.c_int32),
        ("ell_k", C.c_int32),
        ("blk_v0", _vp),
        ("blk_e0", _vp),
        ("e_lv", _vp),
        ("e_cls", _vp),
        ("a_la", _vp),
        ("tile_sptr", _vp),
        ("tile_jpos", _vp),
        ("chunk_ptr", _vp),
        ("chunk_desc", _vp),
        ("val", _vp),
        ("ell", _vp),
        ("run_pv_blk", _vp),
        ("run_pv_win", _vp),
        ("run_blk_r0", _vp),
        ("run_win_lo", _vp),
        ("run_win_n", _vp),
        ("run_max_window", C.c_int32),
        ("run_tiled_edges", C.c_int32),
        ("presum", _vp),
        ("multi_slots", _vp),
        ("max_venue_edges", C.c_int32),
        ("_pad_mve", C.c_int32),
    ]


class Tiled(C.Structure):
    _fields_ = [
        ("n_slices", C.c_int32),
        ("slice_agents", C.c_int32),
        ("direct_table_floats", C.c_int32),
        ("n_work", C.c_int32),
        ("work", _vp),
        ("agent_scratch", _vp),
        ("presum_wgs", C.c_int32),
        ("_pad_presum", C.c_int32),
        ("sets", TiledSet * GJ_MAX_SETS),
    ]


class Plan(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int64),
        ("n_ext_agents", C.c_int64),
        ("n_sets", C.c_int32),
        ("n_blocks", C.c_int32),
        ("n_long_rows", C.c_int32),
        ("n_partial_slots", C.c_int32),
        ("sets", EdgeSet * GJ_MAX_SETS),
        ("blocks", _vp),
        ("long_rows", _vp),
        ("partial", _vp),
        ("agent_class", _vp),
        ("tables", _vp),
        ("n_tables", C.c_int32),
        ("_pad", C.c_int32),
        ("tiled", C.POINTER(Tiled)),
    ]


class Network(C.Structure):
    _fields_ = [
        ("beta", C.c_float),
        ("set", C.c_int32),
        ("mask_kind", C.c_int32),
        ("table", C.c_int32),
    ]


class StepParams(C.Structure):
    _fields_ = [
        ("now", C.c_float),
        ("delta_time", C.c_float),
        ("day_type", C.c_int32),
        ("has_quarantine", C.c_int32),
        ("q_threshold", C.c_float),
        ("n_nets", C.c_int32),
        ("seed", C.c_uint64),
        ("step", C.c_uint64),
        ("agent_offset", C.c_int64),
        ("transpose", C.c_int32),
        ("_pad", C.c_int32),
        ("nets", Network * GJ_MAX_NETS),
        ("clock", _vp),
    ]


class AgentState(C.Structure):
    _fields_ = [
        ("max_infectiousness", _vp),
        ("shape", _vp),
        ("rate", _vp),
        ("shift", _vp),
        ("infection_time", _vp),
        ("is_infected", _vp),
        ("susceptibility", _vp),
        ("transmission", _vp),
        ("q_transmission", _vp),
        ("current_stage", _vp),
    ]


class StepIO(C.Structure):
    _fields_ = [
        ("not_infected_probs", _vp),
        ("new_infected", _vp),
        ("exp_noise", _vp),
        ("trans_susc", _vp),
        ("agent_sums", _vp),
    ]


GJ_MAX_STAGES = 16


class SymptomsParams(C.Structure):
    _fields_ = [
        ("n_stages", C.c_int32),
        ("_pad", C.c_int32),
        ("progress", _vp),
        ("next_kind", C.c_int32 * GJ_MAX_STAGES),
        ("next_loc", C.c_float * GJ_MAX_STAGES),
        ("next_scale", C.c_float * GJ_MAX_STAGES),
        ("rec_kind", C.c_int32 * GJ_MAX_STAGES),
        ("rec_loc", C.c_float * GJ_MAX_STAGES),
        ("rec_scale", C.c_float * GJ_MAX_STAGES),
        ("time", C.c_float),
        ("_pad2", C.c_float),
        ("seed", C.c_uint64),
        ("step", C.c_uint64),
        ("agent_offset", C.c_int64),
    ]


#: every symbol include/gradjune_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "gj_version": (C.c_int, []),
    "gj_error_string": (C.c_char_p, [C.c_int]),
    "gj_check_device": (C.c_int, []),
    "gj_transmission_update": (C.c_int, [C.POINTER(Plan), C.POINTER(AgentState), C.POINTER(StepParams), _vp]),
    "gj_quarantine_transmission": (C.c_int, [C.POINTER(Plan), C.POINTER(AgentState), C.POINTER(StepParams), _vp]),
    "gj_venue_reduce": (C.c_int, [C.POINTER(Plan), C.POINTER(AgentState), C.POINTER(StepParams), _vp]),
    "gj_agent_gather": (
        C.c_int,
        [C.POINTER(Plan), C.POINTER(AgentState), C.POINTER(StepParams), C.POINTER(StepIO), C.c_int, _vp],
    ),
    "gj_sample_infect": (
        C.c_int,
        [C.c_int64, _vp, _vp, C.c_uint64, C.c_uint64, C.c_int64, C.c_float, _vp, _vp, _vp, _vp, _vp],
    ),
    "gj_adjoint_sample": (
        C.c_int,
        [C.c_int64, _vp, _vp, _vp, _vp, C.c_uint64, C.c_uint64, C.c_int64, C.c_float, C.c_float, _vp, _vp, _vp, _vp,
         _vp, _vp, _vp, _vp],
    ),
    "gj_adjoint_transmission": (C.c_int, [C.c_int64, C.POINTER(AgentState), C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "gj_adjoint_beta_partial": (
        C.c_int,
        [C.c_int64, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), _vp, _vp],
    ),
    "gj_adjoint_beta_finish": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp]),
    "gj_symptoms_update": (
        C.c_int,
        [C.c_int64, _vp, _vp, _vp, _vp, _vp, C.POINTER(SymptomsParams), _vp, _vp, _vp],
    ),
    "gj_adjoint_symptoms": (
        C.c_int,
        [C.c_int64, _vp, _vp, _vp, _vp, _vp, C.POINTER(SymptomsParams)] + [_vp] * 10,
    ),
    "gj_step_stats": (C.c_int, [C.c_int64, _vp, _vp, _vp, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _vp, _vp]),
    "gj_step": (C.c_int, [C.POINTER(Plan), C.POINTER(AgentState), C.POINTER(StepParams), C.POINTER(StepIO), _vp]),
    "gj_step_phase": (
        C.c_int,
        [C.POINTER(Plan), C.POINTER(AgentState), C.POINTER(StepParams), C.POINTER(StepIO), C.c_int, _vp],
    ),
    "gj_clock_advance": (C.c_int, [_vp, C.c_double, _vp]),
    "gj_symptoms_step_stats": (
        C.c_int,
        [C.c_int64, _vp, _vp, _vp, _vp, _vp, C.POINTER(SymptomsParams), _vp, _vp, _vp, C.c_int32, C.POINTER(C.c_int32),
         C.c_int32, _vp, _vp],
    ),
    "gj_pack_f32": (C.c_int, [C.c_int64, _vp, _vp, _vp, _vp]),
    "gj_unpack_f32": (C.c_int, [C.c_int64, _vp, _vp, _vp, _vp]),
    "gj_event_create": (C.c_int, [C.POINTER(_vp)]),
    "gj_event_record": (C.c_int, [_vp, _vp]),
    "gj_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "gj_event_destroy": (C.c_int, [_vp]),
}

# ---- graph compile on the device (include/gradjune_hip.h, "graph compile") ----
GJ_COMPILE_COUNTS = 16
GJ_CC_BLOCKS, GJ_CC_SLOTS, GJ_CC_CHUNKS, GJ_CC_MULTI, GJ_CC_OWNED_EDGES, GJ_CC_MAX_DEGREE, GJ_CC_ERROR = 0, 1, 2, 3, 4, 5, 7
GJ_CC_RUN_PRIMARY, GJ_CC_RUN_UNSORTED, GJ_CC_RUN_WINDOW, GJ_CC_WIDE_MULTI = 8, 9, 10, 11


class CompileSet(C.Structure):
    _fields_ = [
        ("agent", _vp), ("venue", _vp), ("agent_class", _vp),
        ("n_edges", C.c_int64), ("n_agents", C.c_int64), ("n_ext_agents", C.c_int64),
        ("n_venues", C.c_int32), ("n_slices", C.c_int32), ("slice_agents", C.c_int32),
        ("sv_max", C.c_int32), ("eb_target", C.c_int32), ("n_blocks", C.c_int32),
    ]


class CompileOut(C.Structure):
    _fields_ = [
        ("blk_e0", _vp), ("e_lv", _vp), ("e_cls", _vp), ("a_la", _vp), ("tile_sptr", _vp), ("tile_jpos", _vp),
        ("chunk_ptr", _vp), ("chunk_desc", _vp), ("slots_cap", C.c_int64), ("chunks_cap", C.c_int64),
    ]


_i64p = C.POINTER(C.c_int64)
SYMBOLS.update({
    "gj_compile_capacity": (C.c_int, [C.POINTER(CompileSet), _i64p, _i64p, _i64p]),
    "gj_compile_workspace_bytes": (C.c_int, [C.POINTER(CompileSet), _i64p]),
    "gj_compile_blocks": (C.c_int, [C.POINTER(CompileSet), _vp, C.c_int32, _vp, _vp, C.c_int64, _vp]),
    "gj_compile_tiles": (C.c_int, [C.POINTER(CompileSet), _vp, C.POINTER(CompileOut), _vp, _vp, C.c_int64, _vp]),
    "gj_compile_wide_descriptors": (C.c_int, [C.POINTER(CompileSet), C.POINTER(CompileOut), C.c_int32, _vp, _vp, _vp]),
    "gj_compile_ell_degrees": (C.c_int, [C.POINTER(CompileSet), _vp, _vp, _vp]),
    "gj_compile_explicit_slots": (C.c_int, [C.POINTER(CompileSet), C.POINTER(CompileOut), _vp, _vp]),
    "gj_compile_multi_slots": (C.c_int, [C.POINTER(CompileSet), C.POINTER(CompileOut), C.c_int32, C.c_int32, _vp,
                                          C.c_int32, _vp, _vp, _vp, C.c_int64, _vp]),
    "gj_compile_runs_pick": (C.c_int, [C.POINTER(CompileSet), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gj_compile_runs_rest": (C.c_int, [C.POINTER(CompileSet), _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp]),
    "gj_compile_runs_index": (C.c_int, [C.POINTER(CompileSet), _vp, _vp, C.c_int32, C.c_int64, _vp, _vp, _vp, _vp, _vp]),
    "gj_compile_ell": (C.c_int, [C.POINTER(CompileSet), C.c_int32, C.c_int64, _vp, _vp, _vp, C.c_int64, _vp, _vp]),
})

_lib: Optional[C.CDLL] = None


class NativeLibraryMissing(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library, binding every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "grad_june_amd has no CPU or eager-PyTorch fallback for the infection path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.gj_version() != GJ_ABI_VERSION:
        raise RuntimeError(f"libgradjune_hip ABI {lib.gj_version()} != binding {GJ_ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().gj_error_string(code)
        raise RuntimeError(f"{what} failed: [{code}] {msg.decode() if msg else '?'}")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream() -> Optional[int]:
    """Raw hipStream_t of torch's current stream on the current device.  This sits on every launch:
    ``torch.cuda.current_stream().cuda_stream`` costs ~8 us, the two C calls below well under 1 us."""
    import torch

    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:       # a torch build without the private accessors
        return torch.cuda.current_stream().cuda_stream
