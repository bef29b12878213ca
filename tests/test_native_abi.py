"""CPU: the C-ABI library builds, loads, and exports every symbol include/gradjune_hip.h declares.
No compute call is made (there is no GPU in the CPU test environment)."""
import ctypes
import os
import re

import pytest

from grad_june_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gradjune_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gj_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(N.LIB_PATH):
        import importlib.util

        spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        m.build()
    return N.load()


def test_header_and_binding_agree(lib):
    decl = declared_symbols()
    assert decl, "no declarations parsed"
    assert sorted(N.SYMBOLS) == decl
    for name in decl:
        assert hasattr(lib, name), f"{name} not exported"


def test_version_and_error_strings(lib):
    assert lib.gj_version() == N.GJ_ABI_VERSION
    assert lib.gj_error_string(0) == b"ok"
    assert b"NULL" in lib.gj_error_string(-1)


def test_struct_sizes_match_the_header_layout():
    assert ctypes.sizeof(N.EdgeSet) == 72
    assert ctypes.sizeof(N.Network) == 16
    assert ctypes.sizeof(N.StepParams) == 56 + 16 * N.GJ_MAX_NETS
    assert ctypes.sizeof(N.AgentState) == 80
    assert ctypes.sizeof(N.StepIO) == 32
    assert ctypes.sizeof(N.Plan) == 16 + 16 + 72 * N.GJ_MAX_SETS + 5 * 8 + 8 + 8
    assert ctypes.sizeof(N.TiledSet) == 16 + 10 * 8
    assert ctypes.sizeof(N.Tiled) == 16 + 8 + 8 + ctypes.sizeof(N.TiledSet) * N.GJ_MAX_SETS


def test_argument_errors_do_not_need_a_gpu(lib):
    """NULL plan -> GJ_E_NULL before anything touches the device."""
    assert lib.gj_step(None, None, None, None, None) == -1
    assert lib.gj_sample_infect(-1, None, None, 0, 0, 0, 0.0, None, None, None, None, None) == -2
    assert lib.gj_pack_f32(0, None, None, None, None) == 0
