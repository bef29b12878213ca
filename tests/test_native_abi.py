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
    assert ctypes.sizeof(N.StepParams) == 56 + 16 * N.GJ_MAX_NETS + 8
    assert ctypes.sizeof(N.AgentState) == 80
    assert ctypes.sizeof(N.StepIO) == 40
    assert ctypes.sizeof(N.Plan) == 16 + 16 + 72 * N.GJ_MAX_SETS + 5 * 8 + 8 + 8
    assert ctypes.sizeof(N.TiledSet) == 16 + 11 * 8 + 5 * 8 + 8 + 8 + 8 + 8  # (+ multi_slots, max_venue_edges and its pad: ABI 6)
    assert ctypes.sizeof(N.Tiled) == 16 + 8 + 8 + 8 + ctypes.sizeof(N.TiledSet) * N.GJ_MAX_SETS


def test_ctypes_layout_equals_what_a_c_compiler_sees(tmp_path):
    """Every struct of include/gradjune_hip.h: sizeof and the offset of every field, as gcc lays them out,
    against the ctypes mirror in _native.py (a silent mismatch would shift every pointer behind it)."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    pairs = {"gj_edge_set": N.EdgeSet, "gj_tiled_set": N.TiledSet, "gj_tiled": N.Tiled, "gj_plan": N.Plan,
             "gj_network": N.Network, "gj_step_params": N.StepParams, "gj_agent_state": N.AgentState,
             "gj_step_io": N.StepIO, "gj_symptoms_params": N.SymptomsParams, "gj_compile_set": N.CompileSet,
             "gj_compile_out": N.CompileOut}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for cname, ct in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for field, _ in ct._fields_:
            lines.append(f'  printf("{cname}.{field} %zu\\n", offsetof({cname}, {field}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, ct in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(ct), cname
        for field, _ in ct._fields_:
            assert int(got[f"{cname}.{field}"]) == getattr(ct, field).offset, f"{cname}.{field}"


def test_argument_errors_do_not_need_a_gpu(lib):
    """NULL plan -> GJ_E_NULL before anything touches the device."""
    assert lib.gj_step(None, None, None, None, None) == -1
    assert lib.gj_sample_infect(-1, None, None, 0, 0, 0, 0.0, None, None, None, None, None) == -2
    assert lib.gj_pack_f32(0, None, None, None, None) == 0
    # graph compile: argument checks come before any device work
    assert lib.gj_compile_capacity(None, None, None, None) == -1
    assert lib.gj_compile_multi_slots(None, None, 0, 0, None, 0, None, None, None, 0, None) == -1
    cs = N.CompileSet(None, None, None, 0, 0, 0, 0, 1, 64, 16, 16, 0)
    cap = [ctypes.c_int64(0) for _ in range(3)]
    assert lib.gj_compile_capacity(ctypes.byref(cs), *[ctypes.byref(c) for c in cap]) == 0
    assert cap[0].value >= 1 and cap[2].value >= 1
    cs.slice_agents = 70000                                   # local agent indices are 16-bit
    assert lib.gj_compile_capacity(ctypes.byref(cs), None, None, None) == -2
    cs.slice_agents, cs.n_edges = 64, 5                       # edges without edge lists
    assert lib.gj_compile_blocks(ctypes.byref(cs), None, 1, None, None, 0, None) == -1


def test_building_the_library_does_not_load_the_oracle():
    """The oracle is test infrastructure: ``__graft_entry__.build()`` - which bench.py calls before its timed regions -
    must neither import it nor make ``oracle/`` / ``tests/`` importable (only smoke() and cpu_baseline() do)."""
    import subprocess
    import sys

    code = ("import sys, __graft_entry__ as g; g.build(); "
            "assert 'gj_oracle' not in sys.modules and 'gj_testlib' not in sys.modules; "
            "assert not any(p.rstrip('/').endswith(('/oracle', '/tests')) for p in sys.path), sys.path; print('clean')")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "clean" in r.stdout, r.stderr[-2000:]
