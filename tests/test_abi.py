"""CPU-side checks of the drop-in boundary: libmrs_swarm.so builds for gfx950, loads, exports every symbol that
include/mrs_swarm.h declares, and its host-only helpers agree with the oracle.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from helpers import oracle_params, to_product_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mrs_swarm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mrs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(mrs):
    from mrs_multirotor_simulator_amd import swarm
    L = C.CDLL(swarm.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 38
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/mrs_swarm.h but not exported"
    assert sorted(swarm.ABI_SYMBOLS) == declared


def test_code_object_is_gfx950(mrs):
    from mrs_multirotor_simulator_amd import swarm
    blob = open(swarm.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"mrs_uav_step_literal" in blob and b"mrs_uav_step_fast" in blob


def test_param_structs_match_oracle_layout(mrs, oracle):
    assert C.sizeof(mrs.ModelParams) == C.sizeof(oracle.ModelParams) == 16 + 12 * 8 + 9 * 8 + 32 * 8
    d_m, d_o = mrs.default_params(), oracle.default_params()
    assert bytes(d_m) == bytes(d_o)


@pytest.mark.parametrize("name", ["x500", "a300", "f330", "f450", "f550", "naki", "robofly", "t650"])
def test_airframe_params_agree_with_oracle(mrs, oracle, name):
    """mrs_calculate_inertia / mrs_scale_allocation (product, host) vs the oracle's restatement: bit-identical."""
    assert bytes(mrs.model_params(name)) == bytes(oracle_params(name))
    p = to_product_params(mrs, oracle_params(name))
    assert p.n_motors == mrs.AIRFRAMES[name]["n_motors"]


def test_no_gpu_means_loud_failure(mrs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mrs.MrsError, match="no HIP device|no CPU fallback"):
        mrs.Swarm(8)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mrs_multirotor_simulator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("oracle's", "").replace("oracle/", "ORACLE_DOC/") or "import oracle" not in txt
                assert "liboracle" not in txt and "uav_oracle" not in txt and "from oracle" not in txt


def test_bench_reaches_the_oracle_only_in_its_cpu_baseline_leg():
    """bench.py may use oracle/ for the `cpu_baseline` leg alone: importing the module and generating the synthetic inputs of every
    workload must leave `oracle` (and tests/helpers, which imports it) unloaded, and the only functions of bench.py that mention
    them are cpu_baseline and its helpers."""
    import subprocess
    code = ("import sys; sys.argv=['bench.py']; sys.path.insert(0, %r); import bench\n"
            "for w in ('actuator', 'position', 'position+collisions'): bench.make_inputs(256, w, 3)\n"
            "bad = [m for m in sys.modules if m == 'helpers' or m == 'oracle' or m.startswith('oracle.')]\n"
            "assert not bad, bad\nprint('clean')") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "clean" in out.stdout, out.stderr[-1500:]
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        src = ast.get_source_segment(open(os.path.join(ROOT, "bench.py")).read(), fn)
        if "from oracle" in src or "import helpers" in src:
            assert fn.name == "cpu_baseline", f"{fn.name} reaches for the oracle"
