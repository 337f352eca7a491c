"""CPU tier: the C-ABI library loads, exports every symbol include/*.h declares, and refuses to compute without a GPU
(no silent CPU fallback).  No compute calls here."""
import ctypes
import glob
import os
import re

import pytest

from helpers import ROOT


def declared_symbols():
    syms = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        syms += re.findall(r"\b((?:cmdr|sharp)_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(s for s in syms if not s.endswith("_fn")))


def test_library_exports_every_declared_symbol():
    from commander_amd import get_lib
    L = get_lib()
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing
    assert len(declared_symbols()) >= 30


def test_no_cpu_fallback_without_gpu():
    from commander_amd import get_lib, device_count, CmdrError
    from commander_amd.cr import CRContext
    from commander_amd import ShtPlan
    if device_count() > 0:
        pytest.skip("GPU present")
    L = get_lib()
    h = ctypes.c_void_p()
    assert L.cmdr_ctx_create(0, ctypes.byref(h)) < 0
    assert b"no CPU path" in L.cmdr_last_error()
    with pytest.raises(CmdrError):
        CRContext(0)
    with pytest.raises(CmdrError):
        ShtPlan(4, 8)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under commander_amd/ may import or link it."""
    for path in glob.glob(os.path.join(ROOT, "commander_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
            txt = open(path).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), path
            if not path.endswith(".py"):
                assert "sht_oracle" not in txt and "oracle/" not in txt.replace("tests/host_emul", ""), path


def test_hand_written_dpp_fmas_keep_their_hazard_distance():
    """The synthesis and the ninth-map accumulation of the adjoint use v_fmac_f64_dpp through inline asm, which the
    compiler's hazard recogniser cannot pad: tools/check_dpp_hazards.py compiles kernels.hip to assembly and checks
    that no DPP operand was written by a VALU instruction within the two instructions before it."""
    import os
    import subprocess
    import sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_dpp_hazards.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 hazards" in r.stdout and not r.stdout.startswith("0 DPP"), r.stdout


def test_missing_runtime_libraries_fail_cleanly(tmp_path):
    """librccl / libhdf5 are bound with dlopen at first use.  When the library cannot be loaded the entry points must
    return < 0 with a message in cmdr_last_error (bench.py's all-ranks fall-back and the chain writer's -1 depend on
    it), not crash: CMDR_RCCL_LIB / CMDR_HDF5_LIB name THE library to load, so a wrong path hides every candidate.
    Fresh process (the binding is cached per process); host emulation of the library, same C ABI."""
    import subprocess
    import sys
    from helpers import emul_lib
    emul_lib()                                    # make sure the emulation library is built
    code = r'''
import ctypes, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from helpers import emul_lib
L = emul_lib()
rc = L.cmdr_rccl_version()
assert rc < 0, rc
msg = L.cmdr_last_error()
assert msg and b"librccl" in msg, msg
buf = ctypes.create_string_buffer(128)
assert L.cmdr_rccl_unique_id(buf) < 0 and L.cmdr_last_error()
a = np.zeros(9)
dp = lambda v: v.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
rc = L.cmdr_chain_write_comp(os.path.join(%r, "c.h5").encode(), 1, b"cmb", dp(a), 2, 1, None, None, None)
assert rc < 0, rc
msg = L.cmdr_last_error()
assert msg and b"libhdf5" in msg, msg
print("clean")
''' % (ROOT, ROOT, str(tmp_path))
    env = dict(os.environ, CMDR_RCCL_LIB="/nonexistent/librccl.so", CMDR_HDF5_LIB="/nonexistent/libhdf5.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "clean" in r.stdout, r.stdout + r.stderr
