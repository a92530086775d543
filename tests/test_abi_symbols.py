"""CPU test: libdva_hip.so loads (no GPU needed) and exports every entry point include/dva.h declares;
the ctypes table in deepviewagg_amd/_lib.py covers exactly the same set.  No compute calls here."""
import ctypes
import os
import re

from conftest import ROOT
from deepviewagg_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dva.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dva_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree():
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in dva.h but not exported: {missing}"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)


def test_library_loads_and_reports_version_without_gpu():
    lib = _lib.load()
    assert lib.dva_version() >= 100
    assert lib.dva_device_count() >= 0


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any HIP call (error codes, no exceptions across the ABI)."""
    lib = _lib.load()
    assert lib.dva_segment_csr_fwd(None, None, None, None, 4, 3, 0, 0, None) == -1      # null ptr
    assert lib.dva_segment_csr_fwd(None, None, None, None, -1, 3, 0, 0, None) == -1     # negative size
    assert lib.dva_deepset_fwd_first(None, None, None, None, None, None, 0, 5, 1, 0, 0, None) == -1
    assert lib.dva_deepset_fwd_layer(None, None, None, None, None, None, None, 0, 0, 7, None) == -1   # bad act_dtype
    assert lib.dva_row_plan(None, -1, 4, None, None, None, None, 0, None) == -1
    assert lib.dva_row_plan_workspace_bytes(1 << 33, 4) == -2
    assert lib.dva_pack_gather_index(None, None, None, 2, 0.5, 1, 1, None, None) == -1  # ratio < 1


def test_camera_struct_layout_matches_oracle():
    """struct dva_camera (product) and struct dvo_camera (oracle) must stay byte-compatible."""
    from oracle import mapping_oracle as M
    assert ctypes.sizeof(_lib.DvaCamera) == ctypes.sizeof(M.Camera)
    assert [f[0] for f in _lib.DvaCamera._fields_] == [f[0] for f in M.Camera._fields_]


def _prototypes():
    """name -> (return type, [parameter types]) parsed from include/dva.h."""
    text = open(os.path.join(ROOT, "include", "dva.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int64_t|int)\s+(dva_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, params = m.group(1), m.group(2), m.group(3).strip()
        ptypes = []
        if params and params != "void":
            for p in params.split(","):
                p = " ".join(p.split())
                ptypes.append("ptr" if "*" in p else re.sub(r"\s+\w+$", "", p).replace("const ", ""))
        out[name] = (ret, ptypes)
    return out


def test_ctypes_signatures_match_the_header_prototypes():
    """Every ctypes prototype has the arity and the scalar / pointer kinds of the C declaration: a mismatch
    would corrupt the call silently (ctypes does not check)."""
    kind = {ctypes.c_void_p: "ptr", ctypes.c_int64: "int64_t", ctypes.c_int32: "int32_t",
            ctypes.c_float: "float", ctypes.c_double: "double"}       # c_int is c_int32 here
    norm = lambda t: "int32_t" if t == "int" else t
    protos = _prototypes()
    assert set(protos) == set(_lib.SIGNATURES)
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        ret, ptypes = protos[name]
        assert kind[restype] == norm(ret), (name, ret, restype)
        got = ["ptr" if (isinstance(a, type) and issubclass(a, ctypes._Pointer)) else kind[a] for a in argtypes]
        want = [norm(t) for t in ptypes]
        assert got == want, (name, got, want)
