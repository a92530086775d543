"""CPU ORACLE (test infrastructure, NOT product code): ctypes binding of ``oracle/attention_oracle.c``, the C + OpenMP
restatement of the view gather + attention pooling tail (pooling.py:284-300, :690-715, :737-810; image.py:1262-1287),
forward and backward, running on the host cores.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle_attention.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB_PATH)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_gather_attention_fwd.restype = None
        L.oracle_gather_attention_fwd.argtypes = [vp] * 6 + [i64, i64, i32, i32, i32, f32] + [vp] * 4
        L.oracle_gather_attention_bwd.restype = None
        L.oracle_gather_attention_bwd.argtypes = [vp] * 10 + [i64, i64, i64, i32, i32, i32, f32] + [vp] * 4
        _lib = L
    return _lib


def num_threads():
    return int(lib().oracle_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def forward(rows, row_idx, compat, csr, gate_w=None, gate_b=None, scaling=True, eps=1e-12):
    """numpy in / out: (out [N, C], att [V, G], gate [N, G], amax [N, G])."""
    rows, compat = _f32(rows), _f32(compat)
    row_idx = np.ascontiguousarray(row_idx, dtype=np.int32)
    csr = np.ascontiguousarray(csr, dtype=np.int64)
    gate_w, gate_b = _f32(gate_w), _f32(gate_b)
    N, (V, G), C = csr.shape[0] - 1, compat.shape, rows.shape[1]
    assert G <= 64
    out = np.empty((N, C), np.float32)
    att = np.zeros((V, G), np.float32)
    gate = np.empty((N, G), np.float32)
    amax = np.empty((N, G), np.int32)
    lib().oracle_gather_attention_fwd(_p(rows), _p(row_idx), _p(compat), _p(csr), _p(gate_w), _p(gate_b), N, V, C, G,
                                      int(scaling), float(eps), _p(out), _p(att), _p(gate), _p(amax))
    return out, att, gate, amax


def backward(grad_out, rows, row_idx, compat, csr, att, gate, amax, gate_w=None, gate_b=None, scaling=True,
             eps=1e-12):
    """(grad_rows [R, C], grad_compat [V, G], grad_gate_w [G], grad_gate_b [G])."""
    grad_out, rows, compat = _f32(grad_out), _f32(rows), _f32(compat)
    row_idx = np.ascontiguousarray(row_idx, dtype=np.int32)
    csr = np.ascontiguousarray(csr, dtype=np.int64)
    gate_w, gate_b = _f32(gate_w), _f32(gate_b)
    N, (V, G), (R, C) = csr.shape[0] - 1, compat.shape, rows.shape
    g_rows = np.empty((R, C), np.float32)
    g_compat = np.empty((V, G), np.float32)
    g_w, g_b = np.zeros(G, np.float32), np.zeros(G, np.float32)
    lib().oracle_gather_attention_bwd(_p(grad_out), _p(rows), _p(row_idx), _p(compat), _p(csr), _p(gate_w), _p(gate_b),
                                      _p(att), _p(gate), _p(amax), N, V, R, C, G, int(scaling), float(eps),
                                      _p(g_rows), _p(g_compat), _p(g_w), _p(g_b))
    return g_rows, g_compat, g_w, g_b
