"""CPU ORACLE (test infrastructure, NOT product code): ctypes binding + orchestration of ``oracle/deepset_oracle.c``, the
C + OpenMP restatement of DeepSetFeat + E_score (pooling.py:604-673, :282; base_modules.py:38-48), forward and backward in
train mode, running on the host cores.

Only tests/ and bench.py's cpu_baseline leg may import this module."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle_deepset.so")
_lib = None
D = 32


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB_PATH)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
        L.oracle_deepset_num_threads.restype = ctypes.c_int
        for name, args in (("oracle_block_fwd", [vp, i64, i32, vp, vp, vp, f32, vp, vp, vp, vp]),
                           ("oracle_block_bwd", [vp, vp, vp, i64, i32] + [vp] * 9),
                           ("oracle_segmax_fwd", [vp, vp, i64, vp, vp]),
                           ("oracle_segmax_bwd", [vp, vp, i64, vp]),
                           ("oracle_concat_fwd", [vp, vp, vp, i64, vp]),
                           ("oracle_concat_bwd", [vp, vp, i64, vp, vp]),
                           ("oracle_score_fwd", [vp, i64, i32, vp, vp, vp]),
                           ("oracle_score_bwd", [vp, vp, i64, i32, vp, vp, vp, vp]),
                           ("oracle_wblock_fwd", [vp, vp, i64, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp]),
                           ("oracle_wblock_bwd", [vp, vp, vp, vp, i64, i32, i32] + [vp] * 9),
                           ("oracle_fusion_concat_fwd", [vp, vp, i64, i32, i32, vp]),
                           ("oracle_fusion_concat_bwd", [vp, i64, i32, i32, vp, vp])):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = args
        _lib = L
    return _lib


def num_threads():
    return int(lib().oracle_deepset_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


BLOCKS = ("mlp_elt_1.0", "mlp_elt_1.1", "mlp_set.0", "mlp_set.1", "mlp_elt_2.0", "mlp_elt_2.1")


def params_from_state_dict(sd, lin_weight, lin_bias):
    """{block: (W, gamma, beta)} + ('Ws', 'bs') from the state dict of a (reference-layout) DeepSetFeat and the score
    Linear."""
    P = {}
    for b in BLOCKS:
        P[b] = (_f(sd[b + ".0.weight"]), _f(sd[b + ".1.batch_norm.weight"]), _f(sd[b + ".1.batch_norm.bias"]))
    P["Ws"], P["bs"] = _f(lin_weight), _f(lin_bias)
    return P


def _block_fwd(x, Wgb, eps=1e-5):
    W, g, b = Wgb
    M, K = x.shape
    z, a = np.empty((M, D), np.float32), np.empty((M, D), np.float32)
    mean, inv = np.empty(D, np.float32), np.empty(D, np.float32)
    lib().oracle_block_fwd(_p(x), M, K, _p(W), _p(g), _p(b), eps, _p(z), _p(a), _p(mean), _p(inv))
    return a, (x, z, mean, inv)


def _block_bwd(da, Wgb, saved, need_dx=True):
    W, g, b = Wgb
    x, z, mean, inv = saved
    M, K = x.shape
    dx = np.empty((M, K), np.float32) if need_dx else None
    dW, dg, db = np.empty((D, K), np.float32), np.empty(D, np.float32), np.empty(D, np.float32)
    lib().oracle_block_bwd(_p(x), _p(z), _p(da), M, K, _p(W), _p(g), _p(b), _p(mean), _p(inv), _p(dx), _p(dW), _p(dg),
                           _p(db))
    return dx, (dW, dg, db)


def forward(P, x_map, csr, use_num=True):
    """scores [V, G] and the cache of the backward."""
    L = lib()
    x_map, csr = _f(x_map), np.ascontiguousarray(csr, dtype=np.int64)
    V, N, G = x_map.shape[0], csr.shape[0] - 1, P["Ws"].shape[0]
    a1, s1 = _block_fwd(x_map, P["mlp_elt_1.0"])
    a2, s2 = _block_fwd(a1, P["mlp_elt_1.1"])
    pooled, arg = np.empty((N, D), np.float32), np.empty((N, D), np.int64)
    L.oracle_segmax_fwd(_p(a2), _p(csr), N, _p(pooled), _p(arg))
    if use_num:
        num = np.sqrt(1.0 / ((csr[1:] - csr[:-1]).astype(np.float32) + np.float32(1e-3))).astype(np.float32)
        x_set = np.ascontiguousarray(np.concatenate([pooled, num[:, None]], 1))
    else:
        x_set = pooled
    b1, s3 = _block_fwd(x_set, P["mlp_set.0"])
    b2, s4 = _block_fwd(b1, P["mlp_set.1"])
    cat = np.empty((V, 2 * D), np.float32)
    L.oracle_concat_fwd(_p(a2), _p(b2), _p(csr), N, _p(cat))
    a5, s5 = _block_fwd(cat, P["mlp_elt_2.0"])
    a6, s6 = _block_fwd(a5, P["mlp_elt_2.1"])
    scores = np.empty((V, G), np.float32)
    L.oracle_score_fwd(_p(a6), V, G, _p(P["Ws"]), _p(P["bs"]), _p(scores))
    return scores, dict(csr=csr, arg=arg, a6=a6, saved=(s1, s2, s3, s4, s5, s6), use_num=use_num)


def backward(P, cache, dscores):
    """Gradients {block: (dW, dgamma, dbeta)}, 'Ws', 'bs' (x_map gets none, as in the product path)."""
    L = lib()
    csr, arg, a6 = cache["csr"], cache["arg"], cache["a6"]
    s1, s2, s3, s4, s5, s6 = cache["saved"]
    V, N, G = a6.shape[0], csr.shape[0] - 1, P["Ws"].shape[0]
    dc = _f(dscores)
    out = {}
    da6 = np.empty((V, D), np.float32)
    dWs, dbs = np.empty((G, D), np.float32), np.empty(G, np.float32)
    L.oracle_score_bwd(_p(a6), _p(dc), V, G, _p(P["Ws"]), _p(da6), _p(dWs), _p(dbs))
    out["Ws"], out["bs"] = dWs, dbs
    da5, out["mlp_elt_2.1"] = _block_bwd(da6, P["mlp_elt_2.1"], s6)
    dcat, out["mlp_elt_2.0"] = _block_bwd(da5, P["mlp_elt_2.0"], s5)
    da2, db2 = np.empty((V, D), np.float32), np.empty((N, D), np.float32)
    L.oracle_concat_bwd(_p(dcat), _p(csr), N, _p(da2), _p(db2))
    db1, out["mlp_set.1"] = _block_bwd(db2, P["mlp_set.1"], s4)
    dxset, out["mlp_set.0"] = _block_bwd(db1, P["mlp_set.0"], s3)
    dpooled = np.ascontiguousarray(dxset[:, :D])
    L.oracle_segmax_bwd(_p(dpooled), _p(arg), N, _p(da2))
    da1, out["mlp_elt_1.1"] = _block_bwd(da2, P["mlp_elt_1.1"], s2)
    _, out["mlp_elt_1.0"] = _block_bwd(da1, P["mlp_elt_1.0"], s1, need_dx=False)
    return out


# ---- E_mod on the map rows + fusion concat (the rest of the pooling step around DeepSetFeat + attention)
def emod_params_from_state_dict(sd):
    """[(W, gamma, beta)] * 2 from the state dict of E_mod = MLP([in_mod, out_mod, out_mod], bias=False)."""
    return [(_f(sd[f"{i}.0.weight"]), _f(sd[f"{i}.1.batch_norm.weight"]), _f(sd[f"{i}.1.batch_norm.bias"]))
            for i in range(2)]


def emod_forward(P_mod, rows, counts, eps=1e-5):
    """E_mod(rows) on the R map rows, train-mode batch statistics weighted by ``counts`` (views per row)."""
    x, cnt = _f(rows), _f(counts)
    saved = []
    for W, g, b in P_mod:
        M, K, O = x.shape[0], x.shape[1], W.shape[0]
        z, a = np.empty((M, O), np.float32), np.empty((M, O), np.float32)
        mean, inv = np.empty(O, np.float32), np.empty(O, np.float32)
        lib().oracle_wblock_fwd(_p(x), _p(cnt), M, K, O, _p(W), _p(g), _p(b), eps, _p(z), _p(a), _p(mean), _p(inv))
        saved.append((x, z, mean, inv))
        x = a
    return x, dict(saved=saved, cnt=cnt)


def emod_backward(P_mod, cache, drows_out, need_dx=True):
    """Gradients [(dW, dgamma, dbeta)] * 2 and d(rows) [R, C_in] from the gradient of E_mod's output rows."""
    da, cnt, grads = _f(drows_out), cache["cnt"], [None, None]
    for i in (1, 0):
        W, g, b = P_mod[i]
        x, z, mean, inv = cache["saved"][i]
        M, K, O = x.shape[0], x.shape[1], W.shape[0]
        dx = np.empty((M, K), np.float32) if (i > 0 or need_dx) else None
        dW, dg, db = np.empty((O, K), np.float32), np.empty(O, np.float32), np.empty(O, np.float32)
        lib().oracle_wblock_bwd(_p(x), _p(z), _p(da), _p(cnt), M, K, O, _p(W), _p(g), _p(b), _p(mean), _p(inv), _p(dx),
                                _p(dW), _p(dg), _p(db))
        grads[i] = (dW, dg, db)
        da = dx
    return grads, da


def fusion_concat_forward(x3d, xpool):
    x3d, xpool = _f(x3d), _f(xpool)
    N, A, C = x3d.shape[0], x3d.shape[1], xpool.shape[1]
    out = np.empty((N, A + C), np.float32)
    lib().oracle_fusion_concat_fwd(_p(x3d), _p(xpool), N, A, C, _p(out))
    return out


def fusion_concat_backward(dout, A):
    dout = _f(dout)
    N, C = dout.shape[0], dout.shape[1] - A
    dx3d, dxpool = np.empty((N, A), np.float32), np.empty((N, C), np.float32)
    lib().oracle_fusion_concat_bwd(_p(dout), N, A, C, _p(dx3d), _p(dxpool))
    return dx3d, dxpool
