"""CPU tests of the host-side CSR containers (pure index logic) against the reference's golden vectors
and the invariants of its debug() methods (csr.py:81-108)."""
import numpy as np
import torch

from conftest import load_golden, t
from deepviewagg_amd.core.multimodal.csr import CSRData, CSRBatch
from deepviewagg_amd.utils.multimodal import tensor_idx


def test_pointers_select_insert_match_reference():
    g = load_golden("lex_csr")
    csr = CSRData(t(g["csr_idx"]), t(g["csr_vals"]), dense=True)
    assert torch.equal(csr.pointers, t(g["csr_pointers"]))
    csr.debug()
    sub = csr[t(g["sel"])]
    assert torch.equal(sub.pointers, t(g["sel_pointers"])) and torch.equal(sub.values[0], t(g["sel_vals"]))
    ins = CSRData(t(g["csr_idx"]), t(g["csr_vals"]), dense=True).insert_empty_groups(t(g["groups"]), num_groups=25)
    assert torch.equal(ins.pointers, t(g["ins_pointers"]))
    ins.debug()
    assert csr[[]].num_groups == 0 and csr[2].num_groups == 1
    assert torch.equal(csr[torch.tensor([True] + [False] * (csr.num_groups - 1))].pointers, csr[0].pointers)


def test_batch_round_trip_matches_reference():
    g = load_golden("lex_csr")
    items = [CSRData(t(g[f"b{i}_pointers"]), t(g[f"b{i}_v0"]), t(g[f"b{i}_v1"]), is_index_value=[True, False])
             for i in range(3)]
    batch = CSRBatch.from_csr_list(items)
    assert torch.equal(batch.pointers, t(g["batch_pointers"]))
    assert torch.equal(batch.values[0], t(g["batch_v0"])) and torch.equal(batch.values[1], t(g["batch_v1"]))
    assert torch.equal(batch.__sizes__, t(g["batch_sizes"]))
    back = batch.to_csr_list()
    for a, b in zip(items, back):
        assert torch.equal(a.pointers, b.pointers)
        assert torch.equal(a.values[0], b.values[0]) and torch.equal(a.values[1], b.values[1])
    assert type(batch[[0, 1]]) is CSRData


def test_nested_csr_batching():
    gen = torch.Generator().manual_seed(0)
    items = []
    for _ in range(3):
        outer = torch.sort(torch.randint(0, 4, (9,), generator=gen))[0]
        inner = torch.sort(torch.randint(0, 9, (30,), generator=gen))[0]
        inner = torch.cat([torch.arange(9), inner]).sort()[0]      # every outer item owns >= 1 inner item
        nested = CSRData(inner, torch.randn(inner.shape[0], generator=gen), dense=True)
        items.append(CSRData(outer, torch.randn(9, generator=gen), nested, dense=True))
    batch = CSRBatch.from_csr_list(items)
    batch.debug()
    for a, b in zip(items, batch.to_csr_list()):
        assert torch.equal(a.pointers, b.pointers) and torch.equal(a.values[0], b.values[0])
        assert torch.equal(a.values[1].pointers, b.values[1].pointers)
        assert torch.equal(a.values[1].values[0], b.values[1].values[0])


def test_tensor_idx():
    assert tensor_idx(3).tolist() == [3] and tensor_idx([1, 2]).tolist() == [1, 2]
    assert tensor_idx(slice(1, 4)).tolist() == [1, 2, 3]
    assert tensor_idx(np.array([0, 2])).tolist() == [0, 2]
    assert tensor_idx(torch.tensor([True, False, True])).tolist() == [0, 2]
    assert tensor_idx(None).shape == (0,)
