"""Vindex (reference: pyro/ops/indexing.py:9-180): the documented identity
    Vindex(x)[..., i, :, j][b..., e] == x[b..., i[b...], e, j[b...]]
checked element by element, plus the plain-indexing fall-backs."""
import itertools

import pytest
import torch

from pyro_amd.ops.indexing import Vindex, vindex


def test_batched_tensor_and_indices_follow_the_documented_identity():
    torch.manual_seed(0)
    x = torch.randn(4, 3, 5, 6)                      # batch [4], event [3, 5, 6]
    i = torch.randint(0, 3, (2, 1))                  # broadcasts against the batch dim
    j = torch.randint(0, 6, (7, 1, 1))
    out = Vindex(x)[..., i, :, j]
    assert out.shape == (7, 2, 4, 5)
    for a, b, c, e in itertools.product(range(7), range(2), range(4), range(5)):
        assert out[a, b, c, e] == x[c, i[b, 0], e, j[a, 0, 0]]


def test_unbatched_tensor_trailing_dims_are_kept():
    x = torch.arange(24.0).reshape(2, 3, 4)
    i = torch.tensor([[1], [0]])
    out = Vindex(x)[i]                               # == x[i] with dims 1.. kept as event dims
    assert out.shape == (2, 1, 3, 4) and torch.equal(out[0, 0], x[1])
    out = Vindex(x)[i, :, torch.tensor([3, 0, 2])]
    assert out.shape == (2, 3, 3)
    assert out[1, 2, 1] == x[0, 1, 2]


def test_plain_indexing_when_no_index_has_dims():
    x = torch.randn(3, 4, 5)
    assert torch.equal(Vindex(x)[..., 2], x[..., 2])
    assert torch.equal(Vindex(x)[1, :, torch.tensor(3)], x[1, :, 3])
    assert torch.equal(vindex(x, 1), x[1]) and Vindex(x)[()] is x and Vindex(x)[...,] is x


def test_unsupported_forms_raise():
    x = torch.randn(3, 4, 5)
    i = torch.tensor([0, 1])
    with pytest.raises(NotImplementedError):
        Vindex(x)[i, ..., 0]
    with pytest.raises(NotImplementedError):
        Vindex(x)[i, 1:3]


def test_index_flattens_nested_index_tuples():
    """pyro.ops.indexing.Index (tests/ops/test_indexing.py): ``t`` may itself be an index tuple."""
    import torch
    from pyro_amd.ops.indexing import Index, index
    x = torch.arange(12.0).reshape(3, 4)
    assert torch.equal(Index(x)[..., 1], x[..., 1])
    assert torch.equal(Index(x)[..., slice(None)], x)
    assert torch.equal(Index(x)[..., (Ellipsis, None)], x.unsqueeze(-1))
    assert torch.equal(index(x, ((1,), (Ellipsis,))), x[1])
    assert index(x, ()) is x
