"""ops/smallops.py + csrc/smallops.hip: small element-wise torch operators recorded and run as one
interpreter launch (an opt-in experiment, see the module's docstring: measured slower than the
separate graph nodes).  Every instruction must reproduce torch's own operator BITWISE (libm functions:
1 ulp), in any mix with operators that are not recorded, with the library's own launches, across the
program capacity, in place and through broadcast / strided operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mode():
    from pyro_amd.ops import smallops
    return smallops


def _same(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape
    if a.dtype.is_floating_point:
        return torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0)) and \
            torch.equal(torch.isnan(a), torch.isnan(b))
    return torch.equal(a, b)


def _inputs(gpu):
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((8, 1024), generator=g).to(gpu)
    y = torch.randn((8, 1024), generator=g).to(gpu)
    row = torch.randn((1024,), generator=g).to(gpu)
    col = (torch.rand((8, 1), generator=g) + 0.5).to(gpu)
    s = torch.tensor(1.7, device=gpu)
    x[0, :4] = torch.tensor([float("nan"), float("inf"), -float("inf"), 0.0])
    return x, y, row, col, s


CASES = {
    "add": lambda x, y, row, col, s: x + y,
    "add_bcast": lambda x, y, row, col, s: x + row,
    "add_col": lambda x, y, row, col, s: col + x,
    "add_imm": lambda x, y, row, col, s: x + 2.5,
    "radd_imm": lambda x, y, row, col, s: 2.5 + x,
    "sub": lambda x, y, row, col, s: x - y,
    "sub_imm": lambda x, y, row, col, s: x - 0.3,
    "rsub_imm": lambda x, y, row, col, s: 1.0 - x,
    "mul": lambda x, y, row, col, s: x * y,
    "mul_0dim": lambda x, y, row, col, s: x * s,
    "mul_imm": lambda x, y, row, col, s: x * 3,
    "div": lambda x, y, row, col, s: x / y,
    "div_col": lambda x, y, row, col, s: x / col,
    "div_imm": lambda x, y, row, col, s: x / 7.0,
    "rdiv_imm": lambda x, y, row, col, s: 2.0 / x,
    "neg": lambda x, y, row, col, s: -x,
    "exp": lambda x, y, row, col, s: torch.exp(x),
    "log": lambda x, y, row, col, s: torch.log(x.abs()),
    "reciprocal": lambda x, y, row, col, s: torch.reciprocal(x),
    "sqrt": lambda x, y, row, col, s: torch.sqrt(y * y),
    "clamp": lambda x, y, row, col, s: torch.clamp(x, -0.5, 0.7),
    "clamp_min": lambda x, y, row, col, s: torch.clamp(x, min=0.1),
    "clamp_max": lambda x, y, row, col, s: x.clamp(max=0.1),
    "clone": lambda x, y, row, col, s: x.t().clone(),
    "transposed_operand": lambda x, y, row, col, s: x.t() * 2.0 + y.t(),
    "slice_operand": lambda x, y, row, col, s: x[:, ::2] + y[:, 1::2],
    "where": lambda x, y, row, col, s: torch.where(y >= 0.0, x, y),
    "where_scalar_branch": lambda x, y, row, col, s: torch.where((y > -0.2) & (y < 0.9), x, torch.zeros((), device=x.device)),
    "compare_and": lambda x, y, row, col, s: ((x >= 0.1) & (y <= 0.3)).logical_and(x < 2.0),
    "chain": lambda x, y, row, col, s: torch.exp(-(x * x) / 2.0) / (col * 2.5) + row,
    "inplace": lambda x, y, row, col, s: y.clone().mul_(2.0).add_(row).div_(col).clamp_(-3.0, 3.0),
    "four_dims": lambda x, y, row, col, s: x.reshape(2, 4, 32, 32) * y.reshape(2, 4, 32, 32)[:, :1] + 1.0,
    "scalar_frame": lambda x, y, row, col, s: s * 2.0 + s,
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_instruction_equals_torch_bitwise(gpu, case):
    so = _mode()
    f = CASES[case]
    ref = f(*_inputs(gpu))
    mode = so.SmallOps()
    with mode:
        got = f(*_inputs(gpu))
    torch.cuda.synchronize()
    assert mode.recorded >= 1 and mode.launches >= 1, (mode.recorded, mode.launches)
    if case in ("log", "exp", "chain"):
        # libm: torch's build and this library's may carry different device-library versions: 1 ulp
        torch.testing.assert_close(got, ref, rtol=2.5e-7, atol=0, equal_nan=True)
    else:
        assert _same(got, ref)


def test_mixed_with_other_operators_and_library_launches(gpu):
    """Recorded runs interleaved with operators that are not recorded (sum, matmul: the mode flushes
    first), with a launch of the library (as_stream flushes) and past the program capacity."""
    so = _mode()
    from pyro_amd import kernels as k

    def f(x, y, row, col, s):
        a = x * 2.0 + row                      # recorded
        t = a.sum(-1, keepdim=True)            # not recorded: flush, then ATen
        b = a / t                              # recorded, reads the ATen result
        for _ in range(45):                    # > PA_SMALLOPS_MAX dependent instructions
            b = b * 1.01 + 0.001
        m = b @ y.t()                          # rocBLAS reads the pending program's output
        lp = k.gamma_implicit_grad(torch.exp(b[:2, :8]) + 0.5, torch.exp(a[:2, :8] * 0.1))   # library launch
        return b, m, lp, torch.exp(lp) - 1.0   # recorded after the library launch

    ref = f(*_inputs(gpu))
    mode = so.SmallOps()
    with mode:
        got = f(*_inputs(gpu))
    torch.cuda.synchronize()
    assert mode.launches >= 4 and mode.recorded >= 95
    for g, r in zip(got, ref):
        if g.shape == (8, 8):                  # the matmul: rocBLAS is deterministic for equal inputs
            assert _same(g, r)
        else:
            assert _same(g, r)


def test_autograd_duals_are_recorded(gpu):
    """The backward operators run on autograd's thread: recorded there too, same gradients."""
    so = _mode()
    x0, y0, row, col, s = _inputs(gpu)
    x0 = torch.nan_to_num(x0, nan=0.5, posinf=1.0, neginf=-1.0)

    def f(x, w):
        z = torch.exp(x * w) / (1.0 + col)
        z = z.clamp(min=1e-3)
        return (torch.log(z) * row).sum()

    outs = []
    for use in (False, True):
        x = x0.clone().requires_grad_()
        w = row.clone().requires_grad_()
        if use:
            mode = so.SmallOps()
            with mode:
                f(x, w).backward()
            assert mode.recorded > 8
        else:
            f(x, w).backward()
        torch.cuda.synchronize()
        outs.append((x.grad.clone(), w.grad.clone()))
    for k in (0, 1):                                   # (exp / log inside: 1 ulp each, see above)
        torch.testing.assert_close(outs[1][k], outs[0][k], rtol=2e-6, atol=1e-6 * float(outs[0][k].abs().max()))
