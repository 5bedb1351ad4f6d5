"""The row-product kernels (ctn_rows_gemm / ctn_rows_wgrad through speech_amd.ops: the dual-path blocks' input
projections, linear layers and their weight / input gradients, src/models/dprnn.py:60-64,104-116 of the reference)
against float64 products of the same tensors: the recipe's shapes, shapes that do not fill a tile in any dimension, row
views with a stride and a column offset (as the recurrent weight gradient takes them), bases that are not 16-byte
aligned, and the subtracting form."""
import pytest
import torch

from conftest import assert_close
from speech_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5

SHAPES = [(1000, 64, 1024), (4100, 256, 64), (777, 1024, 64), (130, 12, 400), (5, 3, 7), (1, 64, 64), (257, 100, 33),
          (33, 128, 512), (5000, 128, 512), (4099, 64, 1024), (70000, 64, 512), (16, 128, 512)]


@pytest.mark.parametrize("R,K,M", SHAPES)
@pytest.mark.parametrize("w_is_km", [False, True])
@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
def test_rows_gemm_matches_float64(R, K, M, w_is_km, arith):
    from speech_amd import ops, _lib
    arith = {"f32": _lib.ARITH_F32, "bf16x3": _lib.ARITH_BF16X3}[arith]
    x = synth.normal((R, K), 1)
    w = synth.normal((K, M) if w_is_km else (M, K), 2)
    b = synth.normal((M,), 3)
    ref = x.double() @ (w.double() if w_is_km else w.double().t()) + b.double()
    y = ops.rows_gemm(x.to(DEV), w.to(DEV), b.to(DEV), w_is_km=w_is_km, arith=arith)
    torch.cuda.synchronize()
    assert_close(y, ref, TOL, "y")
    y0 = ops.rows_gemm(x.to(DEV), w.to(DEV), None, w_is_km=w_is_km, arith=arith)
    assert_close(y0, ref - b.double(), TOL, "y without bias")


@pytest.mark.parametrize("R,K,M", SHAPES)
@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
def test_rows_wgrad_matches_float64(R, K, M, arith):
    from speech_amd import ops, _lib
    arith = {"f32": _lib.ARITH_F32, "bf16x3": _lib.ARITH_BF16X3}[arith]
    dy = synth.normal((R, M), 4)
    x = synth.normal((R, K), 5)
    ref = dy.double().t() @ x.double()
    db = torch.zeros(M, device=DEV)
    dW = ops.rows_wgrad(dy.to(DEV), x.to(DEV), dbias=db, arith=arith)
    torch.cuda.synchronize()
    assert_close(dW, ref, TOL, "dW")
    assert_close(db, dy.double().sum(0), TOL, "dbias")
    # the subtracting form takes the same product out again: what is left is rounding of the atomics' order
    ops.rows_wgrad(dy.to(DEV), x.to(DEV), dW=dW, sign=-1.0, arith=arith)
    torch.cuda.synchronize()
    assert float(dW.abs().max()) <= 1e-5 * float(ref.abs().max())


def test_rows_products_on_strided_and_unaligned_views():
    """Views as LstmFn.backward passes them (a direction's columns of a (rows, directions, features) tensor, shifted by
    one row; one step of every sequence) and operands whose base is 4 bytes off a 16-byte boundary."""
    from speech_amd import ops
    N, T, D, H = 7, 9, 2, 24
    dg = synth.normal((N, T, D, 4 * H), 6).to(DEV)
    h = synth.normal((N, T, D, H), 7).to(DEV)
    for d in range(D):
        a, b = dg.view(N * T, D, 4 * H)[:, d][1:], h.view(N * T, D, H)[:, d][:-1]
        assert_close(ops.rows_wgrad(a, b), a.double().t() @ b.double(), TOL, "shifted product, direction %d" % d)
        a, b = dg[1:, 0, d], h[:-1, T - 1, d]
        assert_close(ops.rows_wgrad(a, b), a.double().t() @ b.double(), TOL, "boundary rows, direction %d" % d)
    flat = synth.normal((40 * 20 + 1,), 8).to(DEV)
    x = flat[1:].view(40, 20)                       # base 4 bytes off
    w = synth.normal((12 * 20 + 3,), 9).to(DEV)[3:].view(12, 20)
    assert x.data_ptr() % 16 != 0
    assert_close(ops.rows_gemm(x, w), x.double() @ w.double().t(), TOL, "unaligned gemm")
    assert_close(ops.rows_wgrad(x, x), x.double().t() @ x.double(), TOL, "unaligned wgrad")


def test_whole_output_weight_gradient_on_strided_views():
    """The whole-output three-piece kernel (M a multiple of 512, K 64 or 128) on the views LstmFn.backward passes: one
    direction's columns of (rows, directions, 4H) gate gradients, shifted by a row, with the subtracting form."""
    from speech_amd import ops, _lib
    N, T, D, H = 37, 61, 2, 128
    dg = synth.normal((N, T, D, 4 * H), 16).to(DEV)
    h = synth.normal((N, T, D, H), 17).to(DEV)
    x = synth.normal((N * T, 64), 18).to(DEV)
    for d in range(D):
        a, b = dg.view(N * T, D, 4 * H)[:, d][1:], h.view(N * T, D, H)[:, d][:-1]
        ref = a.double().t() @ b.double()
        dW = ops.rows_wgrad(a, b, arith=_lib.ARITH_BF16X3)
        assert_close(dW, ref, TOL, "recurrent weights, direction %d" % d)
        ops.rows_wgrad(a, b, dW=dW, sign=-1.0, arith=_lib.ARITH_BF16X3)
        assert float(dW.abs().max()) <= 1e-5 * float(ref.abs().max())
    a = dg.view(N * T, D * 4 * H)
    db = torch.zeros(D * 4 * H, device=DEV)
    assert_close(ops.rows_wgrad(a, x, dbias=db, arith=_lib.ARITH_BF16X3), a.double().t() @ x.double(), TOL, "input weights")
    assert_close(db, a.double().sum(0), TOL, "bias")


def test_linear_fn_matches_float64():
    from speech_amd.functional import LinearFn
    x = synth.normal((6, 50, 256), 10)
    w, b = synth.normal((64, 256), 11) * 0.1, synth.normal((64,), 12)
    gy = synth.normal((6, 50, 64), 13)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    (torch.nn.functional.linear(x64, w64, b64) * gy.double()).sum().backward()
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = LinearFn.apply(xg, wg, bg)
    (y * gy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert_close(y, torch.nn.functional.linear(x64, w64, b64), TOL, "y")
    assert_close(xg.grad, x64.grad, TOL, "dx")
    assert_close(wg.grad, w64.grad, TOL, "dW")
    assert_close(bg.grad, b64.grad, TOL, "db")
