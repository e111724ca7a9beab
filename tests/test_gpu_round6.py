"""GPU, round 6.

  * ragged M in the MFMA GEMMs (tuning key 43): 64-row blocks / waves that lie wholly beyond M issue no operand reads and no MFMAs.  The valid
    rows are computed by exactly the same instructions, so not a bit may differ from the full issue -- every epilogue, every tile kernel the ViT
    shapes use (M = 3075 = 12 x 256 + 3 is the shape this is for: modeling_intern_vit.py:124,136,184-185 at 3 tiles).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import DT, CODE, TOL, dev, rnd, rel, ptr, sync, randn
from omchat_amd import _lib

DTS = ["bf16", "f16"]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile", [2, 10, 11, 9, 1, 12])
@pytest.mark.parametrize("M", [259, 200, 321, 3075, 64, 513])
def test_gemm_dead_row_blocks_skipped_same_bits(gpu_lib, dt, tile, M):
    N, K = 1024, 192
    A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt)
    bias = rnd(randn((N,), 3, 0.1), dt); ls = rnd(randn((N,), 4, 0.1) + 0.1, dt); resid = rnd(randn((M, N), 5), dt)
    dA, dW, db, dl, dr = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt)
    y = A @ W.t() + bias
    for epi in (_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_LS_RESID):
        outs = {}
        try:
            for key in (1, 0):
                gpu_lib.omchat_op_set_tuning(43, key)
                # one guard row behind the matrix: nothing may be written beyond row M - 1
                out = torch.full((M + 1, N), 77.0, dtype=DT[dt], device="cuda")
                _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), N, M, N, K, ptr(db), ptr(dl), ptr(dr), N, epi, tile, None))
                sync()
                outs[key] = out.clone()
        finally:
            gpu_lib.omchat_op_set_tuning(43, 1)
        assert torch.equal(outs[1], outs[0]), (epi, tile, M)
        assert bool((outs[1][M] == 77.0).all())
        if epi == _lib.EPI_NONE:
            assert rel(outs[1][:M], rnd(y, dt)) < TOL[dt]
