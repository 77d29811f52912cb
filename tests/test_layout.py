"""The polyphase formulation the HIP kernels implement (segan_pytorch_amd/layout.py)
against torch.nn.functional on CPU: packings, reflect/zero padded views, phase shift."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from segan_pytorch_amd import layout as lay

torch.backends.mkldnn.enabled = False


@pytest.mark.parametrize('S,K', [(4, 31), (2, 31), (1, 31), (4, 5), (2, 32)])
@pytest.mark.parametrize('roll', [0, 3, -2])
def test_conv_forms(S, K, roll):
    rng = np.random.default_rng(0)
    B, N, M, L = 2, 3, 5, 32 if K < 32 else 64
    x = rng.standard_normal((B, N, L))
    w = rng.standard_normal((M, N, K))
    pl, pr = lay.conv_pad(K, S)
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    a = F.conv1d(F.pad(torch.roll(xt, roll, 2), (pl, pr), mode='reflect'), wt, stride=S)
    assert np.abs(lay.corr_f(x, w, S, pl, lay.PAD_REFLECT, roll) - a.detach().numpy()).max() < 1e-12
    da = rng.standard_normal(tuple(a.shape))
    a.backward(torch.tensor(da))
    Ls = a.shape[2]
    Tcols = (L + pl + pr - 1) // S + 1
    dxp = lay.corr_t(da, w, S, 0, Tcols=Tcols)
    dx = lay.fold_reflect(dxp[:, :, :L + pl + pr], L, pl, roll)
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-12
    assert np.abs(lay.wgrad(da, x, S, K, pl, lay.PAD_REFLECT, roll) - wt.grad.numpy()).max() < 1e-12
    assert Ls == L // S


@pytest.mark.parametrize('S,K', [(4, 31), (2, 31), (4, 32)])
def test_deconv_forms(S, K):
    rng = np.random.default_rng(1)
    B, Mi, No, Ls = 2, 3, 5, 16
    x = rng.standard_normal((B, Mi, Ls))
    w = rng.standard_normal((Mi, No, K))
    pad = lay.deconv_pad(K, S)
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    y = F.conv_transpose1d(xt, wt, stride=S, padding=pad)
    if K % 2:
        y = y[:, :, :-1]
    assert y.shape[2] == S * Ls
    assert np.abs(lay.corr_t(x, w, S, pad) - y.detach().numpy()).max() < 1e-12
    dy = rng.standard_normal(tuple(y.shape))
    y.backward(torch.tensor(dy))
    assert np.abs(lay.corr_f(dy, w, S, pad, lay.PAD_ZERO, 0) - xt.grad.numpy()).max() < 1e-12
    assert np.abs(lay.wgrad(x, dy, S, K, pad, lay.PAD_ZERO) - wt.grad.numpy()).max() < 1e-12


@pytest.mark.parametrize('S', [4, 2, 1])
def test_paired_f_packing_of_31_taps(S):
    """K = 31, an even number (> 2) of contraction channels: the F packing stores the even
    channel's row-30 weights in the zero-tap row of its odd partner and the contraction runs the
    paired schedule (15 + 16 two-row steps per channel pair instead of 16 + 16; layout.corr_f
    restates what corr2_kernel does).  Same results as torch for the conv forward and the deconv
    data gradient; the packed buffer differs from the plain one in exactly those rows."""
    rng = np.random.default_rng(2)
    B, N, M, L, K = 2, 6, 5, 64, 31
    assert lay.f_pair(N, K) and not lay.f_pair(3, K) and not lay.f_pair(2, K) and not lay.f_pair(N, 32)
    x = rng.standard_normal((B, N, L))
    w = rng.standard_normal((M, N, K))
    pl, pr = lay.conv_pad(K, S)
    a = F.conv1d(F.pad(torch.tensor(x), (pl, pr), mode='reflect'), torch.tensor(w), stride=S)
    assert np.abs(lay.corr_f(x, w, S, pl, lay.PAD_REFLECT, 0) - a.numpy()).max() < 1e-12
    U = lay.taps_per_phase(S)
    plain, paired = lay.pack_f(w, S, pair=False), lay.pack_f(w, S)
    diff = np.argwhere(np.abs(plain - paired).max(axis=2) > 0)
    assert sorted(map(tuple, diff)) == [((2 * p + 1) * S + S - 1, U - 1) for p in range(N // 2)]
    assert np.array_equal(paired[S + S - 1, U - 1], w[:, 0, 31 - S])      # channel 1 <- channel 0's row 30
    # deconv data gradient: the F form over the deconv's OUTPUT channels
    Mi, No, Ls = 3, 4, 16
    xd = torch.tensor(rng.standard_normal((B, Mi, Ls)), requires_grad=True)
    wd = rng.standard_normal((Mi, No, K))
    pad = lay.deconv_pad(K, S)
    y = F.conv_transpose1d(xd, torch.tensor(wd), stride=S, padding=pad)[:, :, :S * Ls]
    dy = rng.standard_normal(tuple(y.shape))
    y.backward(torch.tensor(dy))
    assert np.abs(lay.corr_f(dy, wd, S, pad, lay.PAD_ZERO, 0) - xd.grad.numpy()).max() < 1e-12


def test_hi_index_matches_reflect_and_roll():
    L, padL, padR = 20, 14, 15
    x = torch.arange(L, dtype=torch.float64).view(1, 1, L)
    for roll in (0, 4, -3):
        ref = F.pad(torch.roll(x, roll, 2), (padL, padR), mode='reflect')[0, 0]
        got = [lay.hi_index(p, L, padL, lay.PAD_REFLECT, roll) for p in range(L + padL + padR)]
        assert [int(v) for v in ref] == got


def test_phase_table_of_the_reference_geometry():
    # k31/s4 deconv: pad 13 -> phases r=0..2 use c=3, r=3 uses c=4 (one extra input shift)
    assert lay.deconv_pad(31, 4) == 13 and lay.deconv_pad(31, 2) == 14
    assert lay.t_phase_table(4, 13) == [(1, 3), (2, 3), (3, 3), (0, 4)]
    assert lay.conv_pad(31, 4) == (14, 15) and lay.conv_pad(31, 1) == (15, 15)
