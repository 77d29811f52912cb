"""Host-side statement of the data layouts the HIP kernels use.

Everything the two contraction kernels (``corr`` and ``wgrad`` in
``csrc/segan_conv.hip``) do is expressed through three small pieces of index
arithmetic, restated here in plain Python so that the CPU test-suite can check
them against ``torch.nn.functional`` on the same inputs:

* the *polyphase weight packings* ``pack_f`` / ``pack_t`` (what
  ``segan_pack_weights`` writes),
* the *high-resolution view* index map ``hi_index`` (reflect / zero padding and
  the discriminator's circular phase shift, reference
  ``segan/models/modules.py:91-98`` and ``segan/models/discriminator.py:160-172``),
* the per-output-phase tap/shift table ``t_phase_table`` of the transposed form.

A strided 1-D convolution with stride S and K<=32 taps is split into S phases of
U=32/S taps each (tap k = S*u + r).  With that split

  conv fwd      a[m,t]      = sum_{n,r,u} Wf[(n,r),u,m] * X_r[n, t+u]          ("F form")
  deconv fwd    y[n,S*q+r]  = sum_{m,u'} Wt[m,u',(r,n)] * x[m, q+c(r)-(U-1)+u'] ("T form")
  weight grad   dW[m,n,S*u+r] = sum_{b,t} lo[b,m,t] * HI_r[b,n,t+u]            ("W form")

where X_r[n,q] = xpad[n, S*q+r] is phase r of the padded high-resolution
signal.  conv dgrad is the T form, deconv dgrad the F form, and both weight
gradients are the W form; the weight tensor is [m, n, k] in all cases
(Conv1d: [Cout, Cin, K]; ConvTranspose1d: [Cin, Cout, K]).

Nothing in here runs on the product path: the product packs weights on the GPU.
"""
import numpy as np

KP = 32          # taps are padded to 32 so every phase has U = 32 // S taps
PAD_REFLECT = 0
PAD_ZERO = 1


def taps_per_phase(S):
    assert S in (1, 2, 4), S
    return KP // S


def conv_pad(K, S):
    """(left, right) reflect padding of GConv1DBlock (modules.py:91-97)."""
    if S > 1:
        return K // 2 - 1, K // 2
    return K // 2, K // 2


def deconv_pad(K, S):
    """ConvTranspose1d padding of GDeconv1DBlock (modules.py:115)."""
    return max(0, (S - K) // -2)


def f_pair(N, K):
    """Does the F packing of an [m, N, K] weight pair its channels?  (segan_conv_shared.h)"""
    return K == 31 and N > 2 and N % 2 == 0


def pack_f(w, S, pair=None):
    """[m, n, K] -> Wf[(n, r), u, m]  with Wf = w[m, n, S*u + r] (0 past K).

    K = 31 (`f_pair`): the 32nd row of a channel — (r, u) = (S-1, U-1), the padding tap k = 31 —
    multiplies zeros, half of one two-row MFMA.  For every ODD channel n that row holds row 30
    = (S-1, U-2), i.e. tap k = 31 - S, of the even partner n - 1 instead: the contraction kernel
    runs 15 two-row steps on the even channel, 16 on the odd one, and the odd channel's last
    step contracts (its own row 30, the partner's row 30) — against the partner's row-30
    activations in the upper half (`corr_f` below restates that schedule)."""
    w = np.asarray(w)
    M, N, K = w.shape
    U = taps_per_phase(S)
    out = np.zeros((N * S, U, M), dtype=w.dtype)
    for r in range(S):
        for u in range(U):
            k = S * u + r
            if k < K:
                out[r::S, u, :] = w[:, :, k].T        # row index n*S + r
    if f_pair(N, K) if pair is None else pair:
        out[2 * S - 1::2 * S, U - 1, :] = w[:, 0::2, 31 - S].T     # rows (odd n, S-1), tap slot U-1
    return out


def t_phase_table(S, pad):
    """For output phase r of the T form: (rho, c) with j + pad = S*(q + c) + rho."""
    return [((r + pad) % S, (r + pad) // S) for r in range(S)]


def pack_t(w, S, pad, NP=None):
    """[m, n, K] -> Wt[m, u', (r, n)], Wt = w[m, n, S*(U-1-u') + rho(r)] (0 past K).

    Rows are ordered phase-major (row = r*NP + n, NP = n rounded up to 32) so
    that every 32-row MFMA block has one phase and therefore one input shift.
    """
    w = np.asarray(w)
    M, N, K = w.shape
    U = taps_per_phase(S)
    if NP is None:
        NP = (N + 31) // 32 * 32
    out = np.zeros((M, U, S * NP), dtype=w.dtype)
    for r, (rho, _c) in enumerate(t_phase_table(S, pad)):
        for up in range(U):
            k = S * (U - 1 - up) + rho
            if k < K:
                out[:, up, r * NP:r * NP + N] = w[:, :, k]
    return out


def hi_index(p, L, padL, mode, roll):
    """Padded coordinate p -> index into the stored high-res row, or -1 for zero.

    reflect (modules.py:98): i = p - padL mirrored without repeating the edge;
    roll (discriminator.py:160-172): the conv sees torch.roll(h, roll), i.e.
    rolled[i] = h[(i - roll) mod L].
    """
    i = p - padL
    if mode == PAD_REFLECT:
        if i < 0:
            i = -i
        if i >= L:
            i = 2 * (L - 1) - i
        if i < 0 or i >= L:
            return -1            # only reachable through the zero tap k = 31
    else:
        if i < 0 or i >= L:
            return -1
    return (i - roll) % L


# ---------------------------------------------------------------------------------
# Emulations of the kernels' arithmetic (slow, small sizes only; used by tests)
# ---------------------------------------------------------------------------------

def hi_phases(x, S, padL, mode, roll, Q):
    """x [B, N, L] -> X[B, (n, r), Q] with X[b, n*S+r, q] = xpad[b, n, S*q + r]."""
    B, N, L = x.shape
    out = np.zeros((B, N * S, Q), dtype=x.dtype)
    for q in range(Q):
        for r in range(S):
            i = hi_index(S * q + r, L, padL, mode, roll)
            if i >= 0:
                out[:, r::S, q] = x[:, :, i]
    return out


def corr_f(x, w, S, padL, mode, roll=0):
    """F form: returns a[b, m, t], t in [0, L/S)."""
    B, N, L = x.shape
    M = w.shape[0]
    U = taps_per_phase(S)
    Ls = L // S
    X = hi_phases(x, S, padL, mode, roll, Ls + U - 1)
    Wf = pack_f(w, S)                     # [(n,r), u, m]
    out = np.zeros((B, M, Ls), dtype=np.float64)
    if not f_pair(N, w.shape[2]):
        for u in range(U):
            # [B, CV, Ls] x [CV, M]
            out += np.einsum('bct,cm->bmt', X[:, :, u:u + Ls].astype(np.float64),
                             Wf[:, u, :].astype(np.float64))
        return out
    # the paired schedule of corr2_kernel: rows (r, u) of a channel in order, two per step
    Xd, Wd = X.astype(np.float64), Wf.astype(np.float64)
    rows = [(r, u) for r in range(S) for u in range(U)]          # row 30 = (S-1, U-2), 31 = (S-1, U-1)

    def row(n, i, xn=None):
        r, u = rows[i]
        xs = Xd[:, (n if xn is None else xn) * S + r, u:u + Ls]           # [B, Ls]
        return np.einsum('bt,m->bmt', xs, Wd[n * S + r, u, :])

    for a in range(0, N, 2):
        b = a + 1
        for i in range(30):                       # chunk A: 15 steps, row 30 deferred
            out += row(a, i)
        for i in range(31):                       # chunk B: rows 0 .. 30 ...
            out += row(b, i)
        # ... and the upper half of its last step: B's row 31 (= A's row-30 weights) against
        # A's row-30 activations
        r30, u30 = rows[30]
        out += np.einsum('bt,m->bmt', Xd[:, a * S + r30, u30:u30 + Ls], Wd[b * S + S - 1, U - 1, :])
    return out


def corr_t(x, w, S, pad, Tcols=None):
    """T form over the coordinate j + pad = S*t + k.

    x [B, M, Ls], w [M, N, K] -> y[b, n, j] for j in [0, S*Tcols).
    With pad = deconv_pad this is ConvTranspose1d trimmed to S*Ls samples; with
    pad = 0 and Tcols = Ls + U - 1 it is the gradient w.r.t. the *padded* conv
    input (conv dgrad before the reflect fold).
    """
    B, M, Ls = x.shape
    N = w.shape[1]
    U = taps_per_phase(S)
    if Tcols is None:
        Tcols = Ls
    NP = (N + 31) // 32 * 32
    Wt = pack_t(w, S, pad, NP)            # [m, u', (r, n)]
    tab = t_phase_table(S, pad)
    cmin = min(c for _, c in tab)
    y = np.zeros((B, N, S * Tcols), dtype=np.float64)
    # window: t = q + c(r) - (U-1) + u'
    xp = np.zeros((B, M, Tcols + U + 1 + 2 * U), dtype=np.float64)
    off0 = U  # xp[.., off0 + t] = x[.., t]
    xp[:, :, off0:off0 + Ls] = x
    for r, (_rho, c) in enumerate(tab):
        for up in range(U):
            sh = c - (U - 1) + up
            seg = xp[:, :, off0 + sh: off0 + sh + Tcols]      # x[m, q + sh]
            y[:, :, r::S] += np.einsum('bmq,mn->bnq', seg,
                                       Wt[:, up, r * NP:r * NP + N].astype(np.float64))
    assert cmin >= 0
    return y


def fold_reflect(dxp, L, padL, roll):
    """Gradient w.r.t. the padded+rolled input -> gradient w.r.t. the stored row."""
    B, N, P = dxp.shape
    dx = np.zeros((B, N, L), dtype=dxp.dtype)
    for p in range(P):
        i = hi_index(p, L, padL, PAD_REFLECT, roll)
        if i >= 0:
            dx[:, :, i] += dxp[:, :, p]
    return dx


def wgrad(lo, hi, S, K, padL, mode, roll=0):
    """W form: dW[m, n, k] = sum_{b,t} lo[b,m,t] * hipad[b,n,S*t+k]."""
    B, M, Ls = lo.shape
    N = hi.shape[1]
    U = taps_per_phase(S)
    X = hi_phases(hi, S, padL, mode, roll, Ls + U - 1).astype(np.float64)
    dW = np.zeros((M, N, K), dtype=np.float64)
    for u in range(U):
        part = np.einsum('bmt,bct->mc', lo.astype(np.float64), X[:, :, u:u + Ls])  # [m,(n,r)]
        for r in range(S):
            k = S * u + r
            if k < K:
                dW[:, :, k] = part[:, r::S]
    return dW
