"""TEST-ONLY CPU emulation of the ``segan_pytorch_amd.ops`` entry points.

The build container has no GPU, so the kernels themselves can only be checked on the
GPU box (``-m gpu``).  What CAN be checked here is all the host logic above them: the
autograd nodes of ``functional.py`` (which tensor feeds which kernel, the skip / z /
alpha bookkeeping, gradient accumulation into ``.grad``), the flat-arena optimizers,
the training loop and checkpointing.  ``install()`` monkey-patches each ``ops.<entry>``
with a torch-CPU restatement of the SAME contract (include/segan_hip.h), so the host
code runs end to end on CPU against the golden fixtures.  Nothing outside tests/ imports
this module; the product never runs without the HIP library.
"""
import math

import torch
import torch.nn.functional as F

from segan_pytorch_amd import layout, ops, optim

_saved = {}


def _mat(src):
    """Materialise a Src: concat + per-channel affine + PReLU, in float64."""
    x = src.t0 if src.t1 is None else torch.cat((src.t0, src.t1), 1)
    x = x.double()
    if src.scale is not None:
        x = x * src.scale.detach().double().view(1, -1, 1)
    if src.shift is not None:
        x = x + src.shift.detach().double().view(1, -1, 1)
    if src.slope is not None:
        x = torch.where(x > 0, x, x * src.slope.detach().double().view(1, -1, 1))
    return x


def _conv(x, w, b, S, roll, mode, padL):
    K = w.shape[2]
    x = torch.roll(x, roll, 2)
    padR = K - 1 - padL if mode == ops.PAD_REFLECT else None
    if mode == ops.PAD_REFLECT:
        xp = F.pad(x, (padL, padR), mode='reflect')
    else:
        # zero padding in padded coordinates: out[t] = sum_k w[k] x[S t + k - padL]
        L = x.shape[2]
        Ls = L // S
        need = S * (Ls - 1) + K
        xp = F.pad(x, (padL, max(0, need - padL - L)))
    return F.conv1d(xp, w, b, stride=S)


def conv1d_fwd(src, w, bias, S, roll=0, pad_mode=ops.PAD_REFLECT, padL=None, pack=None):
    K = w.shape[2]
    if padL is None:
        padL = layout.conv_pad(K, S)[0]
    b = bias.detach().double() if bias is not None else None
    return _conv(_mat(src), w.detach().double(), b, S, roll, pad_mode, padL).float()


def conv1d_dgrad(da, w, L, S, roll=0, padL=None, pack=None):
    M, N, K = w.shape
    if padL is None:
        padL = layout.conv_pad(K, S)[0]
    with torch.enable_grad():
        x = torch.zeros(da.shape[0], N, L, dtype=torch.float64, requires_grad=True)
        y = _conv(x, w.detach().double(), None, S, roll, ops.PAD_REFLECT, padL)
        (g,) = torch.autograd.grad(y, x, da.double())
    return g.float()


def wgrad(lo, hi, dw, K, S, padL, pad_mode, roll=0):
    with torch.enable_grad():
        w = torch.zeros(dw.shape, dtype=torch.float64, requires_grad=True)
        y = _conv(_mat(hi), w, None, S, roll, pad_mode, padL)
        (g,) = torch.autograd.grad(y, w, _mat(lo))
    dw.add_(g.float())


def _deconv(x, w, b, S):
    K = w.shape[2]
    y = F.conv_transpose1d(x, w, b, stride=S, padding=layout.deconv_pad(K, S))
    return y[:, :, :-1] if K % 2 else y


def deconv1d_fwd(src, w, bias, S, act=ops.ACT_NONE, pack=None):
    b = bias.detach().double() if bias is not None else None
    y = _deconv(_mat(src), w.detach().double(), b, S)
    if act == ops.ACT_TANH:
        y = torch.tanh(y)
    return y.float()


def deconv1d_dgrad(dy, w, S, M0=0, need0=True, need1=True, pack=None):
    M = w.shape[0]
    Ls = dy.shape[2] // S
    with torch.enable_grad():
        x = torch.zeros(dy.shape[0], M, Ls, dtype=torch.float64, requires_grad=True)
        (g,) = torch.autograd.grad(_deconv(x, w.detach().double(), None, S), x, dy.double())
    g = g.float()
    dx0 = g[:, :M0].contiguous() if (M0 > 0 and need0) else None
    dx1 = g[:, M0:].contiguous() if (M - M0 > 0 and need1) else None
    return dx0, dx1


def bn_stats(x, gamma, beta, eps, momentum, running_mean, running_var):
    xd = x.double()
    n = x.shape[0] * x.shape[2]
    mean = xd.mean((0, 2))
    var = xd.var((0, 2), unbiased=False)
    rstd = (var + eps).rsqrt()
    scale = gamma.detach().double() * rstd
    shift = beta.detach().double() - mean * scale
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(momentum * mean.float())
        running_var.mul_(1 - momentum).add_(momentum * (var * n / max(n - 1, 1)).float())
    return mean.float(), rstd.float(), scale.float(), shift.float()


def bn_partial(x):
    xd = x.double()
    n = float(x.shape[0] * x.shape[2])
    mean = xd.mean((0, 2))
    m2 = ((xd - mean.view(1, -1, 1)) ** 2).sum((0, 2))
    return torch.stack((torch.full_like(mean, n), mean, m2), 1).float().unsqueeze(0)   # [1, C, 3]


def bn_final(ws_all, gamma, beta, eps, momentum, running_mean, running_var):
    w = ws_all.double()
    n, mean, m2 = torch.zeros(w.shape[1], dtype=torch.float64), None, None
    for s_ in range(w.shape[0]):                       # Chan et al. pairwise combination
        nb, mb, qb = w[s_, :, 0], w[s_, :, 1], w[s_, :, 2]
        if mean is None:
            n, mean, m2 = nb.clone(), mb.clone(), qb.clone()
            continue
        tot = n + nb
        d = mb - mean
        mean = mean + d * nb / tot
        m2 = m2 + qb + d * d * n * nb / tot
        n = tot
    var = m2 / n
    rstd = (var + eps).rsqrt()
    scale = gamma.detach().double() * rstd
    shift = beta.detach().double() - mean * scale
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(momentum * mean.float())
        running_var.mul_(1 - momentum).add_(momentum * (m2 / (n - 1).clamp_min(1)).float())
    return mean.float(), rstd.float(), scale.float(), shift.float()


def _bn_bwd_terms(a, dh, slope, bn):
    mean, rstd, gamma, beta = bn
    ad = a.double()
    mu, rs = mean.double().view(1, -1, 1), rstd.double().view(1, -1, 1)
    ga, be = gamma.detach().double().view(1, -1, 1), beta.detach().double().view(1, -1, 1)
    sl = slope.detach().double().view(1, -1, 1)
    xh = (ad - mu) * rs
    v = ga * xh + be
    dhd = dh.double()
    g = dhd * torch.where(v > 0, torch.ones_like(v), sl.expand_as(v))
    return xh, v, g, dhd, ga, rs


def act_bwd_bn_reduce(a, dh, slope, bn, dslope=None, dgamma=None, dbeta=None):
    xh, v, g, dhd, ga, rs = _bn_bwd_terms(a, dh, slope, bn)
    _acc(dslope, (dhd * torch.where(v > 0, torch.zeros_like(v), v)).sum((0, 2)))
    db, dg = g.sum((0, 2)), (g * xh).sum((0, 2))
    _acc(dbeta, db)
    _acc(dgamma, dg)
    return torch.stack((db, dg), 1).float().contiguous(), None


def act_bwd_bn_apply(a, dh, slope, bn, totals, count_total, dbias=None, ws=None):
    xh, v, g, dhd, ga, rs = _bn_bwd_terms(a, dh, slope, bn)
    db, dg = totals[:, 0].double().view(1, -1, 1), totals[:, 1].double().view(1, -1, 1)
    d = ga * rs * (g - db / count_total - xh * dg / count_total)
    _acc(dbias, d.sum((0, 2)))
    return d.float()


def affine_prelu(x, scale=None, shift=None, slope=None):
    return _mat(ops.Src(x, scale=scale, shift=shift, slope=slope)).float()


def affine_tanh(x, scale=None, shift=None):
    return torch.tanh(_mat(ops.Src(x, scale=scale, shift=shift))).float()


def scale_mask(x, scale, mask):
    v = x.double() * mask.double()
    if scale is not None:
        v = v * scale.detach().double().view(1, -1, 1)
    return v.float()


def sum_skip(x0, slope0, x1, alpha):
    v = x0.double()
    if slope0 is not None:
        v = torch.where(v > 0, v, v * slope0.detach().double().view(1, -1, 1))
    return (v + alpha.detach().double().view(1, -1, 1) * x1.double()).float()


def pool_time_fwd(x, mode):
    if mode == 'max':
        y, idx = x.max(dim=2)
        # first position attaining the maximum (segan_pool_time_fwd's tie rule)
        first = (x == y.unsqueeze(2)).float().argmax(dim=2)
        return y, first.to(torch.int32)
    return x.mean(dim=2), None


def pool_time_bwd(dy, idx, L, mode):
    B, C = dy.shape
    if mode == 'max':
        dx = torch.zeros(B, C, L)
        dx.scatter_(2, idx.long().unsqueeze(2), dy.unsqueeze(2))
        return dx
    return (dy / L).unsqueeze(2).expand(B, C, L).contiguous()


def bce_logits_const(x, target):
    t = torch.full_like(x.double(), target)
    return F.binary_cross_entropy_with_logits(x.double(), t).float()


def bce_logits_const_bwd(x, target, gout=None, gscale=1.0):
    g = (torch.sigmoid(x.double()) - target) / x.numel() * gscale
    if gout is not None:
        g = g * gout.double()
    return g.float()


def _acc(dst, val):
    if dst is not None:
        dst.add_(val.float().view(dst.shape))


def act_bwd(a, dh, dskip=None, slope=None, alpha=None, bn=None, dslope=None, dalpha=None,
            dgamma=None, dbeta=None, dbias=None):
    ad = a.double()
    sl = slope.detach().double().view(1, -1, 1) if slope is not None else None
    dhd = dh.double() if dh is not None else torch.zeros_like(ad)
    if bn is None:
        g = dhd * (torch.where(ad > 0, torch.ones_like(ad), sl.expand_as(ad)) if sl is not None
                   else 1.0)
        _acc(dslope, (dhd * torch.where(ad > 0, torch.zeros_like(ad), ad)).sum((0, 2)))
        if dskip is not None:
            g = g + alpha.detach().double().view(1, -1, 1) * dskip.double()
            _acc(dalpha, (dskip.double() * ad).sum((0, 2)))
        _acc(dbias, g.sum((0, 2)))
        return g.float()
    mean, rstd, gamma, beta = bn
    mu, rs = mean.double().view(1, -1, 1), rstd.double().view(1, -1, 1)
    ga = gamma.detach().double().view(1, -1, 1)
    be = beta.detach().double().view(1, -1, 1)
    xh = (ad - mu) * rs
    v = ga * xh + be
    # slope None = no activation behind the BatchNorm (the kernel's slope defaults to 1)
    g = dhd * torch.where(v > 0, torch.ones_like(v), sl.expand_as(v)) if sl is not None else dhd
    _acc(dslope, (dhd * torch.where(v > 0, torch.zeros_like(v), v)).sum((0, 2)))
    db, dg = g.sum((0, 2)), (g * xh).sum((0, 2))
    _acc(dbeta, db)
    _acc(dgamma, dg)
    n = a.shape[0] * a.shape[2]
    d = ga * rs * (g - db.view(1, -1, 1) / n - xh * dg.view(1, -1, 1) / n)
    _acc(dbias, d.sum((0, 2)))
    return d.float()


def tanh_bwd(y, dy, clean=None, l1_scale=0.0, dbias=None):
    yd = y.double()
    g = dy.double() if dy is not None else torch.zeros_like(yd)
    if clean is not None:
        g = g + l1_scale * torch.sign(yd - clean.double())
    d = g * (1 - yd * yd)
    _acc(dbias, d.sum((0, 2)))
    return d.float()


def linear_fwd(x, w):
    return (x.double() @ w.detach().double().t()).float()


def linear_dgrad(dy, w):
    return (dy.double() @ w.detach().double()).float()


def linear_wgrad(dy, x, dw):
    dw.add_((dy.double().t() @ x.double()).float())


def bias_prelu_rows(x, bias, slope):
    v = x.double() + (bias.detach().double() if bias is not None else 0.0)
    if slope is not None:
        v = torch.where(v > 0, v, v * slope.detach().double())
    return v.float()


def bias_prelu_rows_bwd(x, bias, slope, dy, dslope, dbias):
    v = x.double() + (bias.detach().double() if bias is not None else 0.0)
    sl = slope.detach().double() if slope is not None else torch.ones(x.shape[1], dtype=torch.float64)
    d = dy.double() * torch.where(v > 0, torch.ones_like(v), sl.expand_as(v))
    if slope is not None:
        _acc(dslope, (dy.double() * torch.where(v > 0, torch.zeros_like(v), v)).sum(0))
    _acc(dbias, d.sum(0))
    return d.float()


def mse_const(x, target):
    return ((x.double() - target) ** 2).mean().float()


def mse_const_bwd(x, target, gout=None, gscale=1.0):
    g = 2.0 * (x.double() - target) / x.numel() * gscale
    if gout is not None:
        g = g * gout.double()
    return g.float()


def l1_mean(x, y):
    return (x.double() - y.double()).abs().mean().float()


def l1_bwd(x, y, gout=None, gscale=1.0):
    g = torch.sign(x.double() - y.double()) / x.numel() * gscale
    if gout is not None:
        g = g * gout.double()
    return g.float()


def mse_mean(x, y):
    return ((x.double() - y.double()) ** 2).mean().float()


def mse_bwd(x, y, gout=None, gscale=1.0):
    g = 2.0 * (x.double() - y.double()) / x.numel() * gscale
    if gout is not None:
        g = g * gout.double()
    return g.float()


def _frame_index(T, n_fft, hop, win):
    NF = 1 + T // hop
    off = (n_fft - win) // 2 - n_fft // 2
    o = (torch.arange(NF).view(-1, 1) * hop + torch.arange(win).view(1, -1) + off)
    o = o.abs()
    o = torch.where(o >= T, 2 * (T - 1) - o, o)
    return o                     # [NF, win] source sample of every frame element


def stft_basis(n_fft, win, device):
    nb = n_fft // 2 + 1
    left = (n_fft - win) // 2
    ang = 2 * math.pi * ((torch.arange(nb).view(1, -1) * (left + torch.arange(win)).view(-1, 1))
                         % n_fft).double() / n_fft
    bs = (torch.cat((torch.cos(ang), -torch.sin(ang)), 1) / math.sqrt(n_fft)).float()
    pitch = (2 * nb + 3) // 4 * 4
    return torch.cat((bs, torch.zeros(win, pitch - 2 * nb)), 1)


def stft_frames(x, n_fft, hop, win):
    B, T = x.shape
    idx = _frame_index(T, n_fft, hop, win)
    return x[:, idx].reshape(-1, win).contiguous()


def stft_spectrum(frames, basis):
    return (frames.double() @ basis.double()).float()


def stft_spectrum_bwd(dS, basis):
    return (dS.double() @ basis.double().t()).float()


def powdb(S, nb, eps=10e-20):
    p = S[:, :nb].double() ** 2 + S[:, nb:2 * nb].double() ** 2
    return (10 * torch.log10(p + eps)).float()


def powdb_bwd(S, ddb, nb, eps=10e-20):
    re, im = S[:, :nb].double(), S[:, nb:2 * nb].double()
    g = ddb.double() * (20.0 / math.log(10.0)) / (re * re + im * im + eps)
    return torch.cat((g * re, g * im, torch.zeros(S.shape[0], S.shape[1] - 2 * nb,
                                                 dtype=torch.float64)), 1).float()


def stft_overlap_add(dframes, B, T, n_fft, hop, win):
    idx = _frame_index(T, n_fft, hop, win).reshape(-1)
    dx = torch.zeros(B, T, dtype=torch.float64)
    dx.index_add_(1, idx, dframes.double().view(B, -1))
    return dx.float()


def _sn_mat(w, dim):
    if dim != 0:
        w = w.permute(dim, *[d for d in range(w.dim()) if d != dim])
    return w.reshape(w.size(0), -1)


def snorm_fwd(w, u, v, dim, power_iteration, eps=1e-12):
    import torch.nn.functional as F
    wm = _sn_mat(w.detach(), dim)
    with torch.no_grad():
        if power_iteration:
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
        sigma = torch.dot(u, torch.mv(wm, v)).reshape(1)
        return (w.detach() / sigma).contiguous(), sigma


def snorm_bwd(dw_sn, w, u, v, sigma, dim, dw):
    wd = w.detach().double()
    d = (dw_sn.double() * wd).sum()
    outer = torch.outer(u.double(), v.double())      # [rows, cols] of the matrix view
    if dim == 0:
        outer = outer.reshape(wd.shape)
    else:
        perm = [dim] + [i for i in range(wd.dim()) if i != dim]
        shape = [wd.shape[i] for i in perm]
        inv = [perm.index(i) for i in range(wd.dim())]
        outer = outer.reshape(shape).permute(inv)
    s = sigma.double()
    dw.add_((dw_sn.double() / s - d / (s * s) * outer).float())


def rmsprop_step(p, g, sq, lr, alpha, eps):
    sq.mul_(alpha).addcmul_(g, g, value=1 - alpha)
    p.addcdiv_(g, sq.sqrt().add_(eps), value=-lr)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step):
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def fill_(t, value):
    return t.fill_(value)


def scale_(t, s):
    return t.mul_(s)


def _chk(t, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError('{} must have {} dims'.format(name, ndim))
    return t


_NAMES = ['conv1d_fwd', 'conv1d_dgrad', 'wgrad', 'deconv1d_fwd', 'deconv1d_dgrad', 'bn_stats',
          'affine_prelu', 'affine_tanh', 'scale_mask', 'sum_skip', 'pool_time_fwd', 'pool_time_bwd', 'bce_logits_const', 'bce_logits_const_bwd', 'act_bwd', 'tanh_bwd', 'linear_fwd', 'linear_dgrad', 'linear_wgrad',
          'bias_prelu_rows', 'bias_prelu_rows_bwd', 'mse_const', 'mse_const_bwd', 'l1_mean',
          'l1_bwd', 'mse_mean', 'mse_bwd', 'stft_basis', 'stft_frames', 'stft_spectrum', 'stft_spectrum_bwd', 'powdb',
          'powdb_bwd', 'stft_overlap_add', 'snorm_fwd', 'snorm_bwd', 'bn_partial', 'bn_final', 'act_bwd_bn_reduce',
          'act_bwd_bn_apply', 'rmsprop_step', 'adam_step', 'fill_', 'scale_', '_chk']


def install():
    if _saved:
        return
    g = globals()
    for n in _NAMES:
        _saved[n] = getattr(ops, n)
        setattr(ops, n, g[n])
    _saved['_require_cuda'] = optim._FlatOptimizer._require_cuda
    optim._FlatOptimizer._require_cuda = lambda self: None


def uninstall():
    if not _saved:
        return
    optim._FlatOptimizer._require_cuda = _saved.pop('_require_cuda')
    for n, f in list(_saved.items()):
        setattr(ops, n, f)
    _saved.clear()
