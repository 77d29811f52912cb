"""CPU ORACLE for the SEGAN+ GAN training step — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file; the product (``segan_pytorch_amd``) never does.

It is a line-by-line functional restatement of the reference's hot path
(santi-pdp/segan_pytorch @ /root/reference), written against plain state_dicts so that
it can be fed the weights of the reference modules or of the HIP modules alike:

* ``gconv_block``        segan/models/modules.py:91-105   (reflect pad, Conv1d, norm, PReLU)
* ``gdeconv_block``      segan/models/modules.py:135-141  (ConvTranspose1d, trim, PReLU/Tanh)
* ``generator_forward``  segan/models/generator.py:180-230 (+ GSkip.forward 64-78, incl. the
                         conv skip of :42-49, the sum merge and pooling-1 conv decoder blocks
                         of :171-176)
* ``roll``               segan/models/discriminator.py:160-172 (phase shift)
* ``discriminator_forward`` segan/models/discriminator.py:150-194 (heads 'none', 'conv', 'gmax',
                         'gavg' of :107-137)
* ``spectral_weight``    torch.nn.utils.spectral_norm as used by modules.py:12-14 and
                         discriminator.py:118-121 ('snorm')
* ``gan_step``           segan/models/model.py:292-321 with nn.MSELoss (train.py:94),
                         F.l1_loss (model.py:79) and optim.RMSprop (model.py:221-222)

The arithmetic itself lives in PyTorch (third-party; the reference pins torch==0.4.1 in
requirements.txt:6, this image has 2.10): the restatement calls the same
``torch.nn.functional`` ops the reference's nn modules dispatch to, on CPU, with oneDNN
disabled (this build's multi-threaded oneDNN ConvTranspose1d is
numerically wrong).

PINNING: ``tests/golden/*.pt`` hold outputs of the REAL reference run in the build
container (``oracle/make_golden.py`` imports it from /root/reference);
``tests/test_oracle.py`` checks this restatement against them.
"""
import torch
import torch.nn.functional as F

torch.backends.mkldnn.enabled = False


def roll(h, r):
    """Circular phase shift, discriminator.py:160-172.  r > 0: 'right'
    (cat(h[-r:], h[:-r])), r < 0: 'left' (cat(h[s:], h[:s]) with s = -r)."""
    if r == 0:
        return h
    if r > 0:
        return torch.cat((h[:, :, -r:], h[:, :, :-r]), dim=2)
    s = -r
    return torch.cat((h[:, :, s:], h[:, :, :s]), dim=2)


def _prelu(a, slope, gate=None):
    """F.prelu(a, slope); with `gate` (a bool tensor of a's shape) the SIDE of every element is
    imposed instead of read off the sign of a: where(gate, a, slope * a).  Test hook for the
    gate-aligned gradient comparisons (tests/test_gpu_kernels.py::
    test_discriminator_gradients_with_aligned_gates): a pre-activation within roundoff of zero may
    fall on either side in two correct fp32 implementations, and the derivative of PReLU is
    discontinuous there; imposing one implementation's sides on the other removes exactly that
    and nothing else (the forward value changes by at most |slope - 1| * |a|, a roundoff)."""
    if gate is None:
        return F.prelu(a, slope)
    sl = slope.view(*([1, -1] + [1] * (a.dim() - 2))) if slope.numel() > 1 else slope
    return torch.where(gate, a, sl * a)


def gconv_block(x, w, b, slope, stride, bn=None, training=True, gate=None):
    """modules.py:91-105.  Returns (h, a, bn_batch_stats)."""
    K = w.shape[2]
    P = (K // 2 - 1, K // 2) if stride > 1 else (K // 2, K // 2)
    a = F.conv1d(F.pad(x, P, mode='reflect'), w, b, stride=stride)
    if bn is not None:
        a = F.batch_norm(a, bn['running_mean'], bn['running_var'], bn['weight'], bn['bias'],
                         training, 0.1, 1e-5)
    h = _prelu(a, slope, gate)
    return h, a


def gdeconv_block(x, w, b, slope, stride, tanh=False, bn=None, training=True, gate=None):
    """modules.py:135-141 (pad from modules.py:115)."""
    K = w.shape[2]
    pad = max(0, (stride - K) // -2)
    h = F.conv_transpose1d(x, w, b, stride=stride, padding=pad)
    if K % 2 != 0:
        h = h[:, :, :-1]
    if bn is not None:
        h = F.batch_norm(h, bn['running_mean'], bn['running_var'], bn['weight'], bn['bias'],
                         training, 0.1, 1e-5)
    return torch.tanh(h) if tanh else _prelu(h, slope, gate)


def _bn_of(sd, p):
    """The BatchNorm1d of block prefix `p` as a dict, or None (build_norm_layer, modules.py:9-18)."""
    if p + 'norm.weight' not in sd:
        return None
    return {'weight': sd[p + 'norm.weight'], 'bias': sd[p + 'norm.bias'],
            'running_mean': sd[p + 'norm.running_mean'], 'running_var': sd[p + 'norm.running_var']}


def spectral_weight(sd, prefix, dim=0, training=True, eps=1e-12):
    """The weight torch.nn.utils.spectral_norm hands to the layer ('snorm': modules.py:12-14,
    discriminator.py:118-121), restated from its documented algorithm (torch is a third-party
    dependency of the reference): W_mat = weight_orig with `dim` first, flattened to 2-D;
    in training mode ONE power iteration under no_grad, in place on the u / v buffers,
        v <- normalize(W_mat^T u),  u <- normalize(W_mat v)      (eps 1e-12)
    then sigma = u . (W_mat v) with u, v treated as constants, weight = weight_orig / sigma.
    `sd[prefix + 'weight_u' / 'weight_v']` are updated in place like the module's buffers."""
    w = sd[prefix + 'weight_orig']
    u, v = sd[prefix + 'weight_u'], sd[prefix + 'weight_v']
    wm = w
    if dim != 0:
        wm = w.permute(dim, *[d for d in range(w.dim()) if d != dim])
    wm = wm.reshape(wm.size(0), -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
    sigma = torch.dot(u.clone(), torch.mv(wm, v.clone()))
    return w / sigma


def _weight(sd, prefix, dim=0, training=True):
    """`prefix`weight, or its spectrally normalised form when the layer carries
    weight_orig / weight_u / weight_v."""
    if prefix + 'weight_orig' in sd:
        return spectral_weight(sd, prefix, dim, training)
    return sd[prefix + 'weight']


def _count(sd, prefix):
    n = 0
    while '{}.{}.act.weight'.format(prefix, n) in sd or \
            '{}.{}.deconv.weight'.format(prefix, n) in sd or \
            '{}.{}.deconv.weight_orig'.format(prefix, n) in sd:
        n += 1
    return n


def generator_forward(sd, x, z, strides, dec_strides=None, ret_hid=False, training=True,
                      skip_merge='concat', skip_dropout=0.0, gates=None):
    """generator.py:180-230.  The architecture is read off the state_dict: a level has a skip
    when ``alpha_<l>.skip_k`` (alpha / constant) or ``alpha_<l>.skip_k.weight`` (conv skip,
    generator.py:42-49) exists; a decoder block is a transposed conv (``deconv``) or, for a
    pooling of 1, a GConv1DBlock (``conv``, generator.py:171-176); a block has a BatchNorm1d
    when ``norm.weight`` exists (norm_type='bnorm'; the running buffers in `sd` are updated in
    place in training mode).  `skip_dropout` > 0: nn.Dropout on the skip path
    (generator.py:53-54,70-71), drawing its masks from torch's global RNG like the reference.
    `gates`: {'enc_<l>' / 'dec_<l>': bool tensor} imposes PReLU sides (test hook, see _prelu)."""
    gates = gates or {}
    n_enc = _count(sd, 'enc_blocks')
    n_dec = _count(sd, 'dec_blocks')
    dec_strides = dec_strides or list(strides)
    hall = {}
    hi = x
    skips = {}
    for l in range(n_enc):
        p = 'enc_blocks.{}.'.format(l)
        hi, lin = gconv_block(hi, _weight(sd, p + 'conv.', 0, training), sd.get(p + 'conv.bias'),
                              sd[p + 'act.weight'], strides[l], bn=_bn_of(sd, p), training=training,
                              gate=gates.get('enc_{}'.format(l)))
        if l < n_enc - 1 and ('alpha_{}.skip_k'.format(l) in sd or
                              'alpha_{}.skip_k.weight'.format(l) in sd):
            skips[l] = lin                       # the PRE-activation (generator.py:185,191)
        hall['enc_{}'.format(l)] = hi
    if z is not None:
        hi = torch.cat((z, hi), dim=1)           # generator.py:205
        hall['enc_zc'] = hi
    enc_idx = n_enc - 1
    for l in range(n_dec):
        if enc_idx in skips and dec_strides[l] > 1:
            hj = skips[enc_idx]
            if 'alpha_{}.skip_k.weight'.format(enc_idx) in sd:      # GSkip 'conv'
                wk = sd['alpha_{}.skip_k.weight'.format(enc_idx)]
                kw = wk.shape[2]
                sk = F.conv1d(hj, wk, sd.get('alpha_{}.skip_k.bias'.format(enc_idx)), stride=1,
                              padding=kw // 2 if kw > 1 else 0)
            else:
                alpha = sd['alpha_{}.skip_k'.format(enc_idx)]
                sk = alpha.repeat(hi.size(0), 1, hj.size(2)) * hj
            if skip_dropout > 0:
                sk = F.dropout(sk, skip_dropout, training)
            # GSkip merge, generator.py:64-76
            hi = torch.cat((hi, sk), dim=1) if skip_merge == 'concat' else sk + hi
        p = 'dec_blocks.{}.'.format(l)
        if p + 'conv.weight' in sd or p + 'conv.weight_orig' in sd:
            hi, _ = gconv_block(hi, _weight(sd, p + 'conv.', 0, training), sd.get(p + 'conv.bias'),
                                sd[p + 'act.weight'], dec_strides[l], bn=_bn_of(sd, p),
                                training=training, gate=gates.get('dec_{}'.format(l)))
        else:
            last = (p + 'act.weight') not in sd
            hi = gdeconv_block(hi, _weight(sd, p + 'deconv.', 1, training), sd[p + 'deconv.bias'],
                               sd.get(p + 'act.weight'), dec_strides[l], tanh=last,
                               bn=_bn_of(sd, p), training=training,
                               gate=gates.get('dec_{}'.format(l)))
        enc_idx -= 1
        hall['dec_{}'.format(l)] = hi
    return (hi, hall) if ret_hid else hi


def discriminator_forward(sd, x, rolls, strides, training=True, ret_act=False, pool_type='none',
                          gates=None):
    """discriminator.py:150-194 (norm 'bnorm', 'snorm' or none; heads 'none', 'conv', 'gmax',
    'gavg').  `sd` must hold the BN running buffers when bnorm; they are updated in place like
    nn.BatchNorm1d.  `gates`: {'h_<l>', 'fc_1', 'fc_3': bool tensor} imposes PReLU sides (test
    hook, see _prelu); `acts` then also carries the pre-activations 'a_<l>', 'fc_a1', 'fc_a3'."""
    gates = gates or {}
    n = _count(sd, 'enc_blocks')
    h = x
    acts = {}
    for l in range(n):
        p = 'enc_blocks.{}.'.format(l)
        h = roll(h, rolls[l])
        bn = _bn_of(sd, p)
        h, a = gconv_block(h, _weight(sd, p + 'conv.', 0, training), sd.get(p + 'conv.bias'),
                           sd[p + 'act.weight'], strides[l], bn=bn, training=training,
                           gate=gates.get('h_{}'.format(l)))
        acts['h_{}'.format(l)] = h
        acts['a_{}'.format(l)] = a
    if pool_type == 'none':
        h = h.view(h.size(0), -1)
        a1 = F.linear(h, _weight(sd, 'fc.0.', 0, training), sd['fc.0.bias'])
        h = _prelu(a1, _weight(sd, 'fc.1.', 0, training), gates.get('fc_1'))
        a3 = F.linear(h, _weight(sd, 'fc.2.', 0, training), sd['fc.2.bias'])
        h = _prelu(a3, _weight(sd, 'fc.3.', 0, training), gates.get('fc_3'))
        acts['fc_a1'], acts['fc_a3'] = a1, a3
        y = F.linear(h, _weight(sd, 'fc.4.', 0, training), sd['fc.4.bias'])
    else:
        if pool_type == 'conv':                  # discriminator.py:122-127,175-179
            h = F.conv1d(h, _weight(sd, 'pool_conv.', 0, training), sd['pool_conv.bias'])
            h = h.view(h.size(0), -1)
            acts['avg_conv_h'] = h
        elif pool_type == 'gmax':                # AdaptiveMaxPool1d(1), :128-132,183-186
            h = h.max(dim=2)[0]
        elif pool_type == 'gavg':                # AdaptiveAvgPool1d(1), :133-137,187-190
            h = h.mean(dim=2)
        else:
            raise ValueError('pool_type {!r}'.format(pool_type))
        y = F.linear(h, _weight(sd, 'fc.', 0, training), sd['fc.bias'])
    acts['logit'] = y
    return (y, acts) if ret_act else y


_BUFFERS = ('running_mean', 'running_var', 'num_batches_tracked', 'weight_u', 'weight_v')


def _leafs(sd, frozen=()):
    """Leaf copies of a state dict: parameters require grad, buffers — and the `frozen` keys
    (--skip_type constant: skip_k.requires_grad = False, generator.py:40-41, so core.py:196-198
    never hands it to the optimizer) — do not."""
    out = {}
    for k, v in sd.items():
        if k.split('.')[-1] in _BUFFERS or k in frozen:
            out[k] = v.clone()
        else:
            out[k] = v.clone().requires_grad_(True)
    return out


def _is_param(k):
    return k.split('.')[-1] not in _BUFFERS


def rmsprop_update(p, g, sq, lr, alpha=0.99, eps=1e-8):
    """torch.optim.RMSprop single-tensor rule (momentum 0, not centered)."""
    sq.mul_(alpha).addcmul_(g, g, value=1 - alpha)
    p.addcdiv_(g, sq.sqrt().add_(eps), value=-lr)


def gan_step(g_sd, d_sd, clean, noisy, z, rolls3, strides, l1_weight=100.0, lr=5e-5,
             g_sq=None, d_sq=None, update=True, dec_strides=None, d_strides=None,
             skip_merge='concat', pool_type='none', reg_loss='l1_loss', frozen=()):
    """One SEGAN step, model.py:292-321.  rolls3 = three roll lists (D real, D fake,
    D fake-for-G).  `reg_loss`: 'l1_loss' | 'mse_loss' = getattr(F, opts.reg_loss) of model.py:79
    (train.py:179); `frozen`: generator keys that are not trained (--skip_type constant).
    Returns a dict with outputs, losses, gradients and (when `update`) the updated parameters /
    RMSprop state."""
    G = _leafs(g_sd, frozen)
    D = _leafs(d_sd)
    out = {}
    gen = lambda: generator_forward(G, noisy, z, strides, dec_strides, skip_merge=skip_merge)
    ds = d_strides or strides
    disc = lambda x, r: discriminator_forward(D, x, r, ds, pool_type=pool_type)
    # (1)+(2) discriminator update
    Genh = gen()
    d_real = disc(torch.cat((clean, noisy), 1), rolls3[0])
    d_real_loss = F.mse_loss(d_real.view(-1), torch.ones(clean.size(0), dtype=clean.dtype))
    d_fake = disc(torch.cat((Genh.detach(), noisy), 1), rolls3[1])
    d_fake_loss = F.mse_loss(d_fake.view(-1), torch.zeros(clean.size(0), dtype=clean.dtype))
    dkeys = [k for k in D if _is_param(k)]
    dgr = torch.autograd.grad(d_real_loss + d_fake_loss, [D[k] for k in dkeys])
    out['Genh'] = Genh.detach()
    out['d_real'] = d_real.detach()
    out['d_fake'] = d_fake.detach()
    out['d_real_loss'] = d_real_loss.detach()
    out['d_fake_loss'] = d_fake_loss.detach()
    out['d_grads'] = {k: g for k, g in zip(dkeys, dgr)}
    d_sq = d_sq or {k: torch.zeros_like(D[k]) for k in dkeys}
    if update:
        with torch.no_grad():
            for k, g in zip(dkeys, dgr):
                rmsprop_update(D[k], g, d_sq[k], lr)
    # (3) generator update through the updated D
    d_fake_ = disc(torch.cat((Genh, noisy), 1), rolls3[2])
    g_adv = F.mse_loss(d_fake_.view(-1), torch.ones(clean.size(0), dtype=clean.dtype))
    g_l1 = l1_weight * getattr(F, reg_loss)(Genh, clean)
    gkeys = [k for k in G if G[k].requires_grad]
    ggr = torch.autograd.grad(g_adv + g_l1, [G[k] for k in gkeys], allow_unused=True)
    # a skip the forward never takes (pooling-1 decoder level) has no gradient: like
    # torch.optim, which skips parameters whose .grad is None
    gkeys = [k for k, g in zip(gkeys, ggr) if g is not None]
    ggr = [g for g in ggr if g is not None]
    out['d_fake_'] = d_fake_.detach()
    out['g_adv_loss'] = g_adv.detach()
    out['g_l1_loss'] = g_l1.detach()
    out['g_grads'] = {k: g for k, g in zip(gkeys, ggr)}
    g_sq = g_sq or {k: torch.zeros_like(G[k]) for k in gkeys}
    if update:
        with torch.no_grad():
            for k, g in zip(gkeys, ggr):
                rmsprop_update(G[k], g, g_sq[k], lr)
    out['G'] = {k: v.detach() for k, v in G.items()}
    out['D'] = {k: v.detach() for k, v in D.items()}
    out['g_sq'], out['d_sq'] = g_sq, d_sq
    return out


def stft_pow_db(x, n_fft=2048):
    """10*log10(|STFT|^2 + 10e-20) of model.py:640-653: rectangular window (window=None),
    win_length 320 centred in n_fft, hop 160, center/reflect padding, normalized."""
    st = torch.stft(x.squeeze(1), n_fft=min(x.size(-1), n_fft), hop_length=160, win_length=320,
                    normalized=True, return_complex=True)
    return 10 * torch.log10(st.abs() ** 2 + 10e-20)


def interf_squares(n, T):
    """The interference pair's square waves, model.py:606-623: per sample one
    random.choice(freqs) and one random.choice(amps) (python `random`, in that order),
    a * scipy.signal.square(2 pi f t) on t = linspace(0, 2, 32000), cut to T samples."""
    import random
    import numpy as np
    from scipy import signal
    freqs, amps = [250, 1000, 4000], [0.01, 0.05, 0.1, 1]
    t = np.linspace(0, 2, 32000)
    out = []
    for _ in range(n):
        f_ = random.choice(freqs)
        a_ = random.choice(amps)
        sq = a_ * signal.square(2 * np.pi * f_ * t)
        out.append(torch.FloatTensor(sq[:T].reshape((1, -1))))
    return torch.cat(out, dim=0).unsqueeze(1)


def wsegan_step(g_sd, d_sd, clean, noisy, z, rolls, perm, names, strides, l1_weight=100.0,
                pow_weight=0.001, lr=5e-5, n_fft=2048, g_sq=None, d_sq=None, squares=None,
                vanilla_gan=False):
    """One WSEGAN step, model.py:577-669 (LSGAN cost; `vanilla_gan`: the BCE-with-logits cost of
    model.py:582-585 for every adversarial term).  rolls: the roll lists in call order
    (D real, D fake, [D misaligned,] [D interference,] D fake-for-G); perm: the batch
    permutation random.shuffle produced for --misalign_pair (model.py:598-600) or None;
    squares: the --interf_pair square waves [B, 1, T] (model.py:606-628) or None.  The weight
    of the summed D loss follows the reference literally: 1/2, 1/3 with the misaligned pair,
    1/4 with the interference pair (with or without the misaligned one)."""
    G = _leafs(g_sd)
    D = _leafs(d_sd)
    B = clean.size(0)
    rolls = list(rolls)
    ones, zeros = torch.ones(B, 1, dtype=clean.dtype), torch.zeros(B, 1, dtype=clean.dtype)
    d_real = discriminator_forward(D, torch.cat((clean, noisy), 1), rolls.pop(0), strides)
    Genh = generator_forward(G, noisy, z, strides)
    d_fake = discriminator_forward(D, torch.cat((Genh.detach(), noisy), 1), rolls.pop(0), strides)
    cost = F.binary_cross_entropy_with_logits if vanilla_gan else F.mse_loss
    d_loss = cost(d_fake, zeros) + cost(d_real, ones)
    d_weight = 0.5
    if perm is not None:
        d_shuf = discriminator_forward(D, torch.cat((clean, clean[perm]), 1), rolls.pop(0), strides)
        d_loss = d_loss + cost(d_shuf, zeros)
        d_weight = 1 / 3
    if squares is not None:
        d_int = discriminator_forward(D, torch.cat((clean + squares, noisy), 1), rolls.pop(0),
                                      strides)
        d_loss = d_loss + cost(d_int, zeros)
        d_weight = 1 / 4
    d_loss = d_weight * d_loss
    dkeys = [k for k in D if _is_param(k)]
    dgr = torch.autograd.grad(d_loss, [D[k] for k in dkeys])
    d_grads = {k: g for k, g in zip(dkeys, dgr)}
    d_sq = d_sq or {k: torch.zeros_like(D[k]) for k in dkeys}
    with torch.no_grad():
        for k, g in zip(dkeys, dgr):
            rmsprop_update(D[k], g, d_sq[k], lr)
    d_fake_ = discriminator_forward(D, torch.cat((Genh, noisy), 1), rolls.pop(0), strides)
    g_adv = cost(d_fake_, ones)
    pow_loss = pow_weight * F.l1_loss(stft_pow_db(Genh, n_fft), stft_pow_db(clean, n_fft))
    mask = torch.zeros(B, 1, Genh.size(2), dtype=clean.dtype)
    for i, n in enumerate(names):
        if 'additive' in n:
            mask[i, 0, :] = 1.0
    den_loss = l1_weight * F.l1_loss(Genh * mask, clean * mask)
    gkeys = [k for k in G if G[k].requires_grad]
    ggr = torch.autograd.grad(g_adv + pow_loss + den_loss, [G[k] for k in gkeys])
    g_grads = {k: g for k, g in zip(gkeys, ggr)}
    g_sq = g_sq or {k: torch.zeros_like(G[k]) for k in gkeys}
    with torch.no_grad():
        for k, g in zip(gkeys, ggr):
            rmsprop_update(G[k], g, g_sq[k], lr)
    return {'G': {k: v.detach() for k, v in G.items()}, 'D': {k: v.detach() for k, v in D.items()},
            'g_sq': g_sq, 'd_sq': d_sq, 'd_loss': d_loss.detach(), 'g_adv': g_adv.detach(),
            'pow_loss': pow_loss.detach(), 'den_loss': den_loss.detach(), 'Genh': Genh.detach(),
            'd_grads': d_grads, 'g_grads': g_grads}
