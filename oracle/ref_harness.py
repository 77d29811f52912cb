"""Import the REAL reference (santi-pdp/segan_pytorch at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY, and only usable in the build container (the GPU box has no
/root/reference): it is how ``oracle/make_golden.py`` generates the fixtures that pin
``oracle/segan_oracle.py``.  The seven I/O / logging packages the reference imports but
the hot path never uses are stubbed with empty modules.
"""
import os
import random
import sys
import types

import torch

REF_ROOT = os.environ.get('SEGAN_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'segan', 'models'))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _NoWriter(object):
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass


def import_reference():
    """Returns the reference's ``segan.models`` package (SEGAN, Generator, ...)."""
    if not available():
        raise RuntimeError('reference not found at {}'.format(REF_ROOT))
    torch.backends.mkldnn.enabled = False
    _stub('torchvision')
    _stub('torchvision.utils')
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    _stub('tensorboardX', SummaryWriter=_NoWriter)
    _stub('ahoproc_tools')
    _stub('ahoproc_tools.io', read_aco_file=None, aco2wav=None, wav2aco=None)
    _stub('ahoproc_tools.interpolate', interpolation=None)
    _stub('librosa')
    _stub('h5py')
    _stub('soundfile')
    ident = lambda *a, **k: (a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f))
    _stub('numba', jit=ident, int32=None, float32=None)
    # the reference package is called `segan`; keep it out of the way of ours
    saved = {k: v for k, v in sys.modules.items() if k == 'segan' or k.startswith('segan.')}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REF_ROOT)
    try:
        import segan.models as ref_models  # noqa
        import segan.models.model as ref_model  # noqa
    finally:
        sys.path.remove(REF_ROOT)
    return ref_models


class ReplayRandom(object):
    """Record / replay the python `random` draws of Discriminator.forward
    (discriminator.py:159-163) as signed rolls."""

    def __init__(self, seed):
        self.seed = seed

    def rolls(self, n_layers, phase_shift, n_forwards):
        random.seed(self.seed)
        out = []
        for _ in range(n_forwards):
            r = []
            for _ in range(n_layers):
                s = random.randint(1, phase_shift)
                right = random.random() > 0.5
                r.append(s if right else -s)
            out.append(r)
        return out
