"""Model base class and checkpoint Saver.

Behavioural mirror of the reference's ``segan/models/core.py`` (Saver: lines 11-151,
Model: lines 154-207): same file names (``weights_<prefix><Name>-<step>.ckpt``), same
JSON index (``<prefix>checkpoints`` with ``latest`` / ``current``), same rotation,
same ``{'step', 'state_dict', 'optimizer'}`` payload and the same partial
``load_pretrained`` rules, so checkpoints move freely between the two code bases.
"""
import json
import os

import torch
import torch.nn as nn


def convert_legacy_generator_keys(state_dict):
    """Old SEGAN-G checkpoints name the blocks gen_enc.i.conv.* / gen_dec.i.conv.*; the
    reference ships weightG_fmt_converter.py:18-44 to rewrite them offline.  Done on the fly
    here so such a checkpoint loads directly: gen_enc -> enc_blocks, gen_dec -> dec_blocks
    with conv -> deconv.  Current-format dicts pass through unchanged."""
    if not any(('gen_enc' in k or 'gen_dec' in k) for k in state_dict):
        return state_dict
    out = type(state_dict)()
    for k, v in state_dict.items():
        if 'gen_enc' in k:
            k = k.replace('gen_enc', 'enc_blocks')
        elif 'gen_dec' in k:
            k = k.replace('gen_dec', 'dec_blocks').replace('conv', 'deconv')
        out[k] = v
    return out


class Saver(object):

    def __init__(self, model, save_path, max_ckpts=5, optimizer=None, prefix=''):
        self.model = model
        self.save_path = save_path
        self.ckpt_path = os.path.join(save_path, '{}checkpoints'.format(prefix))
        self.max_ckpts = max_ckpts
        self.optimizer = optimizer
        self.prefix = prefix

    def _read_index(self):
        if os.path.exists(self.ckpt_path):
            with open(self.ckpt_path, 'r') as f:
                return json.load(f)
        return {'latest': [], 'current': []}

    def save(self, model_name, step, best_val=False):
        os.makedirs(self.save_path, exist_ok=True)
        index = self._read_index()
        fname = '{}-{}.ckpt'.format(model_name, step)
        if best_val:
            fname = 'best_' + fname
        fname = '{}{}'.format(self.prefix, fname)
        latest = index['latest']
        # rotate: drop the oldest once more than max_ckpts are listed (core.py:40-51)
        if latest and self.max_ckpts is not None and len(latest) > self.max_ckpts:
            victim = os.path.join(self.save_path, 'weights_' + latest[0])
            try:
                print('Removing old ckpt {}'.format(victim))
                os.remove(victim)
                latest = latest[1:]
            except FileNotFoundError:
                print('ERROR: ckpt is not there?')
        latest = latest + [fname]
        index['latest'] = latest
        index['current'] = fname
        with open(self.ckpt_path, 'w') as f:
            f.write(json.dumps(index, indent=2))
        payload = {'step': step, 'state_dict': self.model.state_dict()}
        if self.optimizer is not None:
            payload['optimizer'] = self.optimizer.state_dict()
        torch.save(payload, os.path.join(self.save_path, 'weights_' + fname))

    def read_latest_checkpoint(self):
        print('Reading latest checkpoint from {}...'.format(self.ckpt_path))
        if not os.path.exists(self.ckpt_path):
            print('[!] No checkpoint found in {}'.format(self.save_path))
            return False
        return self._read_index()['current']

    def load_weights(self):
        curr = self.read_latest_checkpoint()
        if curr is False:
            print('[!] No weights to be loaded')
            return False
        st = torch.load(os.path.join(self.save_path, 'weights_' + curr), map_location='cpu')
        if 'state_dict' in st:
            self.model.load_state_dict(st['state_dict'])
            if self.optimizer is not None and 'optimizer' in st:
                self.optimizer.load_state_dict(st['optimizer'])
        else:
            self.model.load_state_dict(st)     # legacy: bare state_dict
        print('[*] Loaded weights')
        return True

    def load_pretrained_ckpt(self, ckpt_file, load_last=False, load_opt=True):
        model_dict = self.model.state_dict()
        st = torch.load(ckpt_file, map_location='cpu')
        pt_dict = st['state_dict'] if 'state_dict' in st else st
        pt_dict = convert_legacy_generator_keys(pt_dict)
        keys = list(pt_dict.keys())
        allowed = keys[:] if load_last else keys[:-2]     # core.py:131-135
        pt_dict = {k: v for k, v in pt_dict.items()
                   if k in model_dict and k in allowed and v.size() == model_dict[k].size()}
        print('Current Model keys: ', len(model_dict))
        print('Loading Pt Model keys: ', len(pt_dict))
        print('Loading matching keys: ', list(pt_dict.keys()))
        if len(pt_dict) != len(model_dict):
            print('WARNING: LOADING DIFFERENT NUM OF KEYS')
        model_dict.update(pt_dict)
        self.model.load_state_dict(model_dict)
        for k in model_dict.keys():
            if k not in allowed:
                print('WARNING: {} weights not loaded from pt ckpt'.format(k))
        if self.optimizer is not None and 'optimizer' in st and load_opt:
            self.optimizer.load_state_dict(st['optimizer'])


class Model(nn.Module):

    def __init__(self, name='BaseModel'):
        super().__init__()
        self.name = name
        self.optim = None

    def save(self, save_path, step, best_val=False, saver=None):
        if saver is None:
            if not hasattr(self, 'saver'):
                self.saver = Saver(self, save_path, optimizer=self.optim,
                                   prefix=self.name + '-')
            self.saver.save(self.name, step, best_val=best_val)
        else:
            saver.save(self.name, step, best_val=best_val)

    def load(self, save_path):
        if os.path.isdir(save_path):
            if not hasattr(self, 'saver'):
                self.saver = Saver(self, save_path, optimizer=self.optim,
                                   prefix=self.name + '-')
            self.saver.load_weights()
        else:
            print('Loading ckpt from ckpt: ', save_path)
            self.load_pretrained(save_path)

    def load_pretrained(self, ckpt_path, load_last=False):
        Saver(self, '.', optimizer=self.optim).load_pretrained_ckpt(ckpt_path, load_last)

    def activation(self, name):
        return getattr(nn, name)()

    def parameters(self, recurse=True):
        # only trainable parameters, as the reference (core.py:196-197)
        return filter(lambda p: p.requires_grad, super().parameters(recurse))

    def get_n_params(self):
        return sum(p.numel() for p in self.parameters())

    def load_state_dict(self, state_dict, *args, **kwargs):
        out = super().load_state_dict(state_dict, *args, **kwargs)
        from .. import ops
        ops.bump_weights_epoch()
        return out
