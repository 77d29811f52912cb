"""Model base class and checkpoint Saver.

Behavioural mirror of the reference's ``segan/models/core.py`` (Saver: lines 11-151,
Model: lines 154-207): same file names (``weights_<prefix><Name>-<step>.ckpt``), same
JSON index (``<prefix>checkpoints`` with ``latest`` / ``current``), same rotation,
same ``{'step', 'state_dict', 'optimizer'}`` payload and the same partial
``load_pretrained`` rules, so checkpoints move freely between the two code bases.
"""
import atexit
import json
import os
import threading
import weakref

import torch
import torch.nn as nn

_pending = weakref.WeakSet()      # savers with a write in flight (flushed at interpreter exit)


@atexit.register
def _flush_pending():
    for sv in list(_pending):
        sv.wait()


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
    """Checkpoint writer / reader in the reference's format (core.py:11-151).

    Writing is ASYNCHRONOUS (`async_save`, default on): `save` only snapshots the state — for
    CUDA tensors a device-side clone on the training stream (0.7 GB of weights + optimizer state
    at ~3 TB/s: a fraction of a millisecond) followed by a device-to-host copy into pinned
    buffers on a side stream — and returns; a writer thread waits for the copy, serialises with
    torch.save into a temporary file and renames it into place.  The training loop goes on while
    the reference's synchronous 0.7 GB torch.save (core.py:61-70) would block it; the file
    holds what the synchronous path writes (same keys, tensors, `state_dict._metadata`).  The index
    file is rewritten — and the rotated-out checkpoint deleted — only after the weights file is in
    place.  `wait()` joins the writer (called before the next save, before loading, and at
    interpreter exit)."""

    def __init__(self, model, save_path, max_ckpts=5, optimizer=None, prefix='', async_save=True):
        self.model = model
        self.save_path = save_path
        self.ckpt_path = os.path.join(save_path, '{}checkpoints'.format(prefix))
        self.max_ckpts = max_ckpts
        self.optimizer = optimizer
        self.prefix = prefix
        self.async_save = async_save
        self._writer = None
        self._error = None
        self._pinned = {}       # path in the payload -> pinned host buffer, re-used across saves
        self._stream = None

    # ---- asynchronous writing -----------------------------------------------------------
    def wait(self):
        """Block until the checkpoint being written (if any) is on disk."""
        w, self._writer = self._writer, None
        if w is not None:
            w.join()
        _pending.discard(self)
        if self._error is not None:
            err, self._error = self._error, None
            raise err

    def _snapshot(self, obj, path, events):
        """`obj` with every tensor replaced by a host copy that nothing else writes to: CUDA
        tensors are cloned on the current stream, then copied to a pinned buffer on the side
        stream (asynchronously: `events` collects what to wait for); CPU tensors are cloned."""
        if torch.is_tensor(obj):
            if not obj.is_cuda:
                return obj.detach().clone()
            src = obj.detach().clone()                      # device-side, ordered on the main stream
            buf = self._pinned.get(path)
            if buf is None or buf.shape != src.shape or buf.dtype != src.dtype:
                buf = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
                self._pinned[path] = buf
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=src.device)
            self._stream.wait_stream(torch.cuda.current_stream(src.device))
            with torch.cuda.stream(self._stream):
                buf.copy_(src, non_blocking=True)
                src.record_stream(self._stream)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            events.append(ev)
            return buf
        if isinstance(obj, dict):
            out = type(obj)((k, self._snapshot(v, path + (k,), events)) for k, v in obj.items())
            if hasattr(obj, '_metadata'):      # nn.Module.state_dict()'s per-module versions
                out._metadata = obj._metadata
            return out
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._snapshot(v, path + (i,), events) for i, v in enumerate(obj))
        return obj

    def _write(self, payload, events, final, index, victim):
        try:
            for ev in events:
                ev.synchronize()
            # the pinned buffers are re-used by the next save: serialise from plain copies of
            # their storage only if a save could start before this one ends — it cannot (wait()
            # precedes every snapshot), so they are written directly
            tmp = final + '.tmp'
            torch.save(payload, tmp)
            os.replace(tmp, final)
            self._commit(index, victim)
        except BaseException as e:      # surfaced by the next wait()
            self._error = e

    def _commit(self, index, victim):
        """Publish a checkpoint whose weights file is complete: rewrite the index (atomically) so
        that `current` never names a file that is not on disk — a crash mid-write leaves the
        previous index and the previous checkpoint — and only then drop the rotated-out file."""
        tmp = self.ckpt_path + '.tmp'
        with open(tmp, 'w') as f:
            f.write(json.dumps(index, indent=2))
        os.replace(tmp, self.ckpt_path)
        if victim is not None:
            try:
                print('Removing old ckpt {}'.format(victim))
                os.remove(victim)
            except FileNotFoundError:
                print('ERROR: ckpt is not there?')

    def _read_index(self):
        if os.path.exists(self.ckpt_path):
            with open(self.ckpt_path, 'r') as f:
                return json.load(f)
        return {'latest': [], 'current': []}

    def save(self, model_name, step, best_val=False):
        self.wait()                 # one write in flight per saver
        os.makedirs(self.save_path, exist_ok=True)
        index = self._read_index()
        fname = '{}-{}.ckpt'.format(model_name, step)
        if best_val:
            fname = 'best_' + fname
        fname = '{}{}'.format(self.prefix, fname)
        latest = index['latest']
        # rotate: drop the oldest once more than max_ckpts are listed (core.py:40-51) — after the
        # new file is complete (the reference rewrites the index and deletes first, core.py:40-60;
        # with training going on during the write, that order would leave `current` dangling on a
        # crash)
        final = os.path.join(self.save_path, 'weights_' + fname)
        # a name that is already listed (a re-run in the same save_path restarts at iteration 1, so
        # the '<name>-<step>' names repeat) is overwritten in place: it moves to the end of the list
        # and is never its own victim — the deletion happens AFTER the new file is written
        latest = [n for n in latest if n != fname]
        victim = None
        if latest and self.max_ckpts is not None and len(latest) > self.max_ckpts:
            victim = os.path.join(self.save_path, 'weights_' + latest[0])
            if os.path.exists(victim):
                latest = latest[1:]
            else:
                print('ERROR: ckpt is not there?')
                victim = None
        if victim == final:
            victim = None
        latest = latest + [fname]
        index['latest'] = latest
        index['current'] = fname
        payload = {'step': step, 'state_dict': self.model.state_dict()}
        if self.optimizer is not None:
            payload['optimizer'] = self.optimizer.state_dict()
        if not self.async_save:
            torch.save(payload, final)
            self._commit(index, victim)
            return
        events = []
        snap = self._snapshot(payload, (), events)
        self._writer = threading.Thread(target=self._write, args=(snap, events, final, index, victim),
                                        daemon=True)
        _pending.add(self)
        self._writer.start()

    def read_latest_checkpoint(self):
        self.wait()
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
        path = os.path.join(self.save_path, 'weights_' + curr)
        if not os.path.exists(path):
            # an index written by the reference (or an older build) before its weights file was
            # complete: fall back to the newest listed checkpoint that exists
            for cand in reversed(self._read_index().get('latest', [])):
                if os.path.exists(os.path.join(self.save_path, 'weights_' + cand)):
                    print('[!] {} is missing, loading {} instead'.format(curr, cand))
                    path = os.path.join(self.save_path, 'weights_' + cand)
                    break
        st = torch.load(path, map_location='cpu')
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


def purge_checkpoints(ckpt_dir, verbose=True):
    """Keep only the newest checkpoint of every index file in `ckpt_dir` (what the reference's
    purge_ckpts.py:7-29 does): for each ``*checkpoints`` index, check that every listed
    ``weights_<name>`` exists, delete all but the last entry of ``latest`` (and never the one
    named by ``current``), rewrite the index with that single entry.  Half-written ``.tmp``
    files of an interrupted asynchronous save are removed too.  Returns the removed paths."""
    import glob
    removed = []
    for index in sorted(glob.glob(os.path.join(ckpt_dir, '*checkpoint*'))):
        if index.endswith('.tmp') or os.path.basename(index).startswith('weights_'):
            continue
        with open(index, 'r') as f:
            log = json.load(f)
        latest = list(log.get('latest', []))
        if not latest:
            continue
        missing = [n for n in latest if not os.path.exists(os.path.join(ckpt_dir, 'weights_' + n))]
        if missing:
            raise FileNotFoundError('{} lists checkpoints that are not on disk: {}'.format(index, missing))
        for name in latest[:-1]:
            if name == log.get('current'):
                continue
            path = os.path.join(ckpt_dir, 'weights_' + name)
            os.unlink(path)
            removed.append(path)
            if verbose:
                print('Removed file ', path)
        if verbose:
            print('Kept file ', os.path.join(ckpt_dir, 'weights_' + latest[-1]))
        log['latest'] = [latest[-1]]
        with open(index, 'w') as f:
            f.write(json.dumps(log, indent=2))
    for tmp in glob.glob(os.path.join(ckpt_dir, 'weights_*.tmp')):
        os.unlink(tmp)
        removed.append(tmp)
    return removed


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

    def wait_for_checkpoints(self):
        """Join the asynchronous checkpoint writer of this model's own saver (Saver.wait)."""
        if hasattr(self, 'saver'):
            self.saver.wait()

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

    def all_params(self):
        """Every parameter of the network (trainable or not) as a list, cached: the autograd nodes take
        it on every forward, and walking the module tree (nn.Module.parameters) per call was a tenth of
        the launch path's python time.  Dropped whenever the module tree can have changed: .to() /
        .cuda() / .float() (`_apply`), load_state_dict, add_module / register_parameter on this module,
        train() / eval()."""
        ps = self.__dict__.get('_all_params')
        if ps is None:
            ps = self.__dict__['_all_params'] = list(nn.Module.parameters(self))
        return ps

    def _drop_param_cache(self):
        self.__dict__.pop('_all_params', None)

    def _apply(self, fn, *a, **k):
        self._drop_param_cache()
        return super()._apply(fn, *a, **k)

    def add_module(self, name, module):
        self._drop_param_cache()
        return super().add_module(name, module)

    def register_parameter(self, name, param):
        self._drop_param_cache()
        return super().register_parameter(name, param)

    def train(self, mode=True):
        self._drop_param_cache()
        return super().train(mode)

    def get_n_params(self):
        return sum(p.numel() for p in self.parameters())

    def load_state_dict(self, state_dict, *args, **kwargs):
        self._drop_param_cache()
        out = super().load_state_dict(state_dict, *args, **kwargs)
        from .. import ops
        ops.bump_weights_epoch()
        return out
