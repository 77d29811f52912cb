"""Rewrite an old SEGAN generator checkpoint (blocks named gen_enc.i.conv.* / gen_dec.i.conv.*)
with today's names (enc_blocks.i.conv.* / dec_blocks.i.deconv.*) into ``<file>.v2`` — same
command line and output file as the reference's weightG_fmt_converter.py:

    python weightG_fmt_converter.py <weights ckpt file>

Every top-level entry other than ``state_dict`` is carried over unchanged.  (Loading does not
need this step: ``Model.load_pretrained`` applies the same mapping on the fly.)
"""
import sys

import torch

from segan_pytorch_amd.models.core import convert_legacy_generator_keys


def convert(ckpt_file, out_file=None, verbose=True):
    out_file = out_file or ckpt_file + '.v2'
    ckpt = torch.load(ckpt_file, map_location='cpu', weights_only=False)
    old = ckpt['state_dict']
    new = convert_legacy_generator_keys(old)
    if verbose:
        for ko, kn in zip(old.keys(), new.keys()):
            print('{} -> {}'.format(ko, kn) if ko != kn else 'Keeping {}'.format(ko))
    out = {k: v for k, v in ckpt.items() if 'state_dict' not in k}
    out['state_dict'] = new
    torch.save(out, out_file)
    return out_file


if __name__ == '__main__':
    if len(sys.argv) < 2:
        print('ERROR! Not enough input arguments.')
        print('Usage: {} <weights ckpt file> .'.format(sys.argv[0]))
        sys.exit(1)
    convert(sys.argv[1])
