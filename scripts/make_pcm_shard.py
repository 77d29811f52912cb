"""Cut a clean/noisy wav directory pair into a pre-sliced int16 shard for
`train.py --pcm_shard PREFIX` (format: segan_pytorch_amd/datasets.py:build_pcm_shard).
usage: python scripts/make_pcm_shard.py CLEAN_DIR NOISY_DIR OUT_PREFIX [--slice_size 16384] [--stride 0.5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segan_pytorch_amd.datasets import build_pcm_shard

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('clean_dir')
    ap.add_argument('noisy_dir')
    ap.add_argument('out_prefix')
    ap.add_argument('--slice_size', type=int, default=16384)
    ap.add_argument('--stride', type=float, default=0.5)
    ap.add_argument('--max_samples', type=int, default=None)
    a = ap.parse_args()
    n = build_pcm_shard(a.clean_dir, a.noisy_dir, a.out_prefix, a.slice_size, a.stride, a.max_samples)
    print('{} slices -> {}.pcm16 / .json'.format(n, a.out_prefix))
