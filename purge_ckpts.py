"""Keep only the newest checkpoint per index in a checkpoint directory (same command line as the
reference's purge_ckpts.py: ``python purge_ckpts.py <ckpt_dir>``)."""
import argparse

from segan_pytorch_amd.models.core import purge_checkpoints

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('ckpt_dir', type=str)
    purge_checkpoints(ap.parse_args().ckpt_dir)
