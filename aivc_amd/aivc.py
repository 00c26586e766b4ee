#!/usr/bin/env python3
"""aivc.py CLI (flags of src/aivc.py:16-76): encode, decode, evaluate (PSNR, MS-SSIM, size) -- in one process
(the reference forks three python processes and goes through PNG triplets)."""
import argparse
import os
import sys

from aivc_amd import decode as dec_cli
from aivc_amd import encode as enc_cli
from aivc_amd import evaluate as eval_cli


def gop_name(cfg, gop_size, intra_period):
    """src/aivc.py:80-107"""
    if cfg == 'AI':
        return '1_GOP_0'
    if cfg == 'LDP':
        if intra_period not in range(2, 65535):
            sys.exit('[ERROR]: Intra period should be in [2, 65535] for LDP.')
        return 'LDP_%d' % intra_period
    if cfg == 'RA':
        if intra_period % gop_size:
            sys.exit('[ERROR]: Intra period must be equal a multiple of GOP size')
        return '%d_GOP_%d' % (intra_period // gop_size, gop_size)
    sys.exit('[ERROR]: unknown coding configuration. Should be either RA, AI or LDP.')


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--coding_config', default='RA', type=str)
    p.add_argument('--gop_size', default=32, type=int)
    p.add_argument('--intra_period', default=32, type=int)
    p.add_argument('--model', default='ms_ssim-2021cc-3', type=str)
    p.add_argument('-i', default='../raw_videos/BlowingBubbles_416x240_50_420.yuv', type=str)
    p.add_argument('--start_frame', default=0, type=int)
    p.add_argument('--end_frame', default=90, type=int)
    p.add_argument('--bitstream_out', default='../bitstream.bin', type=str)
    p.add_argument('-o', default='../compressed.yuv', type=str)
    p.add_argument('--rng_seed', default=666, type=int)
    p.add_argument('--cpu', action='store_true')
    a = p.parse_args(argv)
    gop = gop_name(a.coding_config, a.gop_size, a.intra_period)
    common = ['--model', a.model] + (['--cpu'] if a.cpu else [])
    import os
    banner = print if int(os.environ.get('RANK', '0')) == 0 else (lambda *x: None)  # one voice in a multi-rank job
    banner(('*' * 80).center(120))
    banner('Starting encoding'.center(120))
    enc_cli.main(['-i', a.i, '--gop', gop, '--start_frame', str(a.start_frame), '--end_frame', str(a.end_frame),
                  '-o', a.bitstream_out] + common)
    banner(('*' * 80).center(120))
    banner('Starting decoding'.center(120))
    status = dec_cli.main(['-i', a.bitstream_out, '-o', a.o] + common)
    from aivc_amd import parallel
    if parallel.rank_world()[0] != 0:  # multi-rank job: rank 0 evaluates
        return status
    print(('*' * 80).center(120))
    print('Starting evaluation'.center(120))
    eval_cli.main(['--raw', a.i, '--compressed', a.o, '--bitstream', a.bitstream_out, '--start_frame', str(a.start_frame)])
    return status  # 3: decoded and evaluated, but the stream did not decode cleanly (real_life/decode.py)


def cli():
    raise SystemExit(main())


if __name__ == '__main__':
    cli()
