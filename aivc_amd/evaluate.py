#!/usr/bin/env python3
"""evaluate.py CLI (src/evaluate.py): the CLIC-2021 figures of a decoded video and the size of its bitstream --

    PSNR    [dB]: ...      MS-SSIM     : ...      MS-SSIM [dB]: ...      Size [bytes]: ...

The reference walks two folders of <idx>_{y,u,v}.png planes; here --raw / --compressed are planar 8-bit 4:2:0
.yuv files (what encode.py reads and decode.py writes -- no PNG round trip) and the three planes of every
decoded frame are scored on the GPU (aivc_amd/clic21/metrics.py)."""
import argparse
import os

import numpy as np

from aivc_amd.clic21.metrics import evaluate
from aivc_amd.real_life.encode import parse_yuv_name


def _planes(path, w, h, first=0, count=None):
    hc, wc = (h + 1) // 2, (w + 1) // 2
    fsize = h * w + 2 * hc * wc
    n = os.path.getsize(path) // fsize - first
    n = n if count is None else min(n, count)
    buf = np.fromfile(path, np.uint8, count=n * fsize, offset=first * fsize).reshape(n, fsize)
    return [{'y': f[:h * w].reshape(h, w), 'u': f[h * w:h * w + hc * wc].reshape(hc, wc),
             'v': f[h * w + hc * wc:].reshape(hc, wc)} for f in buf]


def evaluate_yuv(raw_path, compressed_path, start_frame=0):
    """frame size from the raw file's name (<Name>_<W>x<H>_..., as encode.py parses it)"""
    w, h = parse_yuv_name(raw_path)
    dec = _planes(compressed_path, w, h)
    raw = _planes(raw_path, w, h, start_frame, len(dec))
    target, submit = {}, {}
    for idx, (r, d) in enumerate(zip(raw, dec)):
        for c in 'yuv':
            target['%d_%s' % (idx, c)] = r[c]
            submit['%d_%s' % (idx, c)] = d[c]
    return evaluate(submit, target)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--raw', default='', type=str, help='planar 4:2:0 .yuv file with the input frames')
    p.add_argument('--compressed', default='', type=str, help='planar 4:2:0 .yuv file written by decode.py')
    p.add_argument('--bitstream', default='../bitstream.bin', type=str, help='Path of the bitstream')
    p.add_argument('--start_frame', default=0, type=int, help='index of the first coded frame inside --raw')
    a = p.parse_args(argv)
    comp = a.compressed if a.compressed.endswith('.yuv') else a.compressed + '.yuv'
    results = evaluate_yuv(a.raw, comp, a.start_frame)
    print('PSNR    [dB]: ' + '%.5f' % (results.get('PSNR')))
    print('MS-SSIM     : ' + '%.5f' % (results.get('MSSSIM')))
    print('MS-SSIM [dB]: ' + '%.5f' % (results.get('MSSSIM_dB')))
    try:
        print('Size [bytes]: ' + '%.0f' % os.path.getsize(a.bitstream))
    except FileNotFoundError:
        print('[ERROR]: bitstream not found, can not evaluate its size!')
        print('Bistream path: ' + a.bitstream)
    return results


if __name__ == '__main__':
    main()
