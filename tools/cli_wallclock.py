#!/usr/bin/env python3
"""Wall clock of the command line on a 128-frame 1080p .yuv on LOCAL DISK (GPU box): what `python -m aivc_amd.aivc` costs
end to end -- file read, host -> device, encode, bitstream write, bitstream read, decode, device -> host, .yuv write --
next to bench.py's HBM-resident figure (planar-YUV I/O replaces the reference's PNG triplets and forks,
src/real_life/encode.py:99-165, src/real_life/decode.py:101-147).  Model build (synthetic stand-in: construction +
calibration) is timed apart: once excluded (a warm process, the service case), once included (a cold command line).
usage: tools/cli_wallclock.py [out.json] [frames] [width] [height]"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)

import aivc_amd  # noqa: E402,F401
import torch  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
    h = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
    from aivc_amd import cli_common, synth
    from aivc_amd.real_life import decode as rdec
    from aivc_amd.real_life import encode as renc
    from bench import gpu_synthetic_unit
    dev = torch.device('cuda:0')
    td = tempfile.mkdtemp(prefix='aivc_cli_')
    raw = os.path.join(td, 'clip_%dx%d_30_420.yuv' % (w, h))
    with open(raw, 'wb') as f:
        for fr in gpu_synthetic_unit(w, h, n, 0, dev, 666):
            for k in 'yuv':
                f.write(fr[k].cpu().numpy().tobytes())
    torch.cuda.synchronize()
    bits, dec = os.path.join(td, 'bits.bin'), os.path.join(td, 'dec.yuv')
    res = {'frames': n, 'width': w, 'height': h, 'yuv_bytes': os.path.getsize(raw), 'tmp_filesystem': td}

    t0 = time.time()
    model = cli_common.get_model('absent', dev)
    torch.cuda.synchronize()
    res['model_build_s'] = round(time.time() - t0, 3)

    decoder = rdec.Decoder({'full_net': model}).eval()

    def run(tag):
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            t0 = time.time()
            renc.encode({'model': model, 'sequence_path': raw, 'GOP_struct_name': '1_GOP_32', 'final_file': bits,
                         'idx_starting_frame': 0, 'idx_end_frame': n - 1})
            torch.cuda.synchronize()
            t1 = time.time()
            rdec.decode_one_video({'decoder': decoder, 'bitstream_path': bits, 'out_file': dec, 'device': str(dev)})
            torch.cuda.synchronize()
            t2 = time.time()
        res[tag] = {'encode_s': round(t1 - t0, 3), 'decode_s': round(t2 - t1, 3), 'encode_fps': round(n / (t1 - t0), 2),
                    'decode_fps': round(n / (t2 - t1), 2), 'encode_plus_decode_fps': round(n / (t2 - t0), 2)}

    run('first_call')   # (includes one-off kernel-parameter packing, pinned buffers, hipMalloc growth)
    run('warm_call')
    res['bitstream_bytes'] = os.path.getsize(bits)
    res['decoded_bytes'] = os.path.getsize(dec)
    wc = res['warm_call']
    res['warm_with_model_build_fps'] = round(n / (wc['encode_s'] + wc['decode_s'] + 2 * res['model_build_s']), 2)
    res['note'] = ('encode = read .yuv from local disk + H2D + FrameCodec.encode_video + write the container; decode = read the '
                   'container + FrameCodec.decode_video + D2H + write .yuv; warm_with_model_build_fps charges a model build to each of '
                   'the two commands (encode.py and decode.py are separate processes in the reference)')
    line = json.dumps(res)
    print(line)
    if out_path:
        with open(out_path, 'w') as f:
            f.write(line + '\n')
    for p in (raw, bits, dec):
        os.remove(p)
    os.rmdir(td)


if __name__ == '__main__':
    main()
