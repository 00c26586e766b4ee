#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's own Python modules.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite use the
committed fixtures.  Nothing from the reference is copied: the fixtures hold seeded inputs, the
reference modules' randomly initialised parameters (state_dict tensors) and the outputs the
reference computes for them on CPU (torch fp32).

    python tools/gen_golden.py            # rewrites tests/golden/
"""
import io
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ['PYTHONDONTWRITEBYTECODE'] = '1'

import numpy as np
import torch

REF = '/root/reference/src'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def install_stubs():
    """torchvision and torchac are absent; the reference only needs two torchvision helpers at
    import time and never calls torchac in what we run."""
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvf = types.ModuleType('torchvision.transforms.functional')
    tvf.to_tensor = lambda img: torch.from_numpy(np.asarray(img, np.float32) / 255.)
    tvf.to_pil_image = lambda *a, **k: None
    tv.transforms = tvt
    tvt.functional = tvf
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tvt
    sys.modules['torchvision.transforms.functional'] = tvf
    sys.modules['torchac'] = types.ModuleType('torchac')


def sd_np(module, prefix='sd.'):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def perturb_gdn(module, gen):
    """move GDN beta/gamma away from their init so the re-parameterisation is exercised"""
    for m in module.modules():
        if type(m).__name__ == 'GDN':
            with torch.no_grad():
                m.beta.add_(torch.rand(m.beta.shape, generator=gen) * 0.5)
                m.gamma.add_(torch.rand(m.gamma.shape, generator=gen) * 0.05)
                # a few entries below the bounds
                m.gamma[0, -1] = 0.0
                m.beta[-1] = 0.0


def randomize(module, gen, scale=1.0):
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith('beta') or name.endswith('gamma'):
                continue
            p.copy_(torch.randn(p.shape, generator=gen) * (scale / max(1, p[0].numel()) ** 0.5))
    perturb_gdn(module, gen)


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print('%-34s %7.1f kB' % (name + '.npz', os.path.getsize(path) / 1e3))


def main():
    global OUT
    if len(sys.argv) > 2 and sys.argv[1] == '--out':
        OUT = sys.argv[2]
    torch.manual_seed(666)  # BallePdfEstim / GainMatrix draw their initial parameters from the global generator
    install_stubs()
    sys.path.insert(0, REF)
    os.makedirs(OUT, exist_ok=True)
    import func_util.console_display as cd
    cd.FLAG_QUIET = True
    from layers.misc.custom_conv_layers import CustomConvLayer, UpscalingLayer, ChengResBlock, ResBlock
    from layers.misc.misc_layers import GDN, Quantizer, PdfParamParameterizer
    from layers.misc.attention import SimplifiedAttention
    from layers.ae.ae_layers import InputLayer, OutputLayer
    from layers.entropy_coding.pdf_estimator import BallePdfEstim
    from layers.multi_rate.gain_matrix import GainMatrix
    from func_util.optical_flow import warp
    from func_util.img_processing import cast_before_png_saving
    from func_util.GOP_structure import generate_gop_struct
    from real_life.bitstream import ArithmeticCoder
    from real_life import header as ref_header
    from real_life import cat_binary_files as ref_cat

    gen = torch.Generator().manual_seed(666)

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, generator=gen) * scale

    # ---- a1: CustomConvLayer -------------------------------------------------------------
    cases = [dict(k_size=5, in_ft=3, out_ft=8, non_linearity='gdn', conv_stride=2, hw=(17, 23)),
             dict(k_size=3, in_ft=8, out_ft=8, non_linearity='leaky_relu', conv_stride=1, hw=(9, 11)),
             dict(k_size=3, in_ft=6, out_ft=12, non_linearity='relu', conv_stride=2, hw=(10, 14)),
             dict(k_size=5, in_ft=9, out_ft=4, non_linearity='no', conv_stride=1, hw=(8, 8)),
             dict(k_size=3, in_ft=8, out_ft=8, non_linearity='gdn_inverse', conv_stride=2, hw=(13, 7)),
             dict(k_size=5, in_ft=16, out_ft=32, non_linearity='gdn', conv_stride=2, hw=(21, 19))]
    for i, c in enumerate(cases):
        hw = c.pop('hw')
        m = CustomConvLayer(**c).eval()
        randomize(m, gen)
        x = rnd(1, c['in_ft'], *hw)
        with torch.no_grad():
            y = m(x)
        save('custom_conv_%d' % i, x=x, y=y, cfg=np.array(repr(c)), **sd_np(m))

    # ---- a2: UpscalingLayer --------------------------------------------------------------
    cases = [dict(k_size=5, in_ft=8, out_ft=6, non_linearity='gdn_inverse', hw=(7, 9)),
             dict(k_size=3, in_ft=8, out_ft=8, non_linearity='leaky_relu', hw=(5, 6)),
             dict(k_size=5, in_ft=12, out_ft=3, non_linearity='no', hw=(6, 5)),
             dict(k_size=3, in_ft=4, out_ft=4, non_linearity='relu', hw=(1, 3))]
    for i, c in enumerate(cases):
        hw = c.pop('hw')
        m = UpscalingLayer(**c).eval()
        randomize(m, gen)
        x = rnd(1, c['in_ft'], *hw)
        with torch.no_grad():
            y = m(x)
        save('upscaling_%d' % i, x=x, y=y, cfg=np.array(repr(c)), **sd_np(m))

    # ---- a3: residual compositions --------------------------------------------------------
    for i, (mode, hw) in enumerate([('plain', (9, 13)), ('down', (9, 13)), ('down', (10, 12)),
                                    ('up_tconv', (5, 7))]):
        m = ChengResBlock(8, mode=mode).eval()
        randomize(m, gen)
        x = rnd(1, 8, *hw)
        with torch.no_grad():
            y = m(x)
        save('cheng_%d' % i, x=x, y=y, cfg=np.array(repr(dict(nb_ft=8, mode=mode))), **sd_np(m))
    m = ResBlock(3, 8).eval()
    randomize(m, gen)
    x = rnd(1, 8, 7, 9)
    with torch.no_grad():
        y = m(x)
    save('resblock_0', x=x, y=y, cfg=np.array(repr(dict(k_size=3, nb_ft=8))), **sd_np(m))
    for i, light in enumerate([True, False]):
        m = SimplifiedAttention(8, lightweight_resblock=light).eval()
        randomize(m, gen)
        x = rnd(1, 8, 6, 10)
        with torch.no_grad():
            y = m(x)
        save('attention_%d' % i, x=x, y=y,
             cfg=np.array(repr(dict(nb_ft=8, lightweight_resblock=light))), **sd_np(m))

    # ---- a4: GDN alone ---------------------------------------------------------------------
    for i, inv in enumerate([False, True]):
        m = GDN(8, inverse=inv).eval()
        perturb_gdn(m, gen)
        x = rnd(1, 8, 5, 7)
        with torch.no_grad():
            y = m(x)
        save('gdn_%d' % i, x=x, y=y, cfg=np.array(repr(dict(ch=8, inverse=inv))),
             consts=np.array([m.beta_bound.item(), m.gamma_bound.item(), m.pedestal.item()], np.float32),
             **sd_np(m))

    # ---- a5 / a6: InputLayer, OutputLayer (+ pad/crop + cast as in Decoder.decode) ---------
    for i, (h, w) in enumerate([(9, 13), (10, 14), (7, 8)]):
        hc, wc = (h + 1) // 2, (w + 1) // 2
        lv = lambda *s: torch.randint(0, 256, s, generator=gen).float() / 255.
        d = {'y': lv(1, 1, h, w), 'u': lv(1, 1, hc, wc), 'v': lv(1, 1, hc, wc)}
        with torch.no_grad():
            x444 = InputLayer()(d)
            z = rnd(1, 3, h + 3, w + 2, scale=0.4) + 0.5
            z[0, :, :2, :2] = torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255]).view(2, 2)
            o = OutputLayer()(z[:, :, :h, :w])
            pad = torch.nn.ReplicationPad2d((0, abs(wc - o['u'].shape[3]), 0, abs(hc - o['u'].shape[2])))
            o = {'y': o['y'][:, :, :h, :w], 'u': pad(o['u'])[:, :, :hc, :wc], 'v': pad(o['v'])[:, :, :hc, :wc]}
            o = cast_before_png_saving({'x': o, 'data_type': 'yuv_dic'})
        save('inout_layer_%d' % i, y=d['y'], u=d['u'], v=d['v'], x444=x444, z=z, oy=o['y'], ou=o['u'],
             ov=o['v'])

    # ---- a7: warp ---------------------------------------------------------------------------
    for i, (h, w, s) in enumerate([(9, 13, 3.0), (8, 8, 20.0), (5, 1, 2.0)]):
        x = rnd(1, 3, h, w)
        flo = rnd(1, 2, h, w, scale=s)
        flo[0, :, 0, 0] = 0.0
        with torch.no_grad():
            y = warp(x, flo)
        save('warp_%d' % i, x=x, flow=flo, y=y)

    # ---- a10 / a11 / a12 ---------------------------------------------------------------------
    pp = PdfParamParameterizer('laplace', 6)
    x = rnd(1, 12, 5, 7, scale=8.0)
    x[0, 6, 0, 0], x[0, 6, 0, 1] = -30.0, 30.0
    with torch.no_grad():
        r = pp(x)
    save('pdf_param_0', x=x, mu=r[0]['mu'], sigma=r[0]['sigma'])
    gm = GainMatrix({'N': 3, 'nb_ft': 6, 'initialize_to_one': False}).eval()
    x = rnd(1, 6, 4, 5)
    outs = {}
    with torch.no_grad():
        for idx in (0, 1, 2, 0.5, 1.25):
            for mode in ('enc', 'dec'):
                outs['y_%s_%s' % (str(idx).replace('.', 'p'), mode)] = gm({'x': x, 'idx_rate': idx, 'mode': mode})['output']
    save('gain_matrix_0', x=x, **outs, **sd_np(gm))
    q = Quantizer().eval()
    x = torch.tensor([-2.5, -1.5, -0.5, 0.5, 1.5, 2.5, 0.49999997, 1.2, -3.7, 254.5, 255.5]).view(1, 1, 1, -1)
    with torch.no_grad():
        save('quantizer_0', x=x, y=q(x))

    # ---- a13: factorised prior CDF table -----------------------------------------------------
    for i, cz in enumerate([4, 7]):
        pe = BallePdfEstim(cz, 'balle', verbose=False)
        with torch.no_grad():
            for p in pe.parameters():
                p.mul_(1.5)
        ac = ArithmeticCoder({'balle_pdf_estim_z': pe, 'device': 'cpu'})
        save('balle_cdf_%d' % i, cdf=ac.pre_computed_z_cdf.detach().reshape(cz, 514), **sd_np(pe))

    # ---- a14: Laplace CDF over a sigma sweep ---------------------------------------------------
    sig = torch.cat([torch.logspace(-4, np.log10(148.41), 150), torch.tensor([1e-4, 148.41316, 1.0, 0.5])])
    sig = sig.view(1, 1, 1, -1)
    with torch.no_grad():
        cdf = ac.get_y_cdf(sig)
    # torchac's quantisation written with plain torch ops (its published normalisation formula)
    q16 = (cdf * (65536. - 513.)).round().to(torch.int32) + torch.arange(514, dtype=torch.int32)
    save('laplace_cdf_0', sigma=sig.reshape(-1), cdf=cdf.reshape(-1, 514), cdf_u16=(q16 & 0xFFFF).reshape(-1, 514).to(torch.int32))

    # ---- a18 / a20: container format + GOP structures -------------------------------------------
    gops = {}
    for name in ['1_GOP_0', 'LDP_8', '2_GOP_8', '1_GOP_32', '2_GOP_16', '1_GOP_2', 'LDP_2', '3_GOP_4']:
        g = generate_gop_struct(name)
        rows = []
        for f, d in g.items():
            ref_i = lambda s: -1 if s is None else int(s.split('_')[-1])
            rows.append([int(f.split('_')[-1]), d['type'], ref_i(d['prev_ref']), ref_i(d['next_ref']), d['coding_order']])
        gops['gop.' + name] = np.array(rows, np.int32)
    save('gop_struct', **gops)

    cont = {}
    with tempfile.TemporaryDirectory() as td:
        cwd = os.getcwd()
        os.chdir(td)
        try:
            rng = np.random.default_rng(1)
            # two GOPs of LDP_2 (3 frames each), frames 5..10
            bdir = os.path.join(td, 'bs') + '/'
            os.makedirs(bdir)
            data_dim = {'x': (48, 80), 'y': (3, 5), 'z': (1, 2)}
            frames = {}
            for g in range(2):
                for f in range(3):
                    idx = 5 + g * 3 + f
                    payload = rng.integers(0, 256, int(rng.integers(8, 60)), dtype=np.uint8).tobytes()
                    frames[idx] = payload
                    with open(bdir + str(idx), 'wb') as fo:
                        fo.write(payload)
                ref_header.write_gop_header({'header_path': bdir + str(g), 'idx_rate': 0.5 if g else 0.,
                                             'GOP_struct_name': 'LDP_2', 'data_dim': data_dim})
                ref_cat.cat_one_gop({'idx_gop': g, 'bitstream_dir': bdir})
                with open(bdir + str(g) + 'g', 'rb') as fi:
                    cont['gop_file_%d' % g] = np.frombuffer(fi.read(), np.uint8)
            ref_cat.cat_one_video({'bitstream_dir': bdir, 'idx_starting_frame': 5, 'idx_end_frame': 9,
                                   'final_bitstream_path': os.path.join(td, 'out', 'video.bin')})
            with open(os.path.join(td, 'out', 'video.bin'), 'rb') as fi:
                cont['video_file'] = np.frombuffer(fi.read(), np.uint8)
            for idx, payload in frames.items():
                cont['frame_%d' % idx] = np.frombuffer(payload, np.uint8)
            # RA header
            ref_header.write_gop_header({'header_path': bdir + '7', 'idx_rate': 0., 'GOP_struct_name': '2_GOP_16',
                                         'data_dim': data_dim})
            with open(bdir + '7h', 'rb') as fi:
                cont['gop_header_2_GOP_16'] = np.frombuffer(fi.read(), np.uint8)
            g2, r2 = ref_header.read_gop_header({'header_path': bdir + '7'})
            cont['gop_header_2_GOP_16_len'] = np.array([len(g2)])
        finally:
            os.chdir(cwd)
    save('container', **cont)

    # ---- a1-a5 at the hot-path widths (64 / 128 channels): parameters and inputs are seeded, not stored --------
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
    import wide_cases
    from layers.misc import custom_conv_layers as ref_ccl, attention as ref_att
    for name, build, kw, in_shape, seed, variants, force_tiles in wide_cases.CASES:
        if build == 'first_layer':
            m = CustomConvLayer(k_size=5, in_ft=3 * kw['n_img'], out_ft=64, non_linearity='gdn', conv_stride=2).eval()
            planes, sha = wide_cases.load_seeded(m, name)
            with torch.no_grad():
                dics = [{k: torch.from_numpy(p[k]).float().unsqueeze(0) / 255. for k in 'yuv'} for p in planes]
                y = m(torch.cat([InputLayer()(d) for d in dics], dim=1))
        else:
            cls = getattr(ref_att, build, None) or getattr(ref_ccl, build)
            m = cls(**kw).eval()
            x, sha = wide_cases.load_seeded(m, name)
            with torch.no_grad():
                y = m(torch.from_numpy(x))
        save('wide_' + name, y=y, sha256=np.array(sha), cfg=np.array(repr(dict(build=build, kw=kw, in_shape=in_shape, seed=seed))))
    # ---- rate estimation (logging): EntropyCoder / ParametricPdf / BallePdfEstim forward and the flag_debug figure ----
    from layers.entropy_coding.entropy_coder import EntropyCoder
    from layers.entropy_coding.pdf_estimator import ParametricPdf
    g2 = torch.Generator().manual_seed(4242)
    torch.manual_seed(4242)  # (BallePdfEstim draws from the global generator; nothing after this point uses it)
    c, h, w = 5, 6, 7
    sigma = torch.exp(torch.rand(1, c, h, w, generator=g2) * 9.0 - 5.0).clamp(1e-4, 148.0)
    mu = torch.randn(1, c, h, w, generator=g2) * 2.0
    y = torch.round(torch.randn(1, c, h, w, generator=g2) * sigma * 1.5 + mu).clamp(-256, 256)
    y[0, 0, 0, :4] = torch.tensor([0., 256., -256., 3.])
    with torch.no_grad():
        pp = ParametricPdf('laplace')
        p_mu = pp(y, [{'mu': mu, 'sigma': sigma}])
        p_zero = pp(y, [{'mu': mu, 'sigma': sigma}], zero_mu=True)
        pe = BallePdfEstim(c, 'balle', verbose=False)
        for p in pe.parameters():
            p.mul_(1.5)
        xz = torch.round(torch.randn(1, c, h, w, generator=g2) * 6.0).clamp(-256, 256)
        xz[0, 1, 0, :3] = torch.tensor([256., -256., 0.])
        p_z = pe(xz)
        cdf_z = ArithmeticCoder({'balle_pdf_estim_z': pe, 'device': 'cpu'}).pre_computed_z_cdf.detach().reshape(c, 514)
        ec = EntropyCoder()
        rate_y, rate_z = ec(p_zero, y), ec(p_z, xz)
        # the figure flag_debug prints (src/real_life/bitstream.py:307-318), before the division by 8000
        b = sigma / torch.sqrt(torch.tensor([2.0]))
        lap = torch.distributions.Laplace(torch.zeros_like(b), b)
        dbg_y = -torch.log2(torch.clamp(lap.cdf(y + 0.5) - lap.cdf(y - 0.5), 2 ** -16, 1.)).sum()
        dbg_z = -torch.log2(torch.clamp(pe(xz), 2 ** -16, 1.)).sum()
    save('rate_est_0', y=y, mu=mu, sigma=sigma, p_mu=p_mu, p_zero=p_zero, xz=xz, p_z=p_z, cdf_z=cdf_z, rate_y=rate_y,
         rate_z=rate_z, dbg_bits_y=dbg_y, dbg_bits_z=dbg_z, **sd_np(pe))
    print('done')


if __name__ == '__main__':
    main()
