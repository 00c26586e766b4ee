"""CPU-side checks: the C ABI library exports every symbol the header declares, module aliases make
full-module pickles loadable, CLI mapping, product path fails loudly without a GPU."""
import ctypes
import io
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'aivc_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'static inline[^{]*\{.*?\n\}', '', src, flags=re.S)  # (inline rules shared with the oracle: not exports)
    return sorted(set(re.findall(r'\b(?:int|const char \*)\s*(aivc_\w+)\s*\(', src)))


def test_hip_library_exports_every_declared_symbol():
    lib_path = os.path.join(ROOT, 'aivc_amd', 'lib', 'libaivc_hip.so')
    if not os.path.exists(lib_path):
        import __graft_entry__ as g
        g.build_hip()
    lib = ctypes.CDLL(lib_path)  # loads without a GPU (no compute call is made)
    names = header_functions()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), 'libaivc_hip.so does not export %s' % n
    lib.aivc_abi_version.restype = ctypes.c_int
    from aivc_amd import abi
    assert lib.aivc_abi_version() == abi.ABI_VERSION
    # every prototype bound by abi.py is declared in the header and vice versa
    assert set(abi.PROTOTYPES) | {'aivc_abi_version', 'aivc_last_error', 'aivc_conv2d_variant', 'aivc_selfcheck_gdn_math'} == set(names)


def test_oracle_exports_ref_twins(oracle):
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libaivc_oracle.so'))
    from aivc_amd import abi
    for n in abi.PROTOTYPES:
        assert hasattr(lib, n + '_ref')


def test_argument_validation_without_gpu():
    """error paths return codes instead of crashing (no kernel is launched)."""
    from aivc_amd import _lib, abi
    fns = _lib.load()
    assert fns['aivc_conv2d'](None, None) == -1
    p = abi.ConvParams(abi.MODE_CONV, 7, 1, 3, 1, 8, 8, 4, 8, 8, 4, 0, 0, 0, 0, 0, 1, 1, None, None, None, 1, None, None)
    assert fns['aivc_conv2d'](ctypes.byref(p), None) == -2  # ksize 7 unsupported
    p = abi.ConvParams(abi.MODE_CONV, 3, 1, 1, 1, 8, 8, 3, 8, 8, 4, 0, 0, 0, 0, 0, 1, 1, None, None, None, 1, None, None)
    assert fns['aivc_conv2d'](ctypes.byref(p), None) == -1  # c_in not a multiple of 4
    p = abi.ConvParams(abi.MODE_CONV, 3, 2, 1, 1, 9, 9, 4, 4, 5, 4, 0, 0, 0, 0, 0, 1, 1, None, None, None, 1, None, None)
    assert fns['aivc_conv2d'](ctypes.byref(p), None) == -1  # wrong output size
    assert fns['aivc_conv2d_variant'](ctypes.byref(abi.ConvParams(
        abi.MODE_CONV, 3, 1, 1, 1, 270, 480, 128, 270, 480, 128, 0, 0, 0, 0, 0, 1, 1, None, None, None, 1, None, None))) == 105  # 64x128 (round-3 tile rules)


def test_product_path_refuses_cpu_tensors():
    from aivc_amd import ops
    from aivc_amd._lib import AivcNativeError
    with pytest.raises(AivcNativeError):
        ops.conv2d(torch.zeros(1, 4, 4, 4), torch.zeros(4, 3, 3, 4))
    from aivc_amd.layers.misc.custom_conv_layers import CustomConvLayer
    with pytest.raises(AivcNativeError):
        CustomConvLayer(3, 4, 4)(torch.zeros(1, 4, 8, 8))
    from aivc_amd.cli_common import resolve_device
    with pytest.raises(SystemExit):
        resolve_device(True)


def test_full_module_pickle_round_trip_through_reference_module_names():
    """A reference .pt is a pickle of the whole nn.Module whose classes live in `layers.*`,
    `models.*`: the aliases must resolve them to this package."""
    import aivc_amd
    from aivc_amd import synth
    from aivc_amd.models import arch
    aivc_amd.install_aliases()
    model = synth.make_model(arch.TINY_WIDTHS, seed=3)
    for net in (model.codec_net.codec_net, model.mode_net.mode_net):
        net.ac = None  # not pickled upstream either

    # write the pickle with the REFERENCE's module paths (what a real .pt contains)
    classes = {type(m) for m in model.modules() if type(m).__module__.startswith('aivc_amd.')}
    saved = {c: c.__module__ for c in classes}
    try:
        for c in classes:
            c.__module__ = c.__module__[len('aivc_amd.'):]
        buf = io.BytesIO()
        torch.save(model, buf)
    finally:
        for c, m in saved.items():
            c.__module__ = m
    raw = buf.getvalue()
    assert b'layers.misc.custom_conv_layers' in raw and b'aivc_amd.layers' not in raw
    loaded = torch.load(io.BytesIO(raw), map_location='cpu', weights_only=False)
    assert type(loaded).__name__ == 'FullNet'
    sd_a, sd_b = model.state_dict(), loaded.state_dict()
    assert sd_a.keys() == sd_b.keys()
    assert all(torch.equal(sd_a[k], sd_b[k]) for k in sd_a)
    # the attribute contract the reference's decoder reads (src/real_life/decode.py:447-453,770-795)
    cn = loaded.codec_net.codec_net
    for attr in ('g_s', 'h_s', 'g_a_ref', 'pdf_y', 'pdf_z', 'pdf_parameterizer', 'out_c_shortcut_y', 'nb_ft_y',
                 'nb_ft_z', 'gain_I', 'flag_gain_p_b', 'gain_P', 'gain_B'):
        assert hasattr(cn, attr)
    for attr in ('codec_net', 'mode_net', 'motion_compensation', 'in_layer', 'out_layer', 'model_param'):
        assert hasattr(loaded, attr)
    assert 'lambda_tradeoff' in loaded.model_param


def test_cli_gop_mapping():
    from aivc_amd.aivc import gop_name
    assert gop_name('AI', 32, 32) == '1_GOP_0'
    assert gop_name('LDP', 32, 8) == 'LDP_8'
    assert gop_name('RA', 16, 32) == '2_GOP_16'   # the reference's sanity_script.sh setting
    assert gop_name('RA', 32, 32) == '1_GOP_32'
    with pytest.raises(SystemExit):
        gop_name('RA', 16, 24)


def test_section_framing_without_gpu():
    from aivc_amd.real_life.bitstream import split_sections
    frame = (0).to_bytes(4, 'big') * 2 + (3).to_bytes(4, 'big') + b'abc' + (1).to_bytes(4, 'big') + b'\x00'
    assert split_sections(frame) == [b'', b'', b'abc', b'\x00']


def test_md5_debug_digest_is_the_reference_procedure(tmp_path):
    """flag_md5sum (src/real_life/bitstream.py:229-234): md5 of the np.savetxt text of the NCHW-flattened latent"""
    import hashlib

    import numpy as np
    import torch
    from aivc_amd.real_life.bitstream import latent_md5
    rng = np.random.default_rng(3)
    q = torch.from_numpy(rng.integers(-256, 256, (1, 5, 7, 3)).astype(np.int16))  # NHWC
    q[0, 0, 0, 0], q[0, 0, 0, 1] = -256, 255
    x_nchw = q.permute(0, 3, 1, 2).to(torch.int16).numpy().astype(int)
    path = tmp_path / 'tmp_tensor.npy'
    np.savetxt(path, x_nchw.flatten())
    want = hashlib.md5(open(path, 'rb').read()).hexdigest().encode()
    assert latent_md5(q) == want and len(want) == 32


def test_debug_digest_kinds(tmp_path, capsys):
    """flag_bitstream_debug: a bare digest is the md5 of the PNG file (the reference's), 'raw:<md5>' that of the 8-bit
    plane; the decoder compares like with like (src/real_life/decode.py:304-326)"""
    import hashlib
    import numpy as np
    import torch
    from aivc_amd.real_life import decode as dec
    rng = np.random.default_rng(1)
    frames = [{k: torch.from_numpy(rng.integers(0, 256, (1, 6, 8) if k == 'y' else (1, 3, 4), dtype=np.uint8)) for k in 'yuv'}]
    d = str(tmp_path / 'dbg')
    dec.write_debug_md5(frames, 5, d)
    assert dec.check_debug_md5(frames, 5, d) == 0
    raw = hashlib.md5(frames[0]['y'].numpy().tobytes()).hexdigest()
    assert dec.plane_md5(frames[0]['y'], 'raw') == raw
    with open(tmp_path / 'dbg' / '5_y.md5', 'w') as f:  # an encoder without Pillow wrote this one
        f.write('raw:' + raw)
    assert dec.check_debug_md5(frames, 5, d) == 0
    with open(tmp_path / 'dbg' / '5_u.md5', 'w') as f:
        f.write('raw:' + raw)  # wrong plane
    assert dec.check_debug_md5(frames, 5, d) == 1
    assert 'Incorrect reconstruction' in capsys.readouterr().out


def test_clip_shard_is_built_once():
    """clip_shard() keeps one ClipShard per (units, world, device): constructing one allocates communicators"""
    import torch
    from aivc_amd import parallel
    a = parallel.clip_shard(4, torch.device('cpu'))
    assert parallel.clip_shard(4, torch.device('cpu')) is a and parallel.clip_shard(3, torch.device('cpu')) is not a
    assert (a.G, a.R, a.units) == (1, 1, [0, 1, 2, 3])


def test_bench_cli_parses():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--help'], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and '--gpus' in (out.stdout + out.stderr) and '--active-y' in (out.stdout + out.stderr)


def test_package_import_asks_for_eight_hardware_queues():
    """the codec's 8 entropy side streams need as many hardware queues: the package sets the runtime's
    variable at import unless the user already did -- checked in a fresh interpreter"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = 'import os, sys; sys.path.insert(0, %r); import aivc_amd; print(os.environ["GPU_MAX_HW_QUEUES"])' % root
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120, env=env)
    assert out.stdout.strip() == '8', out.stderr
    env['GPU_MAX_HW_QUEUES'] = '4'
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120, env=env)
    assert out.stdout.strip() == '4', out.stderr


def test_encode_batch_hands_sections_over_before_the_synthesis():
    """FrameCodec.encode_batch(on_sections=...) exists and encode_units uses it (the flags of a batch leave for the
    host before its CodecNet synthesis is queued): a source-level check, the behaviour is covered by the GPU tests"""
    import inspect
    from aivc_amd import codec
    src = inspect.getsource(codec.FrameCodec.encode_batch)
    assert src.index('on_sections(sections)') < src.index('self.cod.synthesise(')
    assert 'on_sections=' in inspect.getsource(codec.FrameCodec.encode_units)


def _save_under_foreign_module_names(model, rename):
    """torch.save(model) as a pickle whose classes live at other dotted paths: rename(cls) -> module name"""
    import types
    classes = {type(m) for m in model.modules() if type(m).__module__.startswith('aivc_amd.')}
    saved = {c: c.__module__ for c in classes}
    fake = {}
    try:
        for c in classes:
            new = rename(c)
            if new not in sys.modules:
                fake[new] = sys.modules[new] = types.ModuleType(new)
            if new in fake:
                setattr(fake[new], c.__name__, c)
            c.__module__ = new
        buf = io.BytesIO()
        torch.save(model, buf)
    finally:
        for c, m in saved.items():
            c.__module__ = m
        for k in fake:
            del sys.modules[k]
    return buf.getvalue()


def test_load_model_resolves_unknown_module_paths_by_class_name(tmp_path, monkeypatch):
    """The dotted paths of the pickled `models.*` classes are unknown (the package is missing from the reference
    snapshot): a file that puts FullNet at models.net and the conditional coders at models.cond.sub
    loads through load_model's class-name fallback (src/model_mngt/model_management.py:341-361 is a plain torch.load),
    the inspector lists what it asks for without unpickling, and an unknown class is reported by name up front."""
    import aivc_amd
    from aivc_amd import synth
    from aivc_amd.model_mngt import model_management, pickle_compat
    from aivc_amd.models import arch
    aivc_amd.install_aliases()
    model = synth.make_model(arch.TINY_WIDTHS, seed=5)
    for net in (model.codec_net.codec_net, model.mode_net.mode_net):
        net.ac = None

    def rename(c):
        ref = c.__module__[len('aivc_amd.'):]
        if not ref.startswith('models.'):
            return ref  # layer classes: the reference's own paths
        return 'models.net' if c.__name__ == 'FullNet' else 'models.cond.sub'
    raw = _save_under_foreign_module_names(model, rename)
    names = pickle_compat.pickle_globals(raw)
    assert ('models.net', 'FullNet') in names and ('models.cond.sub', 'ConditionalNet') in names
    assert ('layers.misc.custom_conv_layers', 'CustomConvLayer') in names
    assert pickle_compat.unresolved(raw) == []
    (tmp_path / '0_model.pt').write_bytes(raw)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(model_management, 'attach_arithmetic_coders', lambda m, d=None: m)  # (the coder's tables are GPU work)
    loaded = model_management.load_model(prefix='0_', on_cpu=True)
    assert type(loaded).__module__ == 'aivc_amd.models.full_net'
    sd_a, sd_b = model.state_dict(), loaded.state_dict()
    assert sd_a.keys() == sd_b.keys() and all(torch.equal(sd_a[k], sd_b[k]) for k in sd_a)
    # the command-line inspector over the same file
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'inspect_pickle.py'), '0_model.pt'],
                         capture_output=True, text=True)
    assert out.returncode == 0 and 'models.net.FullNet' in out.stdout and 'by NAME -> aivc_amd.models.full_net.FullNet' in out.stdout
    # a class this build does not define: named in the error, before anything is unpickled
    foo = type('FooBNet', (type(model),), {'__module__': 'aivc_amd.models.full_net'})
    model.__class__ = foo
    raw2 = _save_under_foreign_module_names(model, lambda c: 'models.net' if c is foo else rename(c))
    (tmp_path / '1_model.pt').write_bytes(raw2)
    with pytest.raises(ImportError, match='models.net.FooBNet'):
        model_management.load_model(prefix='1_', on_cpu=True)
    # protocol-4 qualified names of nested classes resolve by walking the attributes ...
    import collections
    assert pickle_compat.resolve('collections', 'OrderedDict.fromkeys') == collections.OrderedDict.fromkeys
    # ... and the by-name fallback only applies under the reference's own packages: a third-party module that happens
    # to name a class like one of ours stays unresolved instead of binding to the local class
    assert pickle_compat.resolve('models.anything', 'FullNet') is type(loaded)
    assert pickle_compat.resolve('src.models.anything', 'FullNet') is type(loaded)
    with pytest.raises(AttributeError):
        pickle_compat.resolve('somebody_elses_pkg.nets', 'FullNet')


def test_distributed_watchdog_names_the_wait(capfd, monkeypatch):
    """a host-side wait on communication that lasts longer than AIVC_DIST_WARN_S is reported once, with the wait's name
    and the collectives issued before it (aivc_amd/parallel.py: under RCCL a missing peer only shows at the next host
    synchronisation, far from its cause)"""
    import time
    from aivc_amd import parallel
    monkeypatch.setenv('AIVC_DIST_WARN_S', '0.2')
    monkeypatch.setitem(parallel._WATCH, 'thread', None)
    parallel._start_watchdog()
    parallel._note('all_gather level reconstructions')
    with parallel._host_wait('level reconstructions'):
        time.sleep(0.8)
    err = capfd.readouterr().err
    assert err.count('[aivc_amd.parallel]') == 1 and '"level reconstructions"' in err and 'all_gather level reconstructions' in err
    with parallel._host_wait('quick'):
        pass
    time.sleep(0.3)
    assert '[aivc_amd.parallel]' not in capfd.readouterr().err


def test_unit_lengths_of_a_container_with_mixed_coding_structures():
    """decode_video_sharded places the gathered frames by every unit's OWN length (a container may mix GOP structures:
    decode_video / decode_units accept that) -- parallel.unit_lengths reads them from the GOP records"""
    from aivc_amd import parallel
    from aivc_amd.real_life import cat_binary_files as container
    from aivc_amd.real_life import header as hdr
    frame = (0).to_bytes(4, 'big') * 2 + (1).to_bytes(4, 'big') + b'z' + (1).to_bytes(4, 'big') + b'\x00'
    gops = [container.pack_gop(hdr.gop_header_bytes(name, 0.), [frame] * n) for name, n in (('1_GOP_2', 3), ('1_GOP_4', 5), ('LDP_2', 3))]
    dd = {'x': (48, 80), 'y': (3, 5), 'z': (1, 2)}
    blob = container.pack_video(hdr.video_header_bytes(dd, len(gops), 0, 10), gops)
    assert parallel.unit_lengths(blob) == [3, 5, 3]


def test_truncated_container_is_an_error_not_garbage():
    """every length prefix of the container is checked against the data at hand: a short file raises ContainerError naming
    the record (the reference's fixed-size reads would carry on with what they got, src/real_life/decode.py:329-426)"""
    from aivc_amd.real_life import cat_binary_files as container
    from aivc_amd.real_life import header as hdr
    from aivc_amd.real_life.bitstream import split_sections
    frame = (0).to_bytes(4, 'big') * 2 + (3).to_bytes(4, 'big') + b'abc' + (1).to_bytes(4, 'big') + b'\x00'
    gop = container.pack_gop(hdr.gop_header_bytes('1_GOP_2', 0.), [frame] * 3)
    blob = container.pack_video(hdr.video_header_bytes({'x': (48, 80), 'y': (3, 5), 'z': (1, 2)}, 2, 0, 5), [gop, gop])
    assert len(container.unpack_video(blob)[3]) == 2 and len(container.unpack_gop(gop)[2]) == 3
    for cut in (5, 17, 20, len(blob) - 1, len(blob) - len(gop) + 3):
        with pytest.raises(container.ContainerError):
            container.unpack_video(blob[:cut])
    with pytest.raises(container.ContainerError, match='frame 2 of 3'):
        container.unpack_gop(gop[:-4])
    with pytest.raises(container.ContainerError, match='codecnet_y'):
        split_sections(frame[:-1])
    assert split_sections(frame) == [b'', b'', b'abc', b'\x00']
    # the path API reads a section back while the later ones are not written yet (src/real_life/bitstream.py:333-350)
    assert split_sections(frame[:-5], upto=3) == [b'', b'', b'abc']


def test_rows_of_is_a_view_for_consecutive_rows():
    """codec._rows_of: the decoded latents of a synthesis batch are a view of the entropy stage's batch tensor when its
    frames sit in consecutive rows, a copy otherwise -- the same values either way"""
    import torch
    from aivc_amd.codec import _rows_of
    t = torch.arange(5 * 2 * 3 * 4, dtype=torch.float32).reshape(5, 2, 3, 4)
    u = torch.arange(100, 100 + 2 * 2 * 3 * 4, dtype=torch.float32).reshape(2, 2, 3, 4)
    v = _rows_of([(t, 1), (t, 2), (t, 3)])
    assert v.data_ptr() == t[1:].data_ptr() and v.shape == (3, 2, 3, 4) and torch.equal(v, t[1:4])
    c = _rows_of([(t, 3), (t, 1)])
    assert torch.equal(c, torch.stack([t[3], t[1]])) and c.data_ptr() != t[3:].data_ptr()
    m = _rows_of([(t, 4), (u, 0)])
    assert torch.equal(m, torch.stack([t[4], u[0]]))
    assert _rows_of([(u, 1)]).data_ptr() == u[1:].data_ptr()


def _fake_codec(c_y=8, md5=False):
    """what FrameCodec.check_sections / _entropy_bytes read of a codec: two networks with an ArithmeticCoder each"""
    from types import SimpleNamespace
    from aivc_amd.real_life.bitstream import ArithmeticCoder
    ac = SimpleNamespace(flag_md5sum=md5, _parse_maps=ArithmeticCoder._parse_maps)
    net = SimpleNamespace(ac=ac, nb_ft_y=c_y)
    return SimpleNamespace(mof=net, cod=SimpleNamespace(ac=ac, nb_ft_y=c_y))


def _frame(mof_y, cod_y):
    return b''.join(len(s).to_bytes(4, 'big') + s for s in (b'z', mof_y, b'z', cod_y))


def test_map_lists_of_every_frame_are_checked_before_any_rank_starts():
    """FrameCodec.check_sections: a y section whose map list is malformed is a ContainerError for the WHOLE container on
    the host (every rank holds the bitstream and fails alike) -- not only on the rank that decodes the frame"""
    from aivc_amd.codec import FrameCodec
    from aivc_amd.real_life import cat_binary_files as container
    good = _frame(b'\x02\x00\x03ab', b'\x01\x07c')
    parsed = [('1_GOP_2', 0., [_frame(b'', b'\x00'), good, good])]
    FrameCodec.check_sections(_fake_codec(), parsed)
    for bad in (_frame(b'\x02\x00', b'\x00'),          # two maps announced, one listed
                _frame(b'\x01\x09x', b'\x00'),         # map 9 of 8
                _frame(b'\x00', b''),                  # no map count at all
                _frame(b'\x00', b'\x09' + bytes(9))):  # more maps than the latent has
        with pytest.raises(container.ContainerError):
            FrameCodec.check_sections(_fake_codec(), [('1_GOP_2', 0., [_frame(b'', b'\x00'), good, bad])])
    # the I frame's MOFNet section is not read (nothing decodes it)
    FrameCodec.check_sections(_fake_codec(), [('1_GOP_2', 0., [_frame(b'\x63', b'\x00'), good, good])])
    # under flag_md5sum the list sits behind the 32 characters of the digest
    md5 = b'0' * 32
    FrameCodec.check_sections(_fake_codec(md5=True), [('1_GOP_2', 0., [_frame(b'', md5 + b'\x00')] + [_frame(md5 + b'\x01\x02q', md5 + b'\x00')] * 2)])


def test_entropy_memory_estimate_reads_the_map_count_behind_the_md5_text():
    from aivc_amd.codec import FrameCodec
    dd = {'x': (64, 64), 'y': (4, 4), 'z': (1, 1)}
    plain = [('1_GOP_0', 0., [_frame(b'', b'\x03\x00\x01\x02xyz')])]
    md5 = b'f' * 32  # ('f' = 102 as a map count)
    tagged = [('1_GOP_0', 0., [_frame(b'', md5 + b'\x03\x00\x01\x02xyz')])]
    a = FrameCodec._entropy_bytes(_fake_codec(), plain, [0], dd)
    b = FrameCodec._entropy_bytes(_fake_codec(md5=True), tagged, [0], dd)
    assert a == b == 3 * 16 * 134 + 16 * 4 * 640
