"""State-dict schemas of the reference's recurrent networks + a deterministic synthetic
weight generator.

The reference ships FireNet / FireNet+ checkpoints only (pretrained/*/model.pth); the E2VID,
E2VID+, SSL-E2VID and HyperE2VID blobs are absent (.MISSING_LARGE_BLOBS).  Parity and
benchmarks for those layouts therefore run on weights drawn here.  Parameter names and
shapes are exactly the ones torch's state_dict() of the reference classes yields
(model/unet.py:40-82, model/submodules.py:8-313, model/legacy.py:32-111,
model/model.py:147-190), so a real checkpoint drops in unchanged.
"""
import hashlib
import zlib
from collections import OrderedDict

import numpy as np

E2VID_KWARGS = dict(num_bins=5, base_num_channels=32, num_encoders=3, num_residual_blocks=2,
                    kernel_size=5, norm='BN', use_upsample_conv=False,
                    recurrent_block_type='convlstm', skip_type='sum', final_activation='sigmoid')
# eval.py:135-137 (SSL-E2VID) -- the layout E2VID+ is believed to share
E2VID_PLUS_KWARGS = dict(num_bins=5, base_num_channels=32, num_encoders=3, num_residual_blocks=2,
                         kernel_size=5, norm=None, use_upsample_conv=True,
                         recurrent_block_type='convlstm', skip_type='sum', final_activation='none')


def _conv(shapes, name, cout, cin, k, bias=True):
    shapes[name + '.weight'] = (cout, cin, k, k)
    if bias:
        shapes[name + '.bias'] = (cout,)


def _bn(shapes, name, c):
    shapes[name + '.weight'] = (c,)
    shapes[name + '.bias'] = (c,)
    shapes[name + '.running_mean'] = (c,)
    shapes[name + '.running_var'] = (c,)
    shapes[name + '.num_batches_tracked'] = ()


def _in_tracked(shapes, name, c):
    """InstanceNorm2d(track_running_stats=True): running statistics only (no affine parameters)."""
    shapes[name + '.running_mean'] = (c,)
    shapes[name + '.running_var'] = (c,)
    shapes[name + '.num_batches_tracked'] = ()


def _gru(shapes, name, c):
    for g in ('reset_gate', 'update_gate', 'out_gate'):
        _conv(shapes, f'{name}.{g}', c, 2 * c, 3)


def unet_recurrent_schema(num_bins=5, base_num_channels=32, num_encoders=3, num_residual_blocks=2,
                          kernel_size=5, norm=None, use_upsample_conv=False,
                          recurrent_block_type='convlstm', prefix='unetrecurrent.', use_dynamic_decoder=False, **_):
    """Ordered {name: shape} of E2VIDRecurrent.state_dict() (registration order of
    UNetRecurrent.__init__, model/unet.py:99-106)."""
    s = OrderedDict()
    bn = norm == 'BN'
    inn = norm == 'IN'          # ConvLayers: InstanceNorm2d(track_running_stats=True); ResidualBlocks: plain InstanceNorm2d (no state)
    norm_of = lambda sh, name, c: _bn(sh, name, c) if bn else (_in_tracked(sh, name, c) if inn else None)
    k = kernel_size
    cin = [base_num_channels * 2 ** i for i in range(num_encoders)]
    cout = [base_num_channels * 2 ** (i + 1) for i in range(num_encoders)]
    _conv(s, prefix + 'head.conv2d', base_num_channels, num_bins, k)
    for i in range(num_encoders):
        p = f'{prefix}encoders.{i}'
        _conv(s, p + '.conv.conv2d', cout[i], cin[i], k, bias=not bn)
        norm_of(s, p + '.conv.norm_layer', cout[i])
        if recurrent_block_type == 'convlstm':
            _conv(s, p + '.recurrent_block.Gates', 4 * cout[i], 2 * cout[i], 3)
        else:
            _gru(s, p + '.recurrent_block', cout[i])
    cm = cout[-1]
    for i in range(num_residual_blocks):
        p = f'{prefix}resblocks.{i}'
        _conv(s, p + '.conv1', cm, cm, 3, bias=not bn)
        if bn:
            _bn(s, p + '.bn1', cm); _bn(s, p + '.bn2', cm)
        _conv(s, p + '.conv2', cm, cm, 3, bias=not bn)
    for i, (ci, co) in enumerate(zip(reversed(cout), reversed(cin))):
        p = f'{prefix}decoders.{i}'
        if i == 0 and use_dynamic_decoder:      # DynamicUpsampleLayer (model/submodules.py:100-127, model/hyper/)
            _conv(s, p + '.context_fusion.conv', 32, num_bins + 1, 3)
            s[p + '.dynamic_atom_generation.bases'] = (12, 25)          # Fourier-Bessel table: data, not random
            bn_ = p + '.dynamic_atom_generation.bases_net'
            _conv(s, bn_ + '.0', 64, 32, 3); _bn(s, bn_ + '.1', 64)
            _conv(s, bn_ + '.3', 72, 64, 3); _bn(s, bn_ + '.4', 72)
            s[p + '.dynamic_conv.compositional_coefficients'] = (co, ci * 6, 1, 1)
            s[p + '.dynamic_conv.bias'] = (co,)
            continue
        if use_upsample_conv:
            _conv(s, p + '.conv2d', co, ci, k, bias=not bn)
        else:
            s[p + '.transposed_conv2d.weight'] = (ci, co, k, k)
            if not bn:
                s[p + '.transposed_conv2d.bias'] = (co,)
        norm_of(s, p + '.norm_layer', co)
    _conv(s, prefix + 'pred.conv2d', 1, base_num_channels, 1, bias=not bn)
    norm_of(s, prefix + 'pred.norm_layer', 1)
    return s


def firenet_legacy_schema(num_bins=5, base_num_channels=16, kernel_size=3, prefix='net.', **_):
    """FireNet_legacy.state_dict() (model/legacy.py:32-77)."""
    s = OrderedDict(); c = base_num_channels
    _conv(s, prefix + 'head.conv.conv2d', c, num_bins, kernel_size)
    _gru(s, prefix + 'head.recurrent_block', c)
    _conv(s, prefix + 'resblocks.0.conv.conv1', c, c, 3); _conv(s, prefix + 'resblocks.0.conv.conv2', c, c, 3)
    _gru(s, prefix + 'resblocks.0.recurrent_block', c)
    _conv(s, prefix + 'resblocks.1.conv1', c, c, 3); _conv(s, prefix + 'resblocks.1.conv2', c, c, 3)
    _conv(s, prefix + 'pred.conv2d', 1, c, 1)
    return s


def spade_e2vid_schema(**_):
    """Unet6.state_dict() (model/spade_e2v.py:113-137), the 'SPADE-E2VID' method."""
    s = OrderedDict()
    _conv(s, 'fc', 32, 5, 5)

    def rec(name, cin, cout):
        _conv(s, name + '.conv0', cout, cin, 5, bias=False)
        _bn(s, name + '.bn', cout)
        _conv(s, name + '.recurrent_block.Gates', 4 * cout, 2 * cout, 3)

    def res(name):
        _conv(s, name + '.conv1', 256, 256, 3, bias=False); _conv(s, name + '.conv2', 256, 256, 3, bias=False)
        _bn(s, name + '.bn1', 256); _bn(s, name + '.bn2', 256)

    def up(name, cin, cout):
        _conv(s, name + '.conv0', 4 * cout, cin, 3, bias=False)
        p = name + '.norm.param_free_norm'               # BatchNorm2d(affine=False): running statistics only
        s[p + '.running_mean'] = (cout,); s[p + '.running_var'] = (cout,); s[p + '.num_batches_tracked'] = ()
        _conv(s, name + '.norm.mlp_shared.0', 64, 3, 3)
        _conv(s, name + '.norm.mlp_gamma', cout, 64, 3); _conv(s, name + '.norm.mlp_beta', cout, 64, 3)

    rec('rec0', 32, 64); rec('rec1', 64, 128); rec('rec2', 128, 256)
    res('res0'); res('res1')
    up('up0', 256, 128); up('up1', 128, 64)
    rec('up2', 64, 32)
    _conv(s, 'conv_img', 3, 32, 1)
    _bn(s, 'bn_img', 3)
    return s


def etnet_schema(num_bins=5, norm=None, **_):
    """EITR.state_dict() (model/eitr/u_trans.py:17-58), the 'ET-Net' method."""
    s = OrderedDict()
    bn = norm == 'BN'
    norm_of = lambda name, c: _bn(s, name, c) if bn else (_in_tracked(s, name, c) if norm == 'IN' else None)
    _conv(s, 'head.conv2d', 32, num_bins, 5, bias=not bn); norm_of('head.norm_layer', 32)
    for i, (ci, co) in enumerate([(32, 64), (64, 128), (128, 256)]):
        _conv(s, f'DownsampleConv.{i}.conv.conv2d', co, ci, 5, bias=not bn); norm_of(f'DownsampleConv.{i}.conv.norm_layer', co)
        _conv(s, f'DownsampleConv.{i}.recurrent_block.Gates', 4 * co, 2 * co, 3)

    def mha(p):
        s[p + '.in_proj_weight'] = (768, 256); s[p + '.in_proj_bias'] = (768,)
        s[p + '.out_proj.weight'] = (256, 256); s[p + '.out_proj.bias'] = (256,)

    def ln(p):
        s[p + '.weight'] = (256,); s[p + '.bias'] = (256,)

    def lin(p, o, i):
        s[p + '.weight'] = (o, i); s[p + '.bias'] = (o,)

    def enc(p):
        for l in range(3):
            q = f'{p}.encoder.layers.{l}'
            mha(q + '.self_attn'); ln(q + '.norm1'); lin(q + '.linear1', 1024, 256); lin(q + '.linear2', 256, 1024); ln(q + '.norm2')

    def dec(p):
        for l in range(2):
            q = f'{p}.decoder.layers.{l}'
            mha(q + '.self_attn'); ln(q + '.norm1'); mha(q + '.cross_attn'); ln(q + '.norm21'); ln(q + '.norm22')
            lin(q + '.linear1', 1024, 256); lin(q + '.linear2', 256, 1024); ln(q + '.norm3')

    enc('trans_encoder0'); dec('trans_decoder0')
    _conv(s, 'split1', 256, 128, 2)
    enc('trans_encoder1'); dec('trans_decoder1')
    _conv(s, 'split2', 256, 64, 4)
    enc('trans_encoder2'); dec('trans_decoder2')
    for i, (ci, co) in enumerate([(256, 128), (128, 64), (64, 32)]):
        _conv(s, f'UpsampleConv.{i}.conv2d', co, ci, 5, bias=not bn); norm_of(f'UpsampleConv.{i}.norm_layer', co)
    _conv(s, 'pred.conv2d', 1, 32, 1, bias=not bn); norm_of('pred.norm_layer', 1)
    return s


def firenet_schema(num_bins=5, base_num_channels=16, kernel_size=3, **_):
    """FireNet.state_dict() (model/model.py:154-165), the 'FireNet+' method."""
    s = OrderedDict(); c = base_num_channels
    _conv(s, 'head.conv2d', c, num_bins, kernel_size)
    _gru(s, 'G1', c)
    _conv(s, 'R1.conv1', c, c, 3); _conv(s, 'R1.conv2', c, c, 3)
    _gru(s, 'G2', c)
    _conv(s, 'R2.conv1', c, c, 3); _conv(s, 'R2.conv2', c, c, 3)
    _conv(s, 'pred.conv2d', 1, c, 1)
    return s


def lpips_alex_schema():
    """State dict of pyiqa's LPIPS(net='alex', version='0.1'): AlexNet feature convs + the five 1x1 'lin' heads."""
    s = OrderedDict()
    for name, co, ci, k in [("net.slice1.0", 64, 3, 11), ("net.slice2.3", 192, 64, 5), ("net.slice3.6", 384, 192, 3),
                            ("net.slice4.8", 256, 384, 3), ("net.slice5.10", 256, 256, 3)]:
        _conv(s, name, co, ci, k)
    for l, c in enumerate([64, 192, 384, 256, 256]):
        s[f'lin{l}.model.1.weight'] = (1, c, 1, 1)
    return s


def synth_lpips_state_dict(seed=0):
    """Deterministic stand-in for the (offline-unobtainable) LPIPS weights; the lin heads are non-negative."""
    sd = synth_state_dict(lpips_alex_schema(), seed=seed)
    for k in sd:
        if k.startswith('lin'):
            sd[k] = np.abs(sd[k]).astype(np.float32)
    return sd


def synth_state_dict(schema, seed=0, gain=1.0, fixed=None):
    """Deterministic fp32 numpy weights for a schema.  Each tensor is drawn from its own
    PCG64 stream keyed by (seed, crc32(name)) so adding/removing tensors never shifts the
    others.  Conv weights ~ U(-a, a), a = gain*sqrt(3/fan_in) (unit-variance preserving);
    biases small; BN gamma in [0.8,1.2], beta/mean in [-0.1,0.1], var in [0.5,1.5]."""
    out = OrderedDict()
    fixed = fixed or {}
    for name, shape in schema.items():
        if name in fixed:                 # e.g. the Fourier-Bessel bases buffer of HyperE2VID
            out[name] = np.asarray(fixed[name], dtype=np.float32).reshape(shape)
            continue
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            out[name] = np.array(0, dtype=np.int64)
        elif leaf == 'running_var':
            out[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == 'running_mean':
            out[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif len(shape) == 2:          # nn.Linear / MultiheadAttention projections: fan_in = in_features
            a = gain * np.sqrt(3.0 / shape[1])
            out[name] = rng.uniform(-a, a, shape).astype(np.float32)
        elif len(shape) == 4:
            # ConvTranspose2d weight is [Cin, Cout, k, k]; fan_in there is Cin*k*k/stride^2
            fan_in = shape[1] * shape[2] * shape[3]
            if 'transposed_conv2d' in name:
                fan_in = shape[0] * shape[2] * shape[3] / 4.0
            a = gain * np.sqrt(3.0 / fan_in)
            out[name] = rng.uniform(-a, a, shape).astype(np.float32)
        elif leaf == 'weight':            # BN gamma
            out[name] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        else:                             # conv bias / BN beta
            out[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
    return out


def rescale_encoder_conv(sd, enc=1, K=4096.0, prefix='unetrecurrent.'):
    """The SAME network function with one intermediate tensor K times larger: the output of encoder `enc`'s strided ConvLayer
    (BN layout: gamma and beta times K -- relu is positively homogeneous; no-norm layout: weight and bias times K) and the
    ConvLSTM gate weights that read it (the x half of cat(x, h), submodules.py:227-231) divided by K.  That tensor feeds nothing
    else (model/unet.py:120-123).  With K a power of two every fp32 product scales exactly, so the reference's outputs are
    bit-identical to the unscaled network's -- while the tensor leaves the +-4094 range of the H2 activation format: the test
    vehicle for 'saturation must not change results' (tests/test_gpu_eval.py, tests/golden/eval_loop_e2vid.json)."""
    out = OrderedDict((k, np.array(v, copy=True)) for k, v in sd.items())
    p = f'{prefix}encoders.{enc}'
    if p + '.conv.norm_layer.weight' in out:
        out[p + '.conv.norm_layer.weight'] *= np.float32(K)
        out[p + '.conv.norm_layer.bias'] *= np.float32(K)
    else:
        out[p + '.conv.conv2d.weight'] *= np.float32(K)
        out[p + '.conv.conv2d.bias'] *= np.float32(K)
    g = out[p + '.recurrent_block.Gates.weight']
    C = g.shape[1] // 2
    g[:, :C] /= np.float32(K)
    return out


def state_dict_digest(sd):
    """sha256 over names, shapes and raw bytes (order-independent of dict order)."""
    h = hashlib.sha256()
    for name in sorted(sd):
        a = np.ascontiguousarray(np.asarray(sd[name]))
        h.update(name.encode()); h.update(str(a.shape).encode()); h.update(str(a.dtype).encode())
        h.update(a.tobytes())
    return h.hexdigest()
