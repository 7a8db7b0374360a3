"""Oracle: recurrent-network forward passes on torch-CPU fp32.  TEST INFRASTRUCTURE ONLY.

Functional restatement (state_dict in, tensors out) of the reference's nn.Modules, NCHW:
  UNetRecurrentOracle   model/unet.py:9-143 (BaseUNet/UNetRecurrent) +
                        model/model.py:108-144 (E2VIDRecurrent: prefix 'unetrecurrent.')
  conv_layer            model/submodules.py:8-35   (conv -> BN(eval) -> activation)
  conv_lstm             model/submodules.py:187-245 (gate order in, remember, out, cell)
  conv_gru              model/submodules.py:248-287
  residual_block        model/submodules.py:152-184
  transposed_conv_layer model/submodules.py:38-66  (stride 2, output_padding 1)
  upsample_conv_layer   model/submodules.py:69-97  (bilinear x2, align_corners=False)
  FireNetLegacyOracle   model/legacy.py:32-111,155-187 (prefix 'net.')
  FireNetOracle         model/model.py:147-190      (FireNet+)
BatchNorm is applied un-folded in eval mode (running stats), exactly as the reference does
(eval.py:112), so this oracle also checks the product's BN folding.
"""
import torch
import torch.nn.functional as F


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'], False, 0.1, 1e-5)


def _norm(sd, prefix, x, norm):
    """norm layer of ConvLayer / TransposedConvLayer / UpsampleConvLayer (submodules.py:20-23): BatchNorm2d, or
    InstanceNorm2d(track_running_stats=True) which in eval mode normalises with its RUNNING statistics (no affine)."""
    if norm == 'BN':
        return _bn(sd, prefix, x)
    if norm == 'IN':
        return F.instance_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], None, None, False, 0.1, 1e-5)
    return x


def _act(x, activation):
    if activation is None:
        return x
    return getattr(torch, activation)(x)


def conv_layer(sd, p, x, stride=1, padding=0, activation='relu', norm=None):
    x = F.conv2d(x, sd[p + '.conv2d.weight'], sd.get(p + '.conv2d.bias'), stride, padding)
    x = _norm(sd, p + '.norm_layer', x, norm)
    return _act(x, activation)


def transposed_conv_layer(sd, p, x, padding, activation='relu', norm=None):
    x = F.conv_transpose2d(x, sd[p + '.transposed_conv2d.weight'], sd.get(p + '.transposed_conv2d.bias'),
                           stride=2, padding=padding, output_padding=1)
    x = _norm(sd, p + '.norm_layer', x, norm)
    return _act(x, activation)


def upsample_conv_layer(sd, p, x, padding, activation='relu', norm=None):
    x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    x = F.conv2d(x, sd[p + '.conv2d.weight'], sd.get(p + '.conv2d.bias'), 1, padding)
    x = _norm(sd, p + '.norm_layer', x, norm)
    return _act(x, activation)


def conv_lstm(sd, p, x, state):
    C = x.shape[1]
    if state is None:
        z = torch.zeros_like(x)
        state = (z, z.clone())
    h, c = state
    g = F.conv2d(torch.cat((x, h), 1), sd[p + '.Gates.weight'], sd[p + '.Gates.bias'], padding=1)
    i, r, o, cell = g.chunk(4, 1)
    i, r, o, cell = torch.sigmoid(i), torch.sigmoid(r), torch.sigmoid(o), torch.tanh(cell)
    c2 = (r * c) + (i * cell)
    h2 = o * torch.tanh(c2)
    return h2, c2


def conv_gru(sd, p, x, h):
    if h is None:
        h = torch.zeros_like(x)
    s = torch.cat([x, h], 1)
    update = torch.sigmoid(F.conv2d(s, sd[p + '.update_gate.weight'], sd[p + '.update_gate.bias'], padding=1))
    reset = torch.sigmoid(F.conv2d(s, sd[p + '.reset_gate.weight'], sd[p + '.reset_gate.bias'], padding=1))
    out = torch.tanh(F.conv2d(torch.cat([x, h * reset], 1), sd[p + '.out_gate.weight'],
                              sd[p + '.out_gate.bias'], padding=1))
    return h * (1 - update) + out * update


def dynamic_upsample_layer(sd, p, x, ev_tensor, prev_recs):
    """DynamicUpsampleLayer (model/submodules.py:100-127) over model/hyper/hyper_dynamic.py:7-92:
    bilinear x2; context = conv3(bilinear x1/4 of cat(events, prev_rec)); bases_net (conv-BN-tanh x2) ->
    per-pixel coefficients [6,12] x Fourier-Bessel bases [12,25] -> 6 atoms of 5x5; every input channel is
    filtered by the 6 atoms (unfold + einsum) and a 1x1 conv mixes the C*6 maps; relu."""
    x_up = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    ctx = torch.cat((ev_tensor, prev_recs), dim=1)
    ctx = F.interpolate(ctx, scale_factor=0.25, mode='bilinear', align_corners=False)
    ctx = F.conv2d(ctx, sd[p + '.context_fusion.conv.weight'], sd[p + '.context_fusion.conv.bias'], padding=1)
    bn = p + '.dynamic_atom_generation.bases_net'
    c = F.conv2d(ctx, sd[bn + '.0.weight'], sd[bn + '.0.bias'], padding=1)
    c = torch.tanh(_bn(sd, bn + '.1', c))
    c = F.conv2d(c, sd[bn + '.3.weight'], sd[bn + '.3.bias'], padding=1)
    c = torch.tanh(_bn(sd, bn + '.4', c))
    N, _, H, W = c.shape
    bases = sd[p + '.dynamic_atom_generation.bases']
    atoms = torch.einsum('bmkhw,kl->bmlhw', c.view(N, 6, 12, H, W), bases)
    C = x_up.shape[1]
    u = F.unfold(x_up, kernel_size=5, padding=2).view(N, C, 25, H, W)
    inter = torch.einsum('bmlhw,bclhw->bcmhw', atoms, u).reshape(N, C * 6, H, W)
    out = F.conv2d(inter, sd[p + '.dynamic_conv.compositional_coefficients'], sd[p + '.dynamic_conv.bias'])
    return torch.relu(out)


def residual_block(sd, p, x, norm=None):
    # norm='IN' here is a plain InstanceNorm2d (submodules.py:160-162): instance statistics even in eval mode
    out = F.conv2d(x, sd[p + '.conv1.weight'], sd.get(p + '.conv1.bias'), padding=1)
    if norm == 'BN':
        out = _bn(sd, p + '.bn1', out)
    elif norm == 'IN':
        out = F.instance_norm(out)
    out = torch.relu(out)
    out = F.conv2d(out, sd[p + '.conv2.weight'], sd.get(p + '.conv2.bias'), padding=1)
    if norm == 'BN':
        out = _bn(sd, p + '.bn2', out)
    elif norm == 'IN':
        out = F.instance_norm(out)
    return torch.relu(out + x)


class UNetRecurrentOracle:
    """E2VIDRecurrent (model/model.py:108-144) over UNetRecurrent (model/unet.py:85-143)."""

    def __init__(self, sd, num_bins=5, base_num_channels=32, num_encoders=3, num_residual_blocks=2,
                 kernel_size=5, norm=None, use_upsample_conv=False, recurrent_block_type='convlstm',
                 final_activation='none', prefix='unetrecurrent.', use_dynamic_decoder=False):
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        self.pre, self.k, self.norm = prefix, kernel_size, norm
        self.num_encoders, self.num_res = num_encoders, num_residual_blocks
        self.up, self.rec = use_upsample_conv, recurrent_block_type
        self.final = getattr(torch, final_activation, None)
        self.dynamic = use_dynamic_decoder
        self.reset_states()

    def reset_states(self):
        self.states = [None] * self.num_encoders
        self.prev_recs = None                      # model/model.py:129-131

    def __call__(self, x, taps=None):
        sd, pre, k = self.sd, self.pre, self.k
        ev_tensor = x
        if self.prev_recs is None:                 # model/model.py:139-141
            self.prev_recs = torch.zeros(x.shape[0], 1, x.shape[2], x.shape[3])
        x = conv_layer(sd, pre + 'head', x, 1, k // 2, 'relu', None)   # head has norm=None (unet.py:77-82)
        head = x
        if taps is not None: taps['head'] = x
        blocks = []
        for i in range(self.num_encoders):
            p = f'{pre}encoders.{i}'
            x = conv_layer(sd, p + '.conv', x, 2, k // 2, 'relu', self.norm)
            if taps is not None: taps[f'enc{i}.conv'] = x
            if self.rec == 'convlstm':
                st = conv_lstm(sd, p + '.recurrent_block', x, self.states[i]); x = st[0]
            else:
                st = conv_gru(sd, p + '.recurrent_block', x, self.states[i]); x = st
            self.states[i] = st
            blocks.append(x)
        for i in range(self.num_res):
            x = residual_block(sd, f'{pre}resblocks.{i}', x, self.norm)
            if taps is not None: taps[f'res{i}'] = x
        for i in range(self.num_encoders):
            x = x + blocks[self.num_encoders - i - 1]
            p = f'{pre}decoders.{i}'
            if i == 0 and self.dynamic:
                x = dynamic_upsample_layer(sd, p, x, ev_tensor, self.prev_recs)
            elif self.up:
                x = upsample_conv_layer(sd, p, x, k // 2, 'relu', self.norm)
            else:
                x = transposed_conv_layer(sd, p, x, k // 2, 'relu', self.norm)
            if taps is not None: taps[f'dec{i}'] = x
        img = conv_layer(sd, pre + 'pred', x + head, 1, 0, None, self.norm)
        if self.final is not None:
            img = self.final(img)
        self.prev_recs = img.detach()              # model/model.py:143
        return img


class FireNetLegacyOracle:
    """FireNet_legacy / UNetFire (model/legacy.py:32-111,155-187): head conv3+ConvGRU,
    resblock0 + ConvGRU, resblock1 plain, pred 1x1, no final activation, norm 'none'."""

    def __init__(self, sd, prefix='net.'):
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        self.pre = prefix
        self.num_encoders = 4            # legacy.py:127-130 default -> cropper pads to /16
        self.reset_states()

    def reset_states(self):
        self.states = [None, None]

    def __call__(self, x):
        sd, pre = self.sd, self.pre
        x = conv_layer(sd, pre + 'head.conv', x, 1, 1, 'relu', None)
        x = conv_gru(sd, pre + 'head.recurrent_block', x, self.states[0]); self.states[0] = x
        x = residual_block(sd, pre + 'resblocks.0.conv', x)
        x = conv_gru(sd, pre + 'resblocks.0.recurrent_block', x, self.states[1]); self.states[1] = x
        x = residual_block(sd, pre + 'resblocks.1', x)
        return conv_layer(sd, pre + 'pred', x, 1, 0, None, None)


class FireNetOracle:
    """FireNet (model/model.py:147-190), the 'FireNet+' method; num_encoders forced 0 (eval.py:154-155)."""

    def __init__(self, sd):
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        self.num_encoders = 0
        self.reset_states()

    def reset_states(self):
        self.states = [None, None]

    def __call__(self, x):
        sd = self.sd
        x = conv_layer(sd, 'head', x, 1, 1, 'relu', None)
        x = conv_gru(sd, 'G1', x, self.states[0]); self.states[0] = x
        x = residual_block(sd, 'R1', x)
        x = conv_gru(sd, 'G2', x, self.states[1]); self.states[1] = x
        x = residual_block(sd, 'R2', x)
        return conv_layer(sd, 'pred', x, 1, 0, None, None)


class SpadeE2vidOracle:
    """Unet6 (model/spade_e2v.py:113-179), the 'SPADE-E2VID' method: fc head; three RecurrentConvLayers (conv k5 no
    bias -> BN -> relu -> ConvLSTM; strides 1, 2, 2); two BN residual blocks; two UpConvLayer3 (conv3 -> PixelShuffle(2)
    -> SPADE(BatchNorm2d(affine=False); segmap -> conv3+relu -> gamma, beta) -> relu) fed x + skip; a fourth
    RecurrentConvLayer; conv_img 1x1 -> bn_img -> sigmoid (3 channels = prev_recs); image = their mean.
    First frame: x[:, :3] is min/max-normalised IN PLACE (the head sees it) and used as the segmentation map."""

    def __init__(self, sd):
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        self.num_encoders = 3
        self.reset_states()

    def reset_states(self):
        self.states = None
        self.prev_recs = None

    def _rec(self, name, x, state, stride):
        sd = self.sd
        x = F.conv2d(x, sd[name + '.conv0.weight'], None, stride, 2)
        x = torch.relu(_bn(sd, name + '.bn', x))
        st = conv_lstm(sd, name + '.recurrent_block', x, state)
        return st[0], st

    def _res(self, name, x):
        sd = self.sd
        out = torch.relu(_bn(sd, name + '.bn1', F.conv2d(x, sd[name + '.conv1.weight'], None, padding=1)))
        out = _bn(sd, name + '.bn2', F.conv2d(out, sd[name + '.conv2.weight'], None, padding=1))
        return torch.relu(out + x)

    def _up(self, name, x, seg):
        sd = self.sd
        x = F.pixel_shuffle(F.conv2d(x, sd[name + '.conv0.weight'], None, padding=1), 2)
        p = name + '.norm'
        normalized = F.batch_norm(x, sd[p + '.param_free_norm.running_mean'], sd[p + '.param_free_norm.running_var'],
                                  None, None, False, 0.1, 1e-5)
        seg = F.interpolate(seg, size=x.shape[-2:], mode='nearest')
        actv = torch.relu(F.conv2d(seg, sd[p + '.mlp_shared.0.weight'], sd[p + '.mlp_shared.0.bias'], padding=1))
        gamma = F.conv2d(actv, sd[p + '.mlp_gamma.weight'], sd[p + '.mlp_gamma.bias'], padding=1)
        beta = F.conv2d(actv, sd[p + '.mlp_beta.weight'], sd[p + '.mlp_beta.bias'], padding=1)
        return torch.relu(normalized * (1 + gamma) + beta)

    def __call__(self, x):
        sd = self.sd
        x = x.clone()
        prev = [None] * 4 if self.states is None else self.states
        if self.prev_recs is None:
            x_org = x[:, :3]
            x_org -= x_org.min()
            if x_org.max() > 0:
                x_org /= x_org.max()
        else:
            x_org = self.prev_recs
        head = torch.relu(F.conv2d(x, sd['fc.weight'], sd['fc.bias'], padding=2))
        x0, s0 = self._rec('rec0', head, prev[0], 1)
        x1, s1 = self._rec('rec1', x0, prev[1], 2)
        x2, s2 = self._rec('rec2', x1, prev[2], 2)
        y = self._res('res1', self._res('res0', x2))
        y = self._up('up0', y + x2, x_org)
        y = self._up('up1', y + x1, x_org)
        y, s3 = self._rec('up2', y + x0, prev[3], 1)
        y = F.conv2d(torch.relu(y + head), sd['conv_img.weight'], sd['conv_img.bias'])
        y = torch.sigmoid(_bn(sd, 'bn_img', y))
        self.states = [s0, s1, s2, s3]
        self.prev_recs = y
        return y.mean(1, keepdim=True)


# ----------------------------------------------------------------------------------------------------------------
# ET-Net (model/eitr/): conv-LSTM encoder, three token scales through pre-norm transformer encoders/decoders, bilinear
# upsample decoders.  TEST INFRASTRUCTURE ONLY.
def _layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def _mha(sd, p, q_in, k_in, v_in, nhead=8):
    """nn.MultiheadAttention (batch_first=False) restated: inputs [L, N, E]; packed in_proj; softmax(q k^T / sqrt(d)) v."""
    E = q_in.shape[-1]
    W, b = sd[p + '.in_proj_weight'], sd[p + '.in_proj_bias']
    q = F.linear(q_in, W[:E], b[:E]); k = F.linear(k_in, W[E:2 * E], b[E:2 * E]); v = F.linear(v_in, W[2 * E:], b[2 * E:])
    L, N, _ = q.shape; S = k.shape[0]; d = E // nhead
    q = q.reshape(L, N * nhead, d).transpose(0, 1); k = k.reshape(S, N * nhead, d).transpose(0, 1)
    v = v.reshape(S, N * nhead, d).transpose(0, 1)
    att = torch.softmax(torch.bmm(q, k.transpose(1, 2)) / (d ** 0.5), dim=-1)
    o = torch.bmm(att, v).transpose(0, 1).reshape(L, N, E)
    return F.linear(o, sd[p + '.out_proj.weight'], sd[p + '.out_proj.bias'])


def _enc_layer(sd, p, src):
    """TransformerEncoderLayer.forward (transformer_encoder.py:64-76): pre-norm self-attention + FFN."""
    x = _layer_norm(sd, p + '.norm1', src)
    src2 = src + _mha(sd, p + '.self_attn', x, x, x)
    y = _layer_norm(sd, p + '.norm2', src2)
    return src2 + F.linear(torch.relu(F.linear(y, sd[p + '.linear1.weight'], sd[p + '.linear1.bias'])),
                           sd[p + '.linear2.weight'], sd[p + '.linear2.bias'])


def _dec_layer(sd, p, tgt, memory):
    """TransformerDecoderLayer.forward (transformer_decoder.py:66-84)."""
    x = _layer_norm(sd, p + '.norm1', tgt)
    tgt2 = tgt + _mha(sd, p + '.self_attn', x, x, x)
    q = _layer_norm(sd, p + '.norm21', tgt2); kv = _layer_norm(sd, p + '.norm22', memory)
    tgt4 = tgt2 + _mha(sd, p + '.cross_attn', q, kv, kv)
    y = _layer_norm(sd, p + '.norm3', tgt4)
    return tgt4 + F.linear(torch.relu(F.linear(y, sd[p + '.linear1.weight'], sd[p + '.linear1.bias'])),
                           sd[p + '.linear2.weight'], sd[p + '.linear2.bias'])


def sine_position_table(n_position, d_hid=256):
    """PositionalEncodingSine (position_encoding.py:14-23): float64 numpy table cast to float32."""
    import numpy as np
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    ang = pos / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    ang[:, 0::2] = np.sin(ang[:, 0::2]); ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.FloatTensor(ang)


class ETNetOracle:
    """EITR / mls_tpa (model/eitr/eitr.py:4-16, u_trans.py:13-123)."""

    def __init__(self, sd, norm=None):
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        self.norm = norm
        self.num_encoders = 3            # eval.py:152-153
        self.reset_states()

    def reset_states(self):
        self.states = [None] * 3

    def __call__(self, x):
        sd, norm = self.sd, self.norm
        x = conv_layer(sd, 'head', x, 1, 2, 'relu', norm)
        head = x
        blocks = []
        for i in range(3):
            p = f'DownsampleConv.{i}'
            x = conv_layer(sd, p + '.conv', x, 2, 2, 'relu', norm)
            st = conv_lstm(sd, p + '.recurrent_block', x, self.states[i]); x = st[0]
            self.states[i] = st
            blocks.append(x)
        n, c, H, W = head.shape
        words = [blocks[2].flatten(2).transpose(1, 2),                                            # nn.Unfold(1): the pixels
                 F.conv2d(blocks[1], sd['split1.weight'], sd['split1.bias'], stride=2).flatten(2).transpose(1, 2),
                 F.conv2d(blocks[0], sd['split2.weight'], sd['split2.bias'], stride=4).flatten(2).transpose(1, 2)]
        pos = sine_position_table(words[0].shape[1])[None]
        hs = []
        for s in range(3):
            t = (words[s] + pos).transpose(0, 1)                                                 # [L, N, E]
            for l in range(3):
                t = _enc_layer(sd, f'trans_encoder{s}.encoder.layers.{l}', t)
            hs.append(t)
        hc = []
        for s, mem in enumerate([hs[0], hs[0], hs[1]]):
            t = hs[s]
            for l in range(2):
                t = _dec_layer(sd, f'trans_decoder{s}.decoder.layers.{l}', t, mem)
            hc.append(t)
        t = (hs[0] + hs[1] + hs[2] + hc[0] + hc[1] + hc[2]) / 6
        y = t.permute(1, 2, 0).reshape(n, 256, H // 8, W // 8)                                   # '(h w) n c -> n c h w'
        for i in range(3):
            y = upsample_conv_layer(sd, f'UpsampleConv.{i}', y + blocks[2 - i], 2, 'relu', norm)
        return torch.sigmoid(conv_layer(sd, 'pred', y + head, 1, 0, None, norm))
