// Recurrent-network executor behind evr_model_* (include/evreal_hip.h).
//
// Reference: eval.py:109-158 (model construction/loading), :196-197,226-230 (reset, pad, forward,
// crop); model/unet.py:9-143, model/model.py:108-190, model/legacy.py:32-187, model/submodules.py.
//
// Creation (once per checkpoint): the reference's state_dict is re-laid for the implicit-GEMM kernels
// of conv.hip -- [Cout][tap][Cin] (K contiguous), BatchNorm (eval mode, eval.py:112) folded into
// weights and bias in fp64, ConvLSTM gate rows permuted so one 128-column tile holds the four gates of
// 32 hidden channels, ConvGRU update|reset stacked into one GEMM, ConvTranspose2d split into its four
// stride-2 sub-pixel phases -- and uploaded.
// Reset (once per sequence shape): activations (NHWC fp32) and recurrent state are allocated for
// n_seq sequences and zeroed; every layer's launch plan (ConvArgs) is built for both ping-pong parities
// and kept RESIDENT IN DEVICE MEMORY, so a frame is a fixed chain of kernel launches with 8-byte
// kernargs, no host arithmetic and no synchronisation.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "conv.h"

using namespace evr;

namespace {

struct HostTensor {
    const float* data = nullptr;
    int ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    int64_t numel() const { int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; }
};

struct DevTensor {   // NHWC activation / state
    float* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
    bool packed = false;   // PACKED activation format (conv.h): f16 hi | fp8 lo8 | fp8 x8 per 16 channels
    int c_valid = 0;       // channels evr_model_read_tensor returns (0: all; FireNet's tensors are padded 16 -> 32)
    int64_t numel() const { return (int64_t)n * h * w * c; }
};

struct Conv {   // one prepared implicit-GEMM convolution
    std::string name;
    int kc = 32;
    int cin0 = 0, cin1 = 0;     // input channels (cat)
    int n_gemm = 0, n_valid = 0;
    int k = 3, stride = 1;
    bool transposed = false;
    int wino_s2d = 0;           // exact-fp32 mode, k5 s2 layers: 1 + log2(cin / 8) when d_wino holds the space-to-depth Winograd weights
    int epi = EPI_BIAS, hidden = 0;
    ConvTaps tp;
    int x3 = 0;                 // arithmetic mode (conv.h arith_mode): 2 / 3 = weights packed for that split kernel family, 0 = fp32
    int mx_e = 0;               //   and the exponent of their packed pieces
    double useful_taps = 0;   // (tap, group) pairs that carry weights (direct-conv FLOP accounting)
    std::vector<float> w, b;    // host, prepared layout
    float* d_w = nullptr; float* d_b = nullptr;
    // k5 stride-2 convs: space-to-depth weights [n_gemm][9][4*cin] and the step program of the band kernel
    std::vector<float> w2; std::vector<unsigned> prog; int prog_steps = 0;
    float* d_w2 = nullptr; unsigned* d_prog = nullptr;
    float* d_wino = nullptr;    // exact-fp32 mode, 3x3 stride-1 layers: Winograd-domain weights (wino.hip)
    // per shape
    ConvArgs args[2];
    int wm = 4, nb = 4;
    int arg_slot = -1;
    double flops = 0.0;
};

enum StepKind { ST_HEAD, ST_CONV, ST_UPSAMPLE, ST_ADD, ST_PRED, ST_CTX, ST_CTXCONV, ST_DYN, ST_TOPACKED,
                ST_SP_NEAREST, ST_SP_SEG, ST_SP_APPLY,     // SPADE-E2VID (spade.hip)
                ST_INORM,                                  // InstanceNorm2d of the norm='IN' residual blocks
                ST_LN, ST_ATTN, ST_ADDPOS, ST_MEAN6 };     // ET-Net token kernels (etnet.hip)
struct Step {
    StepKind kind;
    int conv = -1;
    const float* a[2] = {nullptr, nullptr};   // parity-dependent operands (upsample/add/pred x)
    const float* b[2] = {nullptr, nullptr};
    float* out = nullptr;
    int h = 0, w = 0, c = 0;
    int a_packed = 0, b_packed = 0, out_packed = 0;   // activation formats of the operands
};

}  // namespace

struct evr_model {
    evr_model_desc desc;
    std::map<std::string, HostTensor> sd;
    std::vector<Conv> convs;
    // head / pred parameters
    std::vector<float> head_w, head_b, pred_w;
    float pred_b = 0.f;
    float* d_head_w = nullptr; float* d_head_b = nullptr; float* d_pred_w = nullptr;
    unsigned* d_head_wfrag = nullptr;   // head weights in MFMA-fragment order (split modes, k5 x 5 bins x 32 channels)
    int head_wfrag_e = 0;               //   and their exponent (head_pack_wfrag)
    // shape-dependent
    int n_seq = 0, H = 0, W = 0, hp = 0, wp = 0, pad_top = 0, pad_left = 0, iy0 = 0, ix0 = 0;
    int arith = 2;         // arithmetic mode of the 32-channel-chunk convolutions (conv.h arith_mode, narrowed by p6_eligible)
    bool packed = false;   // split mode: tensors between matrix-core convolutions use the PACKED format
    int fmt = 0;           //   value of the `packed` flags of those tensors (conv.h packed_fmt: 1 PACKED, 2 H2, 3 P6)
    int pred_x_packed = 0, pred_skip_packed = 0;
    std::vector<std::pair<float*, size_t>> allocs;   // (pointer, bytes): state and activations, zeroed by every reset
    std::vector<float*> shape_consts;                // per-shape constant tables (ET-Net sine table): freed with the shape, never zeroed
    std::map<std::string, DevTensor> named[2];   // debug names -> tensor valid after a frame of parity p
    std::vector<Step> steps;
    ConvArgs* d_args = nullptr;
    unsigned* d_sat = nullptr;   // [convs.size() + 1] range-guard counters of the packed producers (last: the head conv); evr_model_saturation
    int64_t frame = 0;
    std::string gate_layer; hipEvent_t gate_event = nullptr;   // evr_model_set_gate
    double flops = 0.0;
    HeadArgs head;
    const float* pred_x[2] = {nullptr, nullptr};
    const float* pred_skip[2] = {nullptr, nullptr};
    int pred_c = 0;
    int pred_fused_conv = -1;
    // HyperE2VID dynamic decoder (submodules.py:100-127)
    bool dynamic = false;
    // FireNet (16 channels) on the split kernels: every 16-channel tensor zero-padded to one 32-channel chunk (build_firenet)
    int fire_C = 0;
    std::vector<std::vector<float>> pad_store;
    double dyn_flops = 0.0;   // per launch of the dynamic-filter step (profile table)
    std::vector<float> ctx_w, ctx_b, fb_bases;
    float* d_ctx_w = nullptr; float* d_ctx_b = nullptr; float* d_bases = nullptr;
    float* prev_rec = nullptr;
    HeadArgs ctxconv;
    CtxArgs ctx;   // conv whose epilogue carries the prediction layer (-1: standalone pred kernel)
    // SPADE-E2VID (model/spade_e2v.py): explicit padded input, segmentation map (first frame: normalised input
    // channels; then the previous 3-channel reconstruction), its half-resolution copy, the SPADE MLP heads
    float* sp_xpad = nullptr; float* sp_xorg = nullptr; float* sp_xorg_half = nullptr;
    std::vector<HeadArgs> sp_seg;                 // mlp_shared convs (direct small-Cin kernel), one per UpConvLayer3
    std::vector<std::vector<float>> sp_seg_w, sp_seg_b;
    std::vector<float*> sp_seg_dw, sp_seg_db;
    std::vector<float> sp_pred_w; float sp_pred_b[3] = {0.f, 0.f, 0.f}; float* d_sp_pred_w = nullptr;
    SpadePredArgs sp_pred;
    // ET-Net (model/eitr): LayerNorm parameters, attention launches, the sine position table
    std::vector<std::pair<float*, float*>> et_ln;      // device (weight, bias) per LayerNorm, indexed by Step::conv
    std::vector<AttnArgs> et_attn;                     // indexed by Step::conv
    float* et_pos = nullptr;                           // [L, 256] for the current shape
    const float* et_mean_in[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const float* sp_skip[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    // per-layer event timing (evr_model_profile_*)
    bool prof_on = false;
    std::string prof_filter;
    struct ProfPair { int conv; hipEvent_t a, b; };
    std::vector<ProfPair> prof_pending;
    std::vector<double> prof_ms;
    std::vector<int64_t> prof_n;

    ~evr_model() { release_shape(); for (auto& c : convs) { if (c.d_w) (void)hipFree(c.d_w); if (c.d_b) (void)hipFree(c.d_b); if (c.d_wino) (void)hipFree(c.d_wino);
                                            if (c.d_w2) (void)hipFree(c.d_w2); if (c.d_prog) (void)hipFree(c.d_prog); }
                   if (d_ctx_w) (void)hipFree(d_ctx_w); if (d_ctx_b) (void)hipFree(d_ctx_b); if (d_bases) (void)hipFree(d_bases);
                   if (d_head_wfrag) (void)hipFree(d_head_wfrag); if (d_sat) (void)hipFree(d_sat);
                   for (auto& pr : et_ln) { (void)hipFree(pr.first); (void)hipFree(pr.second); }
                   for (float* q : sp_seg_dw) (void)hipFree(q); for (float* q : sp_seg_db) (void)hipFree(q); if (d_sp_pred_w) (void)hipFree(d_sp_pred_w);
                   if (d_head_w) (void)hipFree(d_head_w); if (d_head_b) (void)hipFree(d_head_b); if (d_pred_w) (void)hipFree(d_pred_w); }
    void release_shape() {
        for (auto& pr : allocs) (void)hipFree(pr.first);
        allocs.clear();
        for (float* q : shape_consts) (void)hipFree(q);
        shape_consts.clear();
        if (d_args) { (void)hipFree(d_args); d_args = nullptr; }
        steps.clear(); named[0].clear(); named[1].clear();
        n_seq = 0; prev_rec = nullptr; sp_xpad = sp_xorg = sp_xorg_half = nullptr; et_pos = nullptr; et_attn.clear();
    }
};

namespace {

// ------------------------------------------------------------------------------------------------
// state_dict access
int find(const evr_model* m, const std::string& name, const HostTensor** out, bool required = true) {
    auto it = m->sd.find(name);
    if (it == m->sd.end()) {
        *out = nullptr;
        if (required) { set_error("state_dict is missing '%s'", name.c_str()); return EVR_ERR_MISSING_TENSOR; }
        return EVR_OK;
    }
    *out = &it->second;
    return EVR_OK;
}

struct Affine { std::vector<double> scale, shift; };   // y = conv*scale + shift (bias and BN folded)

// bias (optional) then BatchNorm in eval mode (optional): y = ((conv + b) - mean)/sqrt(var+eps)*gamma + beta
int make_affine(const evr_model* m, const std::string& bias_name, const std::string& bn_prefix, bool bn, int cout, Affine* out, bool bn_affine = true) {
    out->scale.assign(cout, 1.0); out->shift.assign(cout, 0.0);
    const HostTensor* b = nullptr;
    int rc = find(m, bias_name, &b, false);
    if (rc) return rc;
    if (b) { EVR_REQUIRE(b->numel() == cout, "'%s' has %lld elements, expected %d", bias_name.c_str(), (long long)b->numel(), cout);
             for (int i = 0; i < cout; ++i) out->shift[i] = b->data[i]; }
    if (bn) {
        // bn_affine = false: BatchNorm2d(affine=False) / InstanceNorm2d(track_running_stats=True) in eval mode --
        // running statistics only, gamma = 1, beta = 0
        const HostTensor *g = nullptr, *be = nullptr, *mu, *var;
        if (bn_affine && (rc = find(m, bn_prefix + ".weight", &g))) return rc;
        if (bn_affine && (rc = find(m, bn_prefix + ".bias", &be))) return rc;
        if ((rc = find(m, bn_prefix + ".running_mean", &mu))) return rc;
        if ((rc = find(m, bn_prefix + ".running_var", &var))) return rc;
        EVR_REQUIRE((!g || g->numel() == cout) && (!be || be->numel() == cout) && mu->numel() == cout && var->numel() == cout, "BatchNorm '%s' size mismatch", bn_prefix.c_str());
        for (int i = 0; i < cout; ++i) {
            const double s = (g ? (double)g->data[i] : 1.0) / std::sqrt((double)var->data[i] + 1e-5);
            out->shift[i] = (out->shift[i] - (double)mu->data[i]) * s + (be ? (double)be->data[i] : 0.0);
            out->scale[i] = s;
        }
    } else {
        EVR_REQUIRE(b != nullptr || bias_name.empty(), "missing bias '%s'", bias_name.c_str());
    }
    return EVR_OK;
}

int upload(const std::vector<float>& h, float** d) {
    EVR_HIP(hipMalloc((void**)d, h.size() * sizeof(float) + 64));
    EVR_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return EVR_OK;
}

int round_up(int v, int a) { return (v + a - 1) / a * a; }

// Conv2d weight [cout, cin, k, k] -> [n_gemm][k*k][cin]; row_of(co) gives the GEMM row of output channel co.
template <typename RowFn>
int prep_conv2d(Conv& c, const HostTensor* w, const Affine& af, int cout, int cin, int k, int pad, RowFn row_of, int n_gemm) {
    EVR_REQUIRE(w->ndim == 4 && w->shape[0] == cout && w->shape[1] == cin && w->shape[2] == k && w->shape[3] == k,
                "'%s': weight shape [%lld,%lld,%lld,%lld], expected [%d,%d,%d,%d]", c.name.c_str(), (long long)w->shape[0],
                (long long)w->shape[1], (long long)w->shape[2], (long long)w->shape[3], cout, cin, k, k);
    const int taps = k * k;
    EVR_REQUIRE(taps <= MAX_TAPS, "kernel_size %d too large", k);
    c.w.assign((size_t)n_gemm * taps * cin, 0.f);
    c.b.assign(n_gemm, 0.f);
    for (int co = 0; co < cout; ++co) {
        const int row = row_of(co);
        c.b[row] = (float)af.shift[co];
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t)
                c.w[((size_t)row * taps + t) * cin + ci] = (float)((double)w->data[((size_t)co * cin + ci) * taps + t] * af.scale[co]);
    }
    memset(&c.tp, 0, sizeof(c.tp));
    c.tp.ntaps = taps; c.tp.ngroups = 1; c.tp.grp_cols = n_gemm;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) c.tp.set_tap(ky * k + kx, ky - pad, kx - pad, 1);
    c.useful_taps = taps;
    c.n_gemm = n_gemm; c.k = k;
    return EVR_OK;
}

// ConvTranspose2d(k, stride 2, padding pad, output_padding 1) weight [cin, cout, k, k] -> ONE GEMM whose columns are
// phase-major: group g = py*2+px holds the Cout channels of output pixels (2my+py, 2mx+px).
// out[2my+py] takes ky with (py + pad - ky) even, from input row my + (py + pad - ky)/2; the union of those input
// offsets over the four phases is the shared tap list (3x3 for k = 5), tap_groups marks which phases use a tap.
int prep_tconv(Conv& c, const HostTensor* w, const Affine& af, int cin, int cout, int k, int pad, int grp_cols) {
    EVR_REQUIRE(w->ndim == 4 && w->shape[0] == cin && w->shape[1] == cout && w->shape[2] == k && w->shape[3] == k,
                "'%s': transposed weight shape mismatch", c.name.c_str());
    memset(&c.tp, 0, sizeof(c.tp));
    c.tp.ngroups = 4; c.tp.grp_cols = grp_cols;
    // more than 32 output channels: interleave the four phases per 128-column tile (conv.h ConvTaps::inter) -- EVR_TCONV_INTER=0: phase-major
    static const bool inter_on = getenv("EVR_TCONV_INTER") ? atoi(getenv("EVR_TCONV_INTER")) != 0 : true;
    const bool inter = inter_on && c.kc == 32 && cout == grp_cols && cout % 32 == 0 && cout > 32;
    auto row_of = [&](int g, int co) { return inter ? (co / 32) * 128 + g * 32 + co % 32 : g * grp_cols + co; };
    if (inter) { c.tp.inter = 1; c.tp.grp_cols = 32; }
    // shared taps
    std::vector<std::pair<int, int>> taps;
    auto tap_index = [&](int dy, int dx) {
        for (size_t i = 0; i < taps.size(); ++i) if (taps[i].first == dy && taps[i].second == dx) return (int)i;
        taps.push_back({dy, dx});
        return (int)taps.size() - 1;
    };
    struct Use { int g, t, ky, kx; };
    std::vector<Use> uses;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            c.tp.grp_ofy[py * 2 + px] = py; c.tp.grp_ofx[py * 2 + px] = px;
            for (int ky = 0; ky < k; ++ky) {
                if (((py + pad - ky) & 1) != 0) continue;
                for (int kx = 0; kx < k; ++kx) {
                    if (((px + pad - kx) & 1) != 0) continue;
                    uses.push_back({py * 2 + px, tap_index((py + pad - ky) / 2, (px + pad - kx) / 2), ky, kx});
                }
            }
        }
    EVR_REQUIRE((int)taps.size() <= MAX_TAPS, "transposed conv: too many taps");
    {   // canonical row-major tap order (dy, then dx): what the band kernel of conv.hip walks
        std::vector<std::pair<int, int>> sorted = taps;
        std::sort(sorted.begin(), sorted.end());
        std::vector<int> remap(taps.size());
        for (size_t i = 0; i < taps.size(); ++i)
            remap[i] = (int)(std::find(sorted.begin(), sorted.end(), taps[i]) - sorted.begin());
        for (Use& u : uses) u.t = remap[u.t];
        taps = sorted;
    }
    const int nt = (int)taps.size();
    c.tp.ntaps = nt;
    for (int t = 0; t < nt; ++t) c.tp.set_tap(t, taps[t].first, taps[t].second, 0);
    const int n_gemm = 4 * grp_cols;
    c.w.assign((size_t)n_gemm * nt * cin, 0.f);
    c.b.assign(n_gemm, 0.f);
    for (int g = 0; g < 4; ++g)
        for (int co = 0; co < cout; ++co) c.b[row_of(g, co)] = (float)af.shift[co];
    for (const Use& u : uses) {
        c.tp.tap_groups[u.t] |= 1 << u.g;
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                c.w[((size_t)row_of(u.g, co) * nt + u.t) * cin + ci] =
                    (float)((double)w->data[(((size_t)ci * cout + co) * k + u.ky) * k + u.kx] * af.scale[co]);
    }
    c.useful_taps = (double)uses.size() / 4.0;   // per output pixel of one phase on average: k*k/4
    c.n_gemm = n_gemm; c.k = k; c.transposed = true;
    return EVR_OK;
}

// conv(k5, stride 2, pad 2) in space-to-depth form (conv.hip, conv_band_prog_kernel): a 3x3 stride-1 convolution
// over 2x2 pixel blocks with 4*cin phase-major channels.  in(2y + ky - 2, .) = block row y + dy, phase py with
// 2*dy + py = ky - 2; the (dy = +1, py = 1) and (dx = +1, px = 1) combinations do not exist and get no step.
void prep_s2d(Conv& c) {
    const int cin = c.cin0, nch = cin / 32, nch2 = 4 * nch, k2 = 9 * 4 * cin;
    c.w2.assign((size_t)c.n_gemm * k2, 0.f);
    for (int row = 0; row < c.n_gemm; ++row)
        for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx)
            for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px) {
                const int ky = 2 * dy + py + 2, kx = 2 * dx + px + 2;
                if (ky < 0 || ky > 4 || kx < 0 || kx > 4) continue;
                const float* src = &c.w[((size_t)row * 25 + ky * 5 + kx) * cin];
                float* dst = &c.w2[((size_t)row * 9 + (dy + 1) * 3 + dx + 1) * 4 * cin + (py * 2 + px) * cin];
                memcpy(dst, src, (size_t)cin * sizeof(float));
            }
    // the program: bands = (phase, channel chunk, dy) in K order, steps = their dx taps
    struct Band { int src, dy; std::vector<int> taps; };
    std::vector<Band> bands;
    for (int cc = 0; cc < nch2; ++cc) {
        const int phase = cc / nch, py = phase >> 1, px = phase & 1;
        for (int dy = -1; dy <= 1; ++dy) {
            if (2 * dy + py > 2) continue;
            Band b; b.src = py | (px << 1) | ((cc % nch) << 2); b.dy = dy;
            for (int dx = -1; dx <= 1; ++dx) if (2 * dx + px <= 2) b.taps.push_back((dy + 1) * 3 + dx + 1);
            bands.push_back(b);
        }
    }
    // packed entries (conv.h): entry 0 = first band, then one per step
    auto next_bits = [](const Band& b) { return ((unsigned)b.src << 16) | ((unsigned)(b.dy + 1) << 26); };
    c.prog.assign(BAND_PROG_MAX, 0u);
    size_t n = 0;
    c.prog[n++] = next_bits(bands[0]);
    for (size_t bi = 0; bi < bands.size(); ++bi) {
        const Band& b = bands[bi];
        const int phase = (b.src & 1) * 2 + ((b.src >> 1) & 1), cc = phase * nch + (b.src >> 2);
        for (size_t ti = 0; ti < b.taps.size() && n < (size_t)BAND_PROG_MAX; ++ti) {
            unsigned e = (unsigned)b.taps[ti] | ((unsigned)cc << 8);
            if (ti == 0) {
                e |= 16u;
                if (bi + 1 < bands.size()) {
                    e |= 32u | next_bits(bands[bi + 1]);
                    if (b.taps.size() >= 2) e |= 64u;
                }
            }
            c.prog[n++] = e;
        }
    }
    c.prog_steps = (int)n - 1;
}

int finish_conv(evr_model* m, Conv& c) {
    int rc;
    // arithmetic mode: split (f16 + MX-fp8 corrections, conv.h) for the 32-channel-chunk convolutions unless EVR_FP32=1
    c.x3 = (c.kc == 32) ? m->arith : 0;
    // 16-channel 3x3 layers (FireNet): v_mfma_f32_32x32x16_f16 contracts exactly one 16-channel H2 group, so in the split modes they
    // run the three-f16-product arithmetic on UNPADDED tensors (conv.hip conv3x3_c16_kernel) whatever the mode of the 32-channel
    // layers -- fp32-grade (the goldens hold 1e-5), and no longer the 1/16-rate fp32 MFMA.  EVR_FIRENET_H3=0: the exact-fp32 path.
    static const bool fire_h3 = getenv("EVR_FIRENET_H3") ? atoi(getenv("EVR_FIRENET_H3")) != 0 : true;
    if (c.kc == 16 && m->arith != 0 && fire_h3 && c.k == 3 && c.stride == 1 && !c.transposed && c.cin0 == 16 && (c.cin1 == 0 || c.cin1 == 16) &&
        (m->desc.arch == EVR_ARCH_FIRENET_LEGACY || m->desc.arch == EVR_ARCH_FIRENET) && c.n_gemm == 32)
        c.x3 = 3;
    if (c.x3 && c.k == 5 && c.stride == 2 && !c.transposed && c.cin1 == 0 && c.n_gemm % 64 == 0 && 25 * (c.cin0 / 32) < BAND_PROG_MAX - 2) {
        prep_s2d(c);
        const int e2 = pack_weights_for(c.x3, c.w2);      // (the same values rearranged: the same exponent as c.w below)
        std::vector<float> probe(c.w);
        EVR_REQUIRE(pack_weights_for(c.x3, probe) == e2, "conv %s: the two weight layouts disagree on the packing exponent", c.name.c_str());
        if ((rc = upload(c.w2, &c.d_w2))) return rc;
        EVR_HIP(hipMalloc((void**)&c.d_prog, c.prog.size() * sizeof(unsigned)));
        EVR_HIP(hipMemcpy(c.d_prog, c.prog.data(), c.prog.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    }
    // exact-fp32 mode: the 3x3 stride-1 layers whose shape wino.hip covers (ConvLSTM gates, residual convolutions) get their
    // Winograd-domain weights too -- G g G^T in fp64 -- and run F(2x2, 3x3): 2.25x fewer fp32 MFMAs (EVR_WINO=0: direct form only)
    // (round 6, later: the k5 s2 transposed decoders too -- their four phases are 3x3 convolutions on the input grid sharing one input
    // transform, 16 instead of 25 multiplies per 2x2 outputs and phase; EVR_WINO_TCONV=0: direct form)
    static const bool wino_tconv = getenv("EVR_WINO_TCONV") ? atoi(getenv("EVR_WINO_TCONV")) != 0 : true;
    const bool wino_plain = c.k == 3 && !c.transposed && c.tp.ngroups == 1 && c.n_valid == c.n_gemm &&
        ((c.epi == EPI_LSTM && c.hidden % 16 == 0) || c.epi == EPI_BIAS || c.epi == EPI_BIAS_RELU || c.epi == EPI_RESIDUAL_RELU);
    const bool wino_t = wino_tconv && c.transposed && c.tp.ngroups == 4 && c.tp.ntaps == 9 && c.cin1 == 0 && c.n_valid % 32 == 0 && c.n_gemm == 4 * c.n_valid &&
        (c.epi == EPI_BIAS || c.epi == EPI_BIAS_RELU);
    // ... and the k5 stride-2 encoders: in space-to-depth form (prep_s2d's weight layout) they are 3x3 stride-1 convolutions over 2x2
    // pixel blocks with 4 cin channels, 36 (tap, phase) pairs of which 25 carry weights -- Winograd needs 16 (EVR_WINO_S2D=0: direct form)
    static const bool wino_s2d_on = getenv("EVR_WINO_S2D") ? atoi(getenv("EVR_WINO_S2D")) != 0 : true;
    if (c.x3 == 0 && c.kc == 32 && wino_enabled() && wino_s2d_on && c.k == 5 && c.stride == 2 && !c.transposed && c.tp.ngroups == 1 && c.tp.ntaps == 25 &&
        c.cin1 == 0 && c.cin0 % 8 == 0 && ((c.cin0 / 8) & (c.cin0 / 8 - 1)) == 0 && c.n_gemm % 64 == 0 && c.n_valid == c.n_gemm &&
        (c.epi == EPI_BIAS || c.epi == EPI_BIAS_RELU)) {
        const int cin = c.cin0;
        std::vector<float> w2((size_t)c.n_gemm * 9 * 4 * cin, 0.f), u;
        for (int row = 0; row < c.n_gemm; ++row)
            for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx)
                for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px) {
                    const int ky = 2 * dy + py + 2, kx = 2 * dx + px + 2;
                    if (ky < 0 || ky > 4 || kx < 0 || kx > 4) continue;
                    memcpy(&w2[((size_t)row * 9 + (dy + 1) * 3 + dx + 1) * 4 * cin + (py * 2 + px) * cin], &c.w[((size_t)row * 25 + ky * 5 + kx) * cin],
                           (size_t)cin * sizeof(float));
                }
        wino_pack_weights(w2, c.n_gemm, 4 * cin, 0, u);
        if ((rc = upload(u, &c.d_wino))) return rc;
        c.wino_s2d = 1 + __builtin_ctz((unsigned)(cin / 8));
    }
    if (c.x3 == 0 && c.kc == 32 && wino_enabled() && c.stride == 1 && (wino_plain || wino_t) && c.cin0 % 8 == 0 &&
        (c.cin1 == 0 || c.cin1 == c.cin0) && c.n_gemm % 64 == 0) {
        std::vector<float> u;
        wino_pack_weights(c.w, c.n_gemm, c.cin0 + c.cin1, c.epi == EPI_LSTM ? c.hidden : 0, u);
        if ((rc = upload(u, &c.d_wino))) return rc;
    }
    if (c.x3) c.mx_e = pack_weights_for(c.x3, c.w);
    // mode 3 accumulates products scaled by 2^(e_w + H2_ACT_EXP): the accumulators start at the bias in that scale (exact:
    // a power of two) and the epilogue multiplies by ConvArgs::acc_scale
    if (c.x3 == 3) for (float& b : c.b) b = std::ldexp(b, c.mx_e + H2_ACT_EXP);
    if (c.x3 == 4) for (float& b : c.b) b = std::ldexp(b, c.mx_e);      // (mode 4: weights scaled by 2^e_w, activations unscaled)
    if ((rc = upload(c.w, &c.d_w))) return rc;
    if ((rc = upload(c.b, &c.d_b))) return rc;
    m->convs.push_back(std::move(c));
    return EVR_OK;
}

int pick_kc(int c0, int c1) { return (c0 % 32 == 0 && (c1 == 0 || c1 % 32 == 0)) ? 32 : 16; }

// plain conv: prefix.{weight,bias}; bn under bn_prefix.  The reference's ConvLayer/ResidualBlock drop the conv bias
// when they use BN (submodules.py:13,155); keep_bias_with_bn covers nn.Sequential(Conv2d(bias), BatchNorm2d).
int add_conv(evr_model* m, const std::string& name, const std::string& wname, const std::string& bname, const std::string& bn_prefix,
             bool bn, int cin, int cout, int k, int stride, int epi, bool keep_bias_with_bn = false, bool bn_affine = true) {
    Conv c; c.name = name;
    const HostTensor* w; int rc;
    if ((rc = find(m, wname, &w))) return rc;
    Affine af;
    if ((rc = make_affine(m, (bn && !keep_bias_with_bn) ? std::string() : bname, bn_prefix, bn, cout, &af, bn_affine))) return rc;
    EVR_REQUIRE(cin % 16 == 0, "'%s': %d input channels (need a multiple of 16)", name.c_str(), cin);
    c.kc = pick_kc(cin, 0);
    c.cin0 = cin; c.cin1 = 0; c.stride = stride; c.epi = epi; c.n_valid = cout;
    if ((rc = prep_conv2d(c, w, af, cout, cin, k, k / 2, [](int co) { return co; }, round_up(cout, 32)))) return rc;
    return finish_conv(m, c);
}

int add_tconv(evr_model* m, const std::string& name, const std::string& prefix, bool bn, int cin, int cout, int k, bool inn = false) {
    Conv c; c.name = name;
    const HostTensor* w; int rc;
    if ((rc = find(m, prefix + ".transposed_conv2d.weight", &w))) return rc;
    Affine af;
    if ((rc = make_affine(m, bn ? std::string() : prefix + ".transposed_conv2d.bias", prefix + ".norm_layer", bn || inn, cout, &af, !inn))) return rc;
    c.kc = pick_kc(cin, 0);
    c.cin0 = cin; c.stride = 1; c.epi = EPI_BIAS_RELU; c.n_valid = cout;
    if ((rc = prep_tconv(c, w, af, cin, cout, k, k / 2, round_up(cout, 32)))) return rc;
    return finish_conv(m, c);
}

// ConvLSTM Gates conv (submodules.py:205,227-231): rows permuted to (c/32)*128 + gate*32 + c%32
int add_lstm(evr_model* m, const std::string& name, const std::string& prefix, int C) {
    EVR_REQUIRE(C % 32 == 0, "ConvLSTM with %d hidden channels (need a multiple of 32)", C);
    Conv c; c.name = name;
    const HostTensor* w; int rc;
    if ((rc = find(m, prefix + ".Gates.weight", &w))) return rc;
    Affine af;
    if ((rc = make_affine(m, prefix + ".Gates.bias", "", false, 4 * C, &af))) return rc;
    c.kc = 32; c.cin0 = C; c.cin1 = C; c.stride = 1; c.epi = EPI_LSTM; c.hidden = C; c.n_valid = 4 * C;
    if ((rc = prep_conv2d(c, w, af, 4 * C, 2 * C, 3, 1, [C](int co) { const int g = co / C, ch = co % C; return (ch / 32) * 128 + g * 32 + (ch % 32); }, 4 * C))) return rc;
    return finish_conv(m, c);
}

// ConvGRU (submodules.py:255-285): GEMM 1 = [update | reset] over (x|h); GEMM 2 = candidate over (x|h*r)
int add_gru(evr_model* m, const std::string& name, const std::string& prefix, int C) {
    EVR_REQUIRE(C % 16 == 0, "ConvGRU with %d hidden channels (need a multiple of 16)", C);
    int rc;
    const HostTensor *wz, *wr, *wo;
    if ((rc = find(m, prefix + ".update_gate.weight", &wz))) return rc;
    if ((rc = find(m, prefix + ".reset_gate.weight", &wr))) return rc;
    if ((rc = find(m, prefix + ".out_gate.weight", &wo))) return rc;
    Affine az, ar, ao;
    if ((rc = make_affine(m, prefix + ".update_gate.bias", "", false, C, &az))) return rc;
    if ((rc = make_affine(m, prefix + ".reset_gate.bias", "", false, C, &ar))) return rc;
    if ((rc = make_affine(m, prefix + ".out_gate.bias", "", false, C, &ao))) return rc;
    {
        Conv c; c.name = name + ".zr";
        c.kc = pick_kc(C, C); c.cin0 = C; c.cin1 = C; c.stride = 1; c.epi = EPI_GRU_ZR; c.hidden = C; c.n_valid = 2 * C;
        const int ng = round_up(2 * C, 32);
        Conv tmp; tmp.name = c.name;
        if ((rc = prep_conv2d(c, wz, az, C, 2 * C, 3, 1, [](int co) { return co; }, ng))) return rc;
        if ((rc = prep_conv2d(tmp, wr, ar, C, 2 * C, 3, 1, [C](int co) { return C + co; }, ng))) return rc;
        for (size_t i = (size_t)C * 9 * 2 * C; i < (size_t)2 * C * 9 * 2 * C; ++i) c.w[i] = tmp.w[i];
        for (int i = C; i < 2 * C; ++i) c.b[i] = tmp.b[i];
        if ((rc = finish_conv(m, c))) return rc;
    }
    {
        Conv c; c.name = name + ".out";
        c.kc = pick_kc(C, C); c.cin0 = C; c.cin1 = C; c.stride = 1; c.epi = EPI_GRU_OUT; c.hidden = C; c.n_valid = C;
        if ((rc = prep_conv2d(c, wo, ao, C, 2 * C, 3, 1, [](int co) { return co; }, round_up(C, 32)))) return rc;
        if ((rc = finish_conv(m, c))) return rc;
    }
    return EVR_OK;
}

int conv_index(const evr_model* m, const std::string& name) {
    for (size_t i = 0; i < m->convs.size(); ++i) if (m->convs[i].name == name) return (int)i;
    return -1;
}

// head: Conv2d(num_bins -> C, k) [B*k*k][C]; pred: 1x1 C -> 1 (+BN)
// (head_norm: prefix of a norm layer behind the head convolution -- ET-Net's ConvLayer head takes `norm`, u_trans.py:19; empty: none)
int prep_head_pred(evr_model* m, const std::string& head_prefix, const std::string& pred_prefix, bool pred_bn, int C, bool pred_in = false,
                   const std::string& head_norm = std::string()) {
    const evr_model_desc& d = m->desc;
    const HostTensor* w; int rc;
    if ((rc = find(m, head_prefix + ".weight", &w))) return rc;
    const int B = d.num_bins, k = d.kernel_size;
    EVR_REQUIRE(w->ndim == 4 && w->shape[0] == C && w->shape[1] == B && w->shape[2] == k && w->shape[3] == k, "head weight shape mismatch");
    Affine af;
    if (head_norm.empty()) rc = make_affine(m, head_prefix + ".bias", "", false, C, &af);
    else rc = make_affine(m, pred_bn ? std::string() : head_prefix + ".bias", head_norm, true, C, &af, !pred_in);    // ConvLayer: no conv bias with BN (submodules.py:13)
    if (rc) return rc;
    m->head_w.assign((size_t)B * k * k * C, 0.f); m->head_b.assign(C, 0.f);
    for (int co = 0; co < C; ++co) {
        m->head_b[co] = (float)af.shift[co];
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < k * k; ++t) m->head_w[((size_t)b * k * k + t) * C + co] = (float)((double)w->data[((size_t)co * B + b) * k * k + t] * af.scale[co]);
    }
    if ((rc = find(m, pred_prefix + ".conv2d.weight", &w))) return rc;
    EVR_REQUIRE(w->numel() == C, "pred weight has %lld elements, expected %d", (long long)w->numel(), C);
    Affine ap;
    if ((rc = make_affine(m, pred_bn ? std::string() : pred_prefix + ".conv2d.bias", pred_prefix + ".norm_layer", pred_bn || pred_in, 1, &ap, !pred_in))) return rc;
    m->pred_w.assign(C, 0.f);
    for (int c = 0; c < C; ++c) m->pred_w[c] = (float)((double)w->data[c] * ap.scale[0]);
    m->pred_b = (float)ap.shift[0];
    if ((rc = upload(m->head_w, &m->d_head_w))) return rc;
    if (m->arith != 0 && B == 5 && k == 5 && C == 32) {
        std::vector<unsigned> wf;
        m->head_wfrag_e = head_pack_wfrag(m->head_w.data(), B, wf);
        EVR_HIP(hipMalloc((void**)&m->d_head_wfrag, wf.size() * sizeof(unsigned)));
        EVR_HIP(hipMemcpy(m->d_head_wfrag, wf.data(), wf.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    }
    if ((rc = upload(m->head_b, &m->d_head_b))) return rc;
    if ((rc = upload(m->pred_w, &m->d_pred_w))) return rc;
    return EVR_OK;
}

int build_unet(evr_model* m) {
    const evr_model_desc& d = m->desc;
    const std::string pre = "unetrecurrent.";
    const bool bn = d.norm == EVR_NORM_BN;
    // norm='IN' (submodules.py:22-23,48-49,79-80): the conv layers carry InstanceNorm2d(track_running_stats=True), which in
    // eval mode (eval.py:112) is a fixed per-channel affine from the running statistics -> folded like BatchNorm (gamma = 1,
    // beta = 0, conv bias kept); the residual blocks carry a TRUE InstanceNorm2d (:160-162) -> ST_INORM steps
    const bool inn = d.norm == EVR_NORM_IN;
    // (the dynamic decoder combines with either norm: DynamicUpsampleLayer, submodules.py:100-127, carries no norm layer)
    const int E = d.num_encoders, base = d.base_num_channels, k = d.kernel_size;
    EVR_REQUIRE(E >= 1 && E <= 6 && base % 32 == 0, "UNetRecurrent: num_encoders %d / base_num_channels %d unsupported", E, base);
    int rc;
    if ((rc = prep_head_pred(m, pre + "head.conv2d", pre + "pred", bn, base, inn))) return rc;
    for (int i = 0; i < E; ++i) {
        const int cin = base << i, cout = base << (i + 1);
        const std::string p = pre + "encoders." + std::to_string(i);
        if ((rc = add_conv(m, "enc" + std::to_string(i) + ".conv", p + ".conv.conv2d.weight", p + ".conv.conv2d.bias", p + ".conv.norm_layer", bn || inn, cin, cout, k, 2, EPI_BIAS_RELU, inn, !inn))) return rc;
        if (d.recurrent_block == EVR_REC_CONVLSTM) rc = add_lstm(m, "enc" + std::to_string(i) + ".rec", p + ".recurrent_block", cout);
        else rc = add_gru(m, "enc" + std::to_string(i) + ".rec", p + ".recurrent_block", cout);
        if (rc) return rc;
    }
    const int cm = base << E;
    for (int i = 0; i < d.num_residual_blocks; ++i) {
        const std::string p = pre + "resblocks." + std::to_string(i), n = "res" + std::to_string(i);
        if ((rc = add_conv(m, n + ".conv1", p + ".conv1.weight", p + ".conv1.bias", p + ".bn1", bn, cm, cm, 3, 1, inn ? EPI_BIAS : EPI_BIAS_RELU))) return rc;
        if ((rc = add_conv(m, n + ".conv2", p + ".conv2.weight", p + ".conv2.bias", p + ".bn2", bn, cm, cm, 3, 1, inn ? EPI_BIAS : EPI_RESIDUAL_RELU))) return rc;
    }
    m->dynamic = (d.reserved[1] & 1) != 0;
    for (int i = 0; i < E; ++i) {
        const int cin = base << (E - i), cout = base << (E - i - 1);
        const std::string p = pre + "decoders." + std::to_string(i), n = "dec" + std::to_string(i);
        if (i == 0 && m->dynamic) {
            // DynamicUpsampleLayer: context fusion conv (direct, 6 -> 32), bases_net (2 convs + BN + tanh), FB bases,
            // compositional 1x1 conv over in_channels * num_atoms
            EVR_REQUIRE(E == 3, "the dynamic decoder needs num_encoders == 3 (context is fused at 1/4 resolution)");
            const HostTensor *cw, *cb, *bs;
            const int B1 = d.num_bins + 1;
            if ((rc = find(m, p + ".context_fusion.conv.weight", &cw))) return rc;
            if ((rc = find(m, p + ".context_fusion.conv.bias", &cb))) return rc;
            if ((rc = find(m, p + ".dynamic_atom_generation.bases", &bs))) return rc;
            EVR_REQUIRE(cw->ndim == 4 && cw->shape[0] == 32 && cw->shape[1] == B1 && cw->shape[2] == 3 && cb->numel() == 32, "context_fusion conv shape mismatch");
            EVR_REQUIRE(bs->ndim == 2 && bs->shape[0] == 12 && bs->shape[1] == 25, "Fourier-Bessel bases must be [12,25]");
            m->ctx_w.assign((size_t)B1 * 9 * 32, 0.f); m->ctx_b.assign(cb->data, cb->data + 32);
            for (int co = 0; co < 32; ++co)
                for (int b = 0; b < B1; ++b)
                    for (int t = 0; t < 9; ++t) m->ctx_w[((size_t)b * 9 + t) * 32 + co] = cw->data[((size_t)co * B1 + b) * 9 + t];
            m->fb_bases.assign(bs->data, bs->data + 300);
            if ((rc = upload(m->ctx_w, &m->d_ctx_w))) return rc;
            if ((rc = upload(m->ctx_b, &m->d_ctx_b))) return rc;
            if ((rc = upload(m->fb_bases, &m->d_bases))) return rc;
            const std::string bnet = p + ".dynamic_atom_generation.bases_net";
            if ((rc = add_conv(m, n + ".bn1", bnet + ".0.weight", bnet + ".0.bias", bnet + ".1", true, 32, 64, 3, 1, EPI_BIAS_TANH, true))) return rc;
            if ((rc = add_conv(m, n + ".bn2", bnet + ".3.weight", bnet + ".3.bias", bnet + ".4", true, 64, 72, 3, 1, EPI_BIAS_TANH, true))) return rc;
            if ((rc = add_conv(m, n, p + ".dynamic_conv.compositional_coefficients", p + ".dynamic_conv.bias", "", false, cin * 6, cout, 1, 1, EPI_BIAS_RELU))) return rc;
            continue;
        }
        if (d.use_upsample_conv) rc = add_conv(m, n, p + ".conv2d.weight", p + ".conv2d.bias", p + ".norm_layer", bn || inn, cin, cout, k, 1, EPI_BIAS_RELU, inn, !inn);
        else rc = add_tconv(m, n, p, bn, cin, cout, k, inn);
        if (rc) return rc;
    }
    return EVR_OK;
}

// FireNet's layers have 16 channels -- half a 32-channel K chunk, so they run on the exact-fp32 MFMA (kc = 16).  With
// EVR_FIRENET_PAD32=1 (split modes) every 16-channel tensor is instead ZERO-PADDED to 32 channels at model creation: weights
// [co, ci, k, k] get zero rows for co 16..31 and zero columns for the padded inputs (a cat(x, h) input of 16 + 16 becomes
// 32 + 32 with h at 32..47), biases zeros.  The padded channels stay exactly zero through every layer (relu(0) = 0; ConvGRU:
// h' = 0.5 h + 0.5 tanh(0) = 0 from the zero state), so the real channels see the reference's arithmetic, now on the split
// kernels.  Measured at 240x180, 64 sequences: 12.0k frames/s against 14.2k on the fp32 path (4x the multiply-adds on tiles
// sized for wider layers), so it is NOT the default; it exists because the shipped FireNet checkpoints are the only TRAINED
// weights available offline, and this is how the split arithmetic is exercised on them (tests/test_gpu_modes.py).
static void firenet_pad_state_dict(evr_model* m, const std::string& head, const std::string& pred) {
    for (auto& kv : m->sd) {
        HostTensor& t = kv.second;
        const std::string& name = kv.first;
        if (t.ndim == 1 && t.shape[0] == 16) {                          // biases of 16-channel layers
            m->pad_store.emplace_back(32, 0.f);
            memcpy(m->pad_store.back().data(), t.data, 16 * sizeof(float));
            t.data = m->pad_store.back().data(); t.shape[0] = 32;
        } else if (t.ndim == 4) {
            const int co = (int)t.shape[0], ci = (int)t.shape[1], kk = (int)(t.shape[2] * t.shape[3]);
            const bool is_head = name == head + ".weight", is_pred = name.compare(0, pred.size(), pred) == 0;
            const int co2 = (co == 16) ? 32 : co;
            const int ci2 = is_head ? ci : (ci == 16 ? 32 : (ci == 32 ? 64 : ci));
            if (co2 == co && ci2 == ci) continue;
            (void)is_pred;
            m->pad_store.emplace_back((size_t)co2 * ci2 * kk, 0.f);
            float* dst = m->pad_store.back().data();
            for (int o = 0; o < co; ++o)
                for (int i = 0; i < ci; ++i) {
                    const int i2 = (ci == 32 && !is_head) ? (i < 16 ? i : i + 16) : i;      // cat(x, h): h moves to 32..47
                    memcpy(dst + ((size_t)o * ci2 + i2) * kk, t.data + ((size_t)o * ci + i) * kk, kk * sizeof(float));
                }
            t.data = dst; t.shape[0] = co2; t.shape[1] = ci2;
        }
    }
}

int build_firenet(evr_model* m) {
    const evr_model_desc& d = m->desc;
    const bool legacy = d.arch == EVR_ARCH_FIRENET_LEGACY;
    int C = d.base_num_channels;
    EVR_REQUIRE(C % 16 == 0, "FireNet: base_num_channels %d unsupported", C);
    const std::string head = legacy ? "net.head.conv.conv2d" : "head.conv2d";
    const std::string g1 = legacy ? "net.head.recurrent_block" : "G1";
    const std::string r1 = legacy ? "net.resblocks.0.conv" : "R1";
    const std::string g2 = legacy ? "net.resblocks.0.recurrent_block" : "G2";
    const std::string r2 = legacy ? "net.resblocks.1" : "R2";
    const std::string pred = legacy ? "net.pred" : "pred";
    static const bool pad32 = getenv("EVR_FIRENET_PAD32") ? atoi(getenv("EVR_FIRENET_PAD32")) != 0 : false;
    m->fire_C = C;
    if (C == 16 && m->arith != 0 && pad32) {
        firenet_pad_state_dict(m, head, pred);
        C = m->fire_C = 32;
    }
    int rc;
    if ((rc = prep_head_pred(m, head, pred, false, C))) return rc;
    if ((rc = add_gru(m, "g1", g1, C))) return rc;
    if ((rc = add_conv(m, "r1.conv1", r1 + ".conv1.weight", r1 + ".conv1.bias", "", false, C, C, 3, 1, EPI_BIAS_RELU))) return rc;
    if ((rc = add_conv(m, "r1.conv2", r1 + ".conv2.weight", r1 + ".conv2.bias", "", false, C, C, 3, 1, EPI_RESIDUAL_RELU))) return rc;
    if ((rc = add_gru(m, "g2", g2, C))) return rc;
    if ((rc = add_conv(m, "r2.conv1", r2 + ".conv1.weight", r2 + ".conv1.bias", "", false, C, C, 3, 1, EPI_BIAS_RELU))) return rc;
    if ((rc = add_conv(m, "r2.conv2", r2 + ".conv2.weight", r2 + ".conv2.bias", "", false, C, C, 3, 1, EPI_RESIDUAL_RELU))) return rc;
    return EVR_OK;
}

// SPADE-E2VID = Unet6 (model/spade_e2v.py:113-179).  state_dict names are the module's own.
//   PixelShuffle(2) of conv0's 4C channels: conv channel oc = c*4 + (i*2 + j) lands at output pixel (2y+i, 2x+j),
//   channel c -- exactly a phase-major column-group GEMM (rows g*C + c, g = i*2 + j) whose epilogue writes the
//   interleaved pixels, so the shuffle is free.  SPADE's parameter-free BatchNorm (eval mode) acts per shuffled
//   channel c and folds into the four conv rows of c.
int add_shuffle_conv(evr_model* m, const std::string& name, const std::string& wname, const std::string& bn_prefix, int cin, int C) {
    Conv c; c.name = name;
    const HostTensor* w; int rc;
    if ((rc = find(m, wname, &w))) return rc;
    Affine a1;
    if ((rc = make_affine(m, "", bn_prefix, true, C, &a1, /*bn_affine=*/false))) return rc;
    Affine af; af.scale.resize(4 * C); af.shift.resize(4 * C);
    for (int oc = 0; oc < 4 * C; ++oc) { af.scale[oc] = a1.scale[oc / 4]; af.shift[oc] = a1.shift[oc / 4]; }
    c.kc = pick_kc(cin, 0);
    c.cin0 = cin; c.cin1 = 0; c.stride = 1; c.epi = EPI_BIAS; c.n_valid = C;
    if ((rc = prep_conv2d(c, w, af, 4 * C, cin, 3, 1, [C](int oc) { return (oc % 4) * C + oc / 4; }, 4 * C))) return rc;
    c.tp.ngroups = 4; c.tp.grp_cols = C;
    for (int g = 0; g < 4; ++g) { c.tp.grp_ofy[g] = g >> 1; c.tp.grp_ofx[g] = g & 1; }
    for (int t = 0; t < c.tp.ntaps; ++t) c.tp.tap_groups[t] = 0xF;
    c.transposed = true;              // x2 output grid, as the transposed convolutions (plan_conv)
    c.useful_taps = 9;                // every (tap, phase) block carries weights
    return finish_conv(m, c);
}

// two Conv2d(cin -> C, k3, bias) stacked into one GEMM: rows [0, C) = first, [C, 2C) = second (SPADE's gamma | beta)
int add_pair_conv(evr_model* m, const std::string& name, const std::string& p0, const std::string& p1, int cin, int C) {
    int rc;
    const HostTensor *w0, *w1;
    if ((rc = find(m, p0 + ".weight", &w0))) return rc;
    if ((rc = find(m, p1 + ".weight", &w1))) return rc;
    Affine a0, a1;
    if ((rc = make_affine(m, p0 + ".bias", "", false, C, &a0))) return rc;
    if ((rc = make_affine(m, p1 + ".bias", "", false, C, &a1))) return rc;
    Conv c; c.name = name; Conv tmp; tmp.name = name;
    c.kc = pick_kc(cin, 0); c.cin0 = cin; c.stride = 1; c.epi = EPI_BIAS; c.n_valid = 2 * C;
    if ((rc = prep_conv2d(c, w0, a0, C, cin, 3, 1, [](int co) { return co; }, 2 * C))) return rc;
    if ((rc = prep_conv2d(tmp, w1, a1, C, cin, 3, 1, [C](int co) { return C + co; }, 2 * C))) return rc;
    for (size_t i = (size_t)C * 9 * cin; i < (size_t)2 * C * 9 * cin; ++i) c.w[i] = tmp.w[i];
    for (int i = C; i < 2 * C; ++i) c.b[i] = tmp.b[i];
    return finish_conv(m, c);
}

int build_spade(evr_model* m) {
    const evr_model_desc& d = m->desc;
    EVR_REQUIRE(d.num_bins == 5 && d.base_num_channels == 32 && d.kernel_size == 5, "SPADE-E2VID is Unet6: 5 bins, 32 base channels, k5");
    int rc;
    // fc (head) -- prep_head_pred also wants a 1-channel pred; SPADE's 3-channel conv_img is prepared below
    {
        const HostTensor* w;
        if ((rc = find(m, "fc.weight", &w))) return rc;
        EVR_REQUIRE(w->ndim == 4 && w->shape[0] == 32 && w->shape[1] == 5 && w->shape[2] == 5 && w->shape[3] == 5, "fc weight shape mismatch");
        Affine af;
        if ((rc = make_affine(m, "fc.bias", "", false, 32, &af))) return rc;
        m->head_w.assign((size_t)5 * 25 * 32, 0.f); m->head_b.assign(32, 0.f);
        for (int co = 0; co < 32; ++co) {
            m->head_b[co] = (float)af.shift[co];
            for (int b = 0; b < 5; ++b) for (int t = 0; t < 25; ++t) m->head_w[((size_t)b * 25 + t) * 32 + co] = w->data[((size_t)co * 5 + b) * 25 + t];
        }
        if ((rc = upload(m->head_w, &m->d_head_w))) return rc;
        if ((rc = upload(m->head_b, &m->d_head_b))) return rc;
        if (m->arith != 0) {
            std::vector<unsigned> wf;
            m->head_wfrag_e = head_pack_wfrag(m->head_w.data(), 5, wf);
            EVR_HIP(hipMalloc((void**)&m->d_head_wfrag, wf.size() * sizeof(unsigned)));
            EVR_HIP(hipMemcpy(m->d_head_wfrag, wf.data(), wf.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        }
    }
    const int cin[3] = {32, 64, 128}, cout[3] = {64, 128, 256}, stride[3] = {1, 2, 2};
    for (int i = 0; i < 3; ++i) {
        const std::string p = "rec" + std::to_string(i);
        if ((rc = add_conv(m, p + ".conv", p + ".conv0.weight", "", p + ".bn", true, cin[i], cout[i], 5, stride[i], EPI_BIAS_RELU))) return rc;
        if ((rc = add_lstm(m, p + ".rec", p + ".recurrent_block", cout[i]))) return rc;
    }
    for (int i = 0; i < 2; ++i) {
        const std::string p = "res" + std::to_string(i);
        if ((rc = add_conv(m, p + ".conv1", p + ".conv1.weight", "", p + ".bn1", true, 256, 256, 3, 1, EPI_BIAS_RELU))) return rc;
        if ((rc = add_conv(m, p + ".conv2", p + ".conv2.weight", "", p + ".bn2", true, 256, 256, 3, 1, EPI_RESIDUAL_RELU))) return rc;
    }
    const int uc_in[2] = {256, 128}, uc_out[2] = {128, 64};
    for (int i = 0; i < 2; ++i) {
        const std::string p = "up" + std::to_string(i);
        if ((rc = add_shuffle_conv(m, p + ".conv", p + ".conv0.weight", p + ".norm.param_free_norm", uc_in[i], uc_out[i]))) return rc;
        // mlp_shared: Conv2d(3 -> 64, k3) + ReLU on the segmentation map (direct small-Cin kernel, [B*k*k][cout] weights)
        const HostTensor *w, *b;
        if ((rc = find(m, p + ".norm.mlp_shared.0.weight", &w))) return rc;
        if ((rc = find(m, p + ".norm.mlp_shared.0.bias", &b))) return rc;
        EVR_REQUIRE(w->ndim == 4 && w->shape[0] == 64 && w->shape[1] == 3 && w->shape[2] == 3 && b->numel() == 64, "'%s': mlp_shared shape mismatch", p.c_str());
        std::vector<float> sw((size_t)3 * 9 * 64), sb(b->data, b->data + 64);
        for (int co = 0; co < 64; ++co) for (int ci = 0; ci < 3; ++ci) for (int t = 0; t < 9; ++t) sw[((size_t)ci * 9 + t) * 64 + co] = w->data[((size_t)co * 3 + ci) * 9 + t];
        float *dw, *db;
        if ((rc = upload(sw, &dw))) return rc;
        if ((rc = upload(sb, &db))) return rc;
        m->sp_seg_dw.push_back(dw); m->sp_seg_db.push_back(db);
        if ((rc = add_pair_conv(m, p + ".gb", p + ".norm.mlp_gamma", p + ".norm.mlp_beta", 64, uc_out[i]))) return rc;
    }
    if ((rc = add_conv(m, "up2.conv", "up2.conv0.weight", "", "up2.bn", true, 64, 32, 5, 1, EPI_BIAS_RELU))) return rc;
    if ((rc = add_lstm(m, "up2.rec", "up2.recurrent_block", 32))) return rc;
    {   // conv_img (1x1, 32 -> 3, bias) + bn_img folded
        const HostTensor* w;
        if ((rc = find(m, "conv_img.weight", &w))) return rc;
        EVR_REQUIRE(w->numel() == 96, "conv_img weight has %lld elements, expected 96", (long long)w->numel());
        Affine ap;
        if ((rc = make_affine(m, "conv_img.bias", "bn_img", true, 3, &ap))) return rc;
        m->sp_pred_w.assign(96, 0.f);
        for (int k = 0; k < 3; ++k) {
            m->sp_pred_b[k] = (float)ap.shift[k];
            for (int ci = 0; ci < 32; ++ci) m->sp_pred_w[k * 32 + ci] = (float)((double)w->data[k * 32 + ci] * ap.scale[k]);
        }
        if ((rc = upload(m->sp_pred_w, &m->d_sp_pred_w))) return rc;
    }
    return EVR_OK;
}

// ET-Net = EITR / mls_tpa (model/eitr/u_trans.py:13-123).  Linear layers and the MultiheadAttention projections are
// 1x1 convolutions over the token tensor [n, L, 1, 256]; the patch embeddings split1 / split2 are k x k stride-k convs.
int add_linear(evr_model* m, const std::string& name, const std::string& wname, const std::string& bname, int row0, int nout, int nin, int epi) {
    const HostTensor *w, *b; int rc;
    if ((rc = find(m, wname, &w))) return rc;
    if ((rc = find(m, bname, &b))) return rc;
    EVR_REQUIRE(w->ndim == 2 && w->shape[1] == nin && w->shape[0] >= row0 + nout && b->numel() >= row0 + nout, "'%s': linear weight shape mismatch", wname.c_str());
    HostTensor wv; wv.data = w->data + (size_t)row0 * nin; wv.ndim = 4; wv.shape[0] = nout; wv.shape[1] = nin; wv.shape[2] = 1; wv.shape[3] = 1;
    Affine af; af.scale.assign(nout, 1.0); af.shift.resize(nout);
    for (int i = 0; i < nout; ++i) af.shift[i] = b->data[row0 + i];
    Conv c; c.name = name;
    c.kc = pick_kc(nin, 0); c.cin0 = nin; c.cin1 = 0; c.stride = 1; c.epi = epi; c.n_valid = nout;
    if ((rc = prep_conv2d(c, &wv, af, nout, nin, 1, 0, [](int co) { return co; }, round_up(nout, 32)))) return rc;
    return finish_conv(m, c);
}

int add_patch_conv(evr_model* m, const std::string& name, const std::string& prefix, int cin, int cout, int k) {
    const HostTensor* w; int rc;
    if ((rc = find(m, prefix + ".weight", &w))) return rc;
    Affine af;
    if ((rc = make_affine(m, prefix + ".bias", "", false, cout, &af))) return rc;
    Conv c; c.name = name;
    c.kc = pick_kc(cin, 0); c.cin0 = cin; c.cin1 = 0; c.stride = k; c.epi = EPI_BIAS; c.n_valid = cout;
    if ((rc = prep_conv2d(c, w, af, cout, cin, k, 0, [](int co) { return co; }, round_up(cout, 32)))) return rc;   // pad 0: taps (ky, kx)
    return finish_conv(m, c);
}

int add_layernorm(evr_model* m, const std::string& prefix, int* index) {
    const HostTensor *w, *b; int rc;
    if ((rc = find(m, prefix + ".weight", &w))) return rc;
    if ((rc = find(m, prefix + ".bias", &b))) return rc;
    EVR_REQUIRE(w->numel() == 256 && b->numel() == 256, "'%s': LayerNorm(256) expected", prefix.c_str());
    std::vector<float> hw(w->data, w->data + 256), hb(b->data, b->data + 256);
    float *dw, *db;
    if ((rc = upload(hw, &dw))) return rc;
    if ((rc = upload(hb, &db))) return rc;
    *index = (int)m->et_ln.size();
    m->et_ln.push_back({dw, db});
    return EVR_OK;
}

struct EtLayer { int ln1, ln2, ln21, ln22, ln3; };       // LayerNorm indices (encoder: ln1, ln2; decoder: ln1, ln21, ln22, ln3)

int build_etnet(evr_model* m) {
    const evr_model_desc& d = m->desc;
    const bool bn = d.norm == EVR_NORM_BN, inn = d.norm == EVR_NORM_IN;
    EVR_REQUIRE(d.base_num_channels == 32 && d.kernel_size == 5, "ET-Net: 32 base channels, k5");
    int rc;
    // every ConvLayer / RecurrentConvLayer / UpsampleConvLayer of mls_tpa takes `norm` (u_trans.py:16-52): BatchNorm in eval mode,
    // or InstanceNorm2d(track_running_stats=True) = a fixed affine from the running statistics -- both folded (see build_unet)
    if ((rc = prep_head_pred(m, "head.conv2d", "pred", bn, 32, inn, (bn || inn) ? "head.norm_layer" : ""))) return rc;
    for (int i = 0; i < 3; ++i) {
        const int cin = 32 << i, cout = 64 << i;
        const std::string p = "DownsampleConv." + std::to_string(i);
        if ((rc = add_conv(m, "enc" + std::to_string(i) + ".conv", p + ".conv.conv2d.weight", p + ".conv.conv2d.bias", p + ".conv.norm_layer", bn || inn, cin, cout, 5, 2, EPI_BIAS_RELU, inn, !inn))) return rc;
        if ((rc = add_lstm(m, "enc" + std::to_string(i) + ".rec", p + ".recurrent_block", cout))) return rc;
    }
    if ((rc = add_patch_conv(m, "split1", "split1", 128, 256, 2))) return rc;
    if ((rc = add_patch_conv(m, "split2", "split2", 64, 256, 4))) return rc;
    int dummy;
    for (int s = 0; s < 3; ++s) {
        for (int l = 0; l < 3; ++l) {
            const std::string p = "trans_encoder" + std::to_string(s) + ".encoder.layers." + std::to_string(l), n = "te" + std::to_string(s) + "." + std::to_string(l);
            if ((rc = add_layernorm(m, p + ".norm1", &dummy))) return rc;
            if ((rc = add_layernorm(m, p + ".norm2", &dummy))) return rc;
            if ((rc = add_linear(m, n + ".qkv", p + ".self_attn.in_proj_weight", p + ".self_attn.in_proj_bias", 0, 768, 256, EPI_BIAS))) return rc;
            if ((rc = add_linear(m, n + ".out", p + ".self_attn.out_proj.weight", p + ".self_attn.out_proj.bias", 0, 256, 256, EPI_BIAS))) return rc;
            if ((rc = add_linear(m, n + ".ff1", p + ".linear1.weight", p + ".linear1.bias", 0, 1024, 256, EPI_BIAS_RELU))) return rc;
            if ((rc = add_linear(m, n + ".ff2", p + ".linear2.weight", p + ".linear2.bias", 0, 256, 1024, EPI_BIAS))) return rc;
        }
        for (int l = 0; l < 2; ++l) {
            const std::string p = "trans_decoder" + std::to_string(s) + ".decoder.layers." + std::to_string(l), n = "td" + std::to_string(s) + "." + std::to_string(l);
            if ((rc = add_layernorm(m, p + ".norm1", &dummy))) return rc;
            if ((rc = add_layernorm(m, p + ".norm21", &dummy))) return rc;
            if ((rc = add_layernorm(m, p + ".norm22", &dummy))) return rc;
            if ((rc = add_layernorm(m, p + ".norm3", &dummy))) return rc;
            if ((rc = add_linear(m, n + ".qkv", p + ".self_attn.in_proj_weight", p + ".self_attn.in_proj_bias", 0, 768, 256, EPI_BIAS))) return rc;
            if ((rc = add_linear(m, n + ".out", p + ".self_attn.out_proj.weight", p + ".self_attn.out_proj.bias", 0, 256, 256, EPI_BIAS))) return rc;
            if ((rc = add_linear(m, n + ".cq", p + ".cross_attn.in_proj_weight", p + ".cross_attn.in_proj_bias", 0, 256, 256, EPI_BIAS))) return rc;
            if ((rc = add_linear(m, n + ".ckv", p + ".cross_attn.in_proj_weight", p + ".cross_attn.in_proj_bias", 256, 512, 256, EPI_BIAS))) return rc;
            if ((rc = add_linear(m, n + ".cout", p + ".cross_attn.out_proj.weight", p + ".cross_attn.out_proj.bias", 0, 256, 256, EPI_BIAS))) return rc;
            if ((rc = add_linear(m, n + ".ff1", p + ".linear1.weight", p + ".linear1.bias", 0, 1024, 256, EPI_BIAS_RELU))) return rc;
            if ((rc = add_linear(m, n + ".ff2", p + ".linear2.weight", p + ".linear2.bias", 0, 256, 1024, EPI_BIAS))) return rc;
        }
    }
    const int uin[3] = {256, 128, 64}, uout[3] = {128, 64, 32};
    for (int i = 0; i < 3; ++i) {
        const std::string p = "UpsampleConv." + std::to_string(i);
        if ((rc = add_conv(m, "dec" + std::to_string(i), p + ".conv2d.weight", p + ".conv2d.bias", p + ".norm_layer", bn || inn, uin[i], uout[i], 5, 1, EPI_BIAS_RELU, inn, !inn))) return rc;
    }
    return EVR_OK;
}

// ------------------------------------------------------------------------------------------------
// shape-dependent planning
int alloc(evr_model* m, DevTensor* t, int n, int h, int w, int c, hipStream_t stream, bool packed = false) {
    t->n = n; t->h = h; t->w = w; t->c = c; t->packed = packed;
    EVR_REQUIRE(!packed || c % 16 == 0, "PACKED activation tensor with %d channels", c);
    EVR_HIP(hipMalloc((void**)&t->p, (size_t)t->numel() * sizeof(float) + 256));
    m->allocs.push_back({t->p, (size_t)t->numel() * sizeof(float)});
    EVR_HIP(hipMemsetAsync(t->p, 0, (size_t)t->numel() * sizeof(float), stream));
    EVR_REQUIRE(t->numel() * 4 < 0xFFFFFF00LL, "activation tensor of %lld elements exceeds the 4 GiB buffer-descriptor range; lower n_seq", (long long)t->numel());
    return EVR_OK;
}

// fills both parities of a conv's launch plan; in0/in1/out/... given per parity
struct ConvIO {
    const float* in0[2]; const float* in1[2]; float* out[2];
    const float* residual[2] = {nullptr, nullptr}; const float* post_add[2] = {nullptr, nullptr};
    float* state[2] = {nullptr, nullptr}; float* aux0[2] = {nullptr, nullptr};
    bool in_packed = false, out_packed = false, res_packed = false, padd_packed = false, state_packed = false;
};

void plan_conv(evr_model* m, int ci, int n, int hin, int win, const ConvIO& io, int cout_total) {
    Conv& c = m->convs[ci];
    for (int p = 0; p < 2; ++p) {
        ConvArgs& a = c.args[p];
        memset(&a, 0, sizeof(a));
        a.in0 = io.in0[p]; a.in1 = io.in1[p];
        a.c0 = c.cin0; a.c1 = c.cin1; a.in_mode = c.cin1 ? IN_CAT : IN_SINGLE;
        a.n = n; a.hin = hin; a.win = win;
        if (c.transposed) { a.hm = hin; a.wm = win; a.stride = 1; a.os = 2; a.hout = 2 * hin; a.wout = 2 * win; }
        else { a.hm = hin / c.stride; a.wm = win / c.stride; a.stride = c.stride; a.os = 1; a.hout = a.hm; a.wout = a.wm; }
        a.tp = c.tp;
        a.wgt = c.d_w; a.bias = c.d_b; a.cout = c.n_gemm; a.n_valid = c.n_valid;
        a.out = io.out[p]; a.cout_total = cout_total;
        a.epi = c.epi; a.residual = io.residual[p]; a.post_add = io.post_add[p];
        a.state = io.state[p]; a.aux0 = io.aux0[p]; a.hidden = c.hidden; a.x3 = c.x3;
        a.acc_scale = (c.x3 == 3) ? std::ldexp(1.0f, -(c.mx_e + H2_ACT_EXP)) : (c.x3 == 4 ? std::ldexp(1.0f, -c.mx_e) : 1.0f);
        a.mx_sa = 127 - MX_LO_EXP; a.mx_sb = 127 - c.mx_e; a.group_store = use_group_store(); set_fastdiv(a);
        a.wgt2 = c.d_w2; a.prog = c.d_prog; a.prog_steps = c.prog_steps;
        a.wgt_wino = c.d_wino; a.wino_s2d = c.wino_s2d; set_wino_grid(a);
        a.sat = m->d_sat ? m->d_sat + ci : nullptr;
        a.in_packed = io.in_packed; a.out_packed = io.out_packed; a.res_packed = io.res_packed;
        a.padd_packed = io.padd_packed; a.state_packed = io.state_packed;
        if (const char* e = getenv("EVR_ABLATE")) a.debug_ablate = (c.epi == EPI_LSTM || getenv("EVR_ABLATE_ALL")) ? atoi(e) : 0;
    }
    pick_conv_tile(c.args[0], c.kc, &c.wm, &c.nb);
    // direct-conv FLOPs: a transposed conv counts its k*k taps once per INPUT pixel (= k*k/4 per output pixel x 4 phases)
    const double taps = c.transposed ? c.useful_taps * 4.0 : c.useful_taps;
    c.flops = 2.0 * (double)n * c.args[0].hm * c.args[0].wm * taps * (c.cin0 + c.cin1) * c.n_valid;
    m->flops += c.flops;
}

void push_conv(evr_model* m, int ci) { Step s; s.kind = ST_CONV; s.conv = ci; m->steps.push_back(s); }

void name2(evr_model* m, const std::string& name, const DevTensor& t0, const DevTensor& t1) { m->named[0][name] = t0; m->named[1][name] = t1; }

// Fold the 1x1 prediction conv (+ skip-sum with `skip`, final activation, centre crop) into conv `ci`'s epilogue
// when its GEMM has a single N tile; otherwise the standalone pred kernel runs.  reserved[0] bit 0 (debug) keeps
// the conv's own NHWC output for evr_model_read_tensor.
void try_fuse_pred(evr_model* m, int ci, const float* skip, bool skip_packed, float* skip_dot = nullptr) {
    Conv& c = m->convs[ci];
    m->pred_fused_conv = -1;
    m->head.pred_w = nullptr; m->head.pred_dot = nullptr;
    if (c.n_gemm != 32 * c.nb || c.n_gemm > 128) return;
    if (c.epi != EPI_BIAS_RELU && c.epi != EPI_RESIDUAL_RELU && c.epi != EPI_BIAS) return;
    // the skip is the head tensor and the matrix-core head kernel runs: it also writes sum_c pred_w[c] * head[c] per pixel,
    // and the decoder's epilogue adds that one float instead of loading and unpacking the 32 skip channels
    const bool by_dot = skip_dot && m->head.wfrag && m->head.cout == 32 && getenv("EVR_NO_PRED_DOT") == nullptr;
    if (by_dot) { m->head.pred_w = m->d_pred_w; m->head.pred_dot = skip_dot; }
    for (int p = 0; p < 2; ++p) {
        ConvArgs& a = c.args[p];
        a.post_add = by_dot ? nullptr : skip; a.padd_packed = skip_packed ? 1 : 0;
        a.pred_skip_dot = by_dot ? skip_dot : nullptr;
        a.pred_w = m->d_pred_w; a.pred_b = m->pred_b; a.pred_sigmoid = m->desc.final_activation == EVR_ACT_SIGMOID;
        a.crop_h = m->H; a.crop_w = m->W; a.crop_y0 = m->iy0; a.crop_x0 = m->ix0;
        a.prev_rec = m->prev_rec;
        if (!(m->desc.reserved[0] & 1)) a.out = nullptr;
    }
    m->pred_fused_conv = ci;
}

int plan_unet(evr_model* m, hipStream_t stream) {
    const evr_model_desc& d = m->desc;
    const int E = d.num_encoders, base = d.base_num_channels, n = m->n_seq;
    const bool lstm = d.recurrent_block == EVR_REC_CONVLSTM;
    int rc;
    const int P = m->packed ? m->fmt : 0;   // every tensor that feeds a matrix-core convolution is PACKED (1) / H2 (2) / P6 (3) in the split modes
    DevTensor head;
    if ((rc = alloc(m, &head, n, m->hp, m->wp, base, stream, P))) return rc;
    name2(m, "head", head, head);
    m->head.out = head.p; m->head.out_packed = P;
    m->head.wfrag = P ? m->d_head_wfrag : nullptr;      // matrix-core head conv in the split modes

    // x[p]: current activation pointer per parity
    const float* x[2] = {head.p, head.p};
    int h = m->hp, w = m->wp;
    std::vector<DevTensor> blk0(E), blk1(E);   // encoder outputs (skip connections) per parity
    for (int i = 0; i < E; ++i) {
        const int cout = base << (i + 1);
        const std::string en = "enc" + std::to_string(i);
        DevTensor cv;
        if ((rc = alloc(m, &cv, n, h / 2, w / 2, cout, stream, P))) return rc;
        ConvIO io{};
        io.in_packed = P; io.out_packed = P;
        io.in0[0] = x[0]; io.in0[1] = x[1]; io.in1[0] = io.in1[1] = nullptr; io.out[0] = io.out[1] = cv.p;
        const int ci = conv_index(m, en + ".conv");
        plan_conv(m, ci, n, h, w, io, cout);
        push_conv(m, ci);
        name2(m, en + ".conv", cv, cv);
        h /= 2; w /= 2;
        if (lstm) {
            DevTensor hb[2], cb;
            if ((rc = alloc(m, &hb[0], n, h, w, cout, stream, P))) return rc;
            if ((rc = alloc(m, &hb[1], n, h, w, cout, stream, P))) return rc;
            if ((rc = alloc(m, &cb, n, h, w, cout, stream))) return rc;      // cell state: fp32, never a GEMM operand
            ConvIO r{};
            r.in_packed = P; r.out_packed = P;
            for (int p = 0; p < 2; ++p) { r.in0[p] = cv.p; r.in1[p] = hb[p].p; r.out[p] = hb[1 - p].p; r.state[p] = cb.p; }
            const int ri = conv_index(m, en + ".rec");
            plan_conv(m, ri, n, h, w, r, cout);
            push_conv(m, ri);
            x[0] = hb[1].p; x[1] = hb[0].p;
            blk0[i] = hb[1]; blk1[i] = hb[0];
            name2(m, "h" + std::to_string(i), hb[1], hb[0]);
            name2(m, "c" + std::to_string(i), cb, cb);
        } else {
            DevTensor hs, z, hr;
            if ((rc = alloc(m, &hs, n, h, w, cout, stream, P))) return rc;
            if ((rc = alloc(m, &z, n, h, w, cout, stream))) return rc;
            if ((rc = alloc(m, &hr, n, h, w, cout, stream, P))) return rc;
            ConvIO a{}, b{};
            a.in_packed = b.in_packed = P; a.out_packed = b.out_packed = P; a.state_packed = b.state_packed = P;
            for (int p = 0; p < 2; ++p) {
                a.in0[p] = cv.p; a.in1[p] = hs.p; a.out[p] = hr.p; a.state[p] = hs.p; a.aux0[p] = z.p;
                b.in0[p] = cv.p; b.in1[p] = hr.p; b.out[p] = hs.p; b.state[p] = hs.p; b.aux0[p] = z.p;
            }
            const int zi = conv_index(m, en + ".rec.zr"), oi = conv_index(m, en + ".rec.out");
            plan_conv(m, zi, n, h, w, a, cout); push_conv(m, zi);
            plan_conv(m, oi, n, h, w, b, cout); push_conv(m, oi);
            x[0] = x[1] = hs.p;
            blk0[i] = blk1[i] = hs;
            name2(m, "h" + std::to_string(i), hs, hs);
        }
    }
    const int cm = base << E;
    // where the first decoder's skip-sum can be fused: into the last plain conv before it
    const bool fuse = !d.use_upsample_conv;
    int last_plain = -1;   // conv index whose epilogue may take post_add
    const bool inn = d.norm == EVR_NORM_IN;
    int last_inorm = -1;   // step index of the last ST_INORM (takes the first decoder's fused skip instead of a conv epilogue)
    for (int i = 0; i < d.num_residual_blocks; ++i) {
        const std::string rn = "res" + std::to_string(i);
        DevTensor t, o;
        if ((rc = alloc(m, &t, n, h, w, cm, stream, P))) return rc;
        if ((rc = alloc(m, &o, n, h, w, cm, stream, P))) return rc;
        const int c1 = conv_index(m, rn + ".conv1"), c2 = conv_index(m, rn + ".conv2");
        if (inn) {
            // conv (bias) -> InstanceNorm2d -> relu -> conv (bias) -> InstanceNorm2d -> + x -> relu  (submodules.py:169-184)
            DevTensor raw;
            if ((rc = alloc(m, &raw, n, h, w, cm, stream))) return rc;                 // PLAIN: the norm kernel's input
            ConvIO a{}, b{};
            a.in_packed = b.in_packed = P;
            for (int p = 0; p < 2; ++p) { a.in0[p] = x[p]; a.out[p] = raw.p; b.in0[p] = t.p; b.out[p] = raw.p; }
            plan_conv(m, c1, n, h, w, a, cm); push_conv(m, c1);
            { Step s; s.kind = ST_INORM; s.a[0] = s.a[1] = raw.p; s.out = t.p; s.h = h; s.w = w; s.c = cm; s.out_packed = P; m->steps.push_back(s); }
            plan_conv(m, c2, n, h, w, b, cm); push_conv(m, c2);
            { Step s; s.kind = ST_INORM; s.a[0] = s.a[1] = raw.p; s.b[0] = x[0]; s.b[1] = x[1]; s.b_packed = P; s.out = o.p; s.h = h; s.w = w; s.c = cm;
              s.out_packed = P; m->steps.push_back(s); last_inorm = (int)m->steps.size() - 1; }
            x[0] = x[1] = o.p;
            name2(m, rn, o, o);
            last_plain = -1;
            continue;
        }
        ConvIO a{}, b{};
        a.in_packed = b.in_packed = P; a.out_packed = b.out_packed = P; b.res_packed = P;
        for (int p = 0; p < 2; ++p) {
            a.in0[p] = x[p]; a.out[p] = t.p;
            b.in0[p] = t.p; b.out[p] = o.p; b.residual[p] = x[p];
        }
        plan_conv(m, c1, n, h, w, a, cm); push_conv(m, c1);
        plan_conv(m, c2, n, h, w, b, cm); push_conv(m, c2);
        x[0] = x[1] = o.p;
        name2(m, rn, o, o);
        last_plain = c2;
    }
    (void)last_inorm;
    for (int i = 0; i < E; ++i) {
        const int cin = base << (E - i), cout = base << (E - i - 1);
        const DevTensor sk[2] = {blk0[E - 1 - i], blk1[E - 1 - i]};
        const std::string dn = "dec" + std::to_string(i);
        const int di = conv_index(m, dn);
        DevTensor o;
        if (i == 0 && m->dynamic) {
            DevTensor up, ctxp, ctxf, t1, coef, inter, prev;
            if ((rc = alloc(m, &up, n, 2 * h, 2 * w, cin, stream))) return rc;
            if ((rc = alloc(m, &ctxp, n, 1, (d.num_bins + 1) * (m->hp / 4), m->wp / 4, stream))) return rc;   // planar [n,B+1,hp/4,wp/4]
            if ((rc = alloc(m, &ctxf, n, m->hp / 4, m->wp / 4, 32, stream, P))) return rc;
            if ((rc = alloc(m, &t1, n, 2 * h, 2 * w, 64, stream, P))) return rc;
            if ((rc = alloc(m, &coef, n, 2 * h, 2 * w, 72, stream))) return rc;
            if ((rc = alloc(m, &inter, n, 2 * h, 2 * w, cin * 6, stream))) return rc;
            if ((rc = alloc(m, &prev, n, m->hp, m->wp, 1, stream))) return rc;
            m->prev_rec = prev.p;
            { Step s; s.kind = ST_UPSAMPLE; s.out = up.p; s.h = h; s.w = w; s.c = cin; s.a_packed = P; s.b_packed = P;
              for (int p = 0; p < 2; ++p) { s.a[p] = x[p]; s.b[p] = sk[p].p; }
              m->steps.push_back(s); }
            h *= 2; w *= 2;
            EVR_REQUIRE(h == m->hp / 4 && w == m->wp / 4, "dynamic decoder: context resolution mismatch");
            memset(&m->ctx, 0, sizeof(m->ctx));
            m->ctx.prev_rec = prev.p; m->ctx.n = n; m->ctx.B = d.num_bins; m->ctx.H = m->H; m->ctx.W = m->W; m->ctx.hp = m->hp; m->ctx.wp = m->wp;
            m->ctx.pad_top = m->pad_top; m->ctx.pad_left = m->pad_left; m->ctx.out = ctxp.p;
            { Step s; s.kind = ST_CTX; m->steps.push_back(s); }
            memset(&m->ctxconv, 0, sizeof(m->ctxconv));
            m->ctxconv.vox = ctxp.p; m->ctxconv.n = n; m->ctxconv.B = d.num_bins + 1; m->ctxconv.H = h; m->ctxconv.W = w; m->ctxconv.hp = h; m->ctxconv.wp = w;
            m->ctxconv.k = 3; m->ctxconv.cout = 32; m->ctxconv.wgt = m->d_ctx_w; m->ctxconv.bias = m->d_ctx_b; m->ctxconv.out = ctxf.p; m->ctxconv.relu = 0; m->ctxconv.out_packed = P;
            { Step s; s.kind = ST_CTXCONV; m->steps.push_back(s); }
            const int b1 = conv_index(m, dn + ".bn1"), b2 = conv_index(m, dn + ".bn2");
            ConvIO a1{}, a2{}, a3{};
            a1.in_packed = P; a1.out_packed = P; a2.in_packed = P;     // coeff stays PLAIN (VALU kernel); the filtered tensor is
            a3.in_packed = P; a3.out_packed = P;                       // converted in place for the matrix-core conv
            for (int p = 0; p < 2; ++p) { a1.in0[p] = ctxf.p; a1.out[p] = t1.p; a2.in0[p] = t1.p; a2.out[p] = coef.p; a3.in0[p] = inter.p; }
            plan_conv(m, b1, n, h, w, a1, 64); push_conv(m, b1);
            plan_conv(m, b2, n, h, w, a2, 72); push_conv(m, b2);
            { Step s; s.kind = ST_DYN; s.a[0] = s.a[1] = up.p; s.b[0] = s.b[1] = coef.p; s.out = inter.p; s.h = h; s.w = w; s.c = cin; s.out_packed = (cin <= 256) ? P : 0; m->steps.push_back(s); }
            // (the filter kernel writes its output in the 1x1 convolution's operand format itself when it can: c <= 256)
            if (P && cin > 256) { Step s; s.kind = ST_TOPACKED; s.out = inter.p; s.h = h; s.w = w; s.c = cin * 6; m->steps.push_back(s); }
            if ((rc = alloc(m, &o, n, h, w, cout, stream, P))) return rc;
            for (int p = 0; p < 2; ++p) a3.out[p] = o.p;
            plan_conv(m, di, n, h, w, a3, cout); push_conv(m, di);
            m->dyn_flops = 2.0 * n * h * w * (double)cin * 25 * 6;
            m->flops += 2.0 * n * h * w * (double)cin * 25 * 6 + 2.0 * n * h * w * 72.0 * 25 + 2.0 * n * h * w * 9.0 * (d.num_bins + 1) * 32;
            name2(m, "ctx", ctxf, ctxf); name2(m, "coeff", coef, coef);
            last_plain = -1;   // the next decoder adds its skip itself
        } else if (d.use_upsample_conv) {
            DevTensor up;
            if ((rc = alloc(m, &up, n, 2 * h, 2 * w, cin, stream, P))) return rc;
            Step s; s.kind = ST_UPSAMPLE; s.out = up.p; s.h = h; s.w = w; s.c = cin; s.a_packed = P; s.b_packed = P; s.out_packed = P;
            for (int p = 0; p < 2; ++p) { s.a[p] = x[p]; s.b[p] = sk[p].p; }
            m->steps.push_back(s);
            name2(m, dn + ".up", up, up);
            h *= 2; w *= 2;
            if ((rc = alloc(m, &o, n, h, w, cout, stream, P))) return rc;
            ConvIO a{};
            a.in_packed = P; a.out_packed = P;      // the upsample kernel writes the conv's operand format
            for (int p = 0; p < 2; ++p) { a.in0[p] = up.p; a.out[p] = o.p; }
            plan_conv(m, di, n, h, w, a, cout); push_conv(m, di);
        } else {
            const float* xin[2] = {x[0], x[1]};
            if (fuse && last_plain >= 0) {
                // skip_sum (model_util.py:4-5) folded into the producer's epilogue
                for (int p = 0; p < 2; ++p) { m->convs[last_plain].args[p].post_add = sk[p].p; m->convs[last_plain].args[p].padd_packed = P; }
            } else {
                DevTensor sum;
                if ((rc = alloc(m, &sum, n, h, w, cin, stream, P))) return rc;
                Step s; s.kind = ST_ADD; s.out = sum.p; s.h = h; s.w = w; s.c = cin; s.a_packed = s.b_packed = s.out_packed = P;
                for (int p = 0; p < 2; ++p) { s.a[p] = x[p]; s.b[p] = sk[p].p; }
                m->steps.push_back(s);
                xin[0] = xin[1] = sum.p;
            }
            if ((rc = alloc(m, &o, n, 2 * h, 2 * w, cout, stream, P))) return rc;
            ConvIO a{};
            a.in_packed = P; a.out_packed = P;
            for (int p = 0; p < 2; ++p) { a.in0[p] = xin[p]; a.out[p] = o.p; }
            plan_conv(m, di, n, h, w, a, cout); push_conv(m, di);
            h *= 2; w *= 2;
            last_plain = di;
        }
        x[0] = x[1] = o.p;
        name2(m, dn, o, o);
    }
    m->pred_x[0] = x[0]; m->pred_x[1] = x[1];
    m->pred_skip[0] = m->pred_skip[1] = head.p;
    m->pred_c = base; m->pred_x_packed = P; m->pred_skip_packed = P;
    DevTensor hdot;      // sum_c pred_w[c] * head[c] per pixel, written by the matrix-core head kernel (try_fuse_pred)
    if ((rc = alloc(m, &hdot, n, m->hp, m->wp, 1, stream))) return rc;
    try_fuse_pred(m, conv_index(m, "dec" + std::to_string(E - 1)), head.p, P, hdot.p);
    if (P == 3 && m->pred_fused_conv >= 0 && (d.reserved[0] & 1)) {
        // the debug copy of the last decoder's own output is written run by run (4 channels): PLAIN in the P6 mode
        const std::string dn = "dec" + std::to_string(E - 1);
        for (int p = 0; p < 2; ++p) { m->convs[m->pred_fused_conv].args[p].out_packed = 0; m->named[p][dn].packed = 0; }
    }
    EVR_REQUIRE(!m->dynamic || m->pred_fused_conv >= 0, "dynamic decoder: the prediction layer could not be fused (prev_recs needs it)");
    return EVR_OK;
}

int plan_firenet(evr_model* m, hipStream_t stream) {
    const int C = m->fire_C, Creal = m->desc.base_num_channels, n = m->n_seq, h = m->hp, w = m->wp;
    const int P = m->packed ? m->fmt : 0;      // (padded to 32 channels: every tensor between the convolutions is PACKED / H2)
    int rc;
    DevTensor x0, hs[2], z, hr, t, r[2];
    if ((rc = alloc(m, &x0, n, h, w, C, stream, P))) return rc;
    if ((rc = alloc(m, &hs[0], n, h, w, C, stream, P))) return rc;
    if ((rc = alloc(m, &hs[1], n, h, w, C, stream, P))) return rc;
    if ((rc = alloc(m, &z, n, h, w, C, stream))) return rc;             // the update gate stays fp32 (never a GEMM operand)
    if ((rc = alloc(m, &hr, n, h, w, C, stream, P))) return rc;
    if ((rc = alloc(m, &t, n, h, w, C, stream, P))) return rc;
    if ((rc = alloc(m, &r[0], n, h, w, C, stream, P))) return rc;
    if ((rc = alloc(m, &r[1], n, h, w, C, stream, P))) return rc;
    for (DevTensor* q : {&x0, &hs[0], &hs[1], &z, &hr, &t, &r[0], &r[1]}) q->c_valid = Creal;
    m->head.out = x0.p; m->head.out_packed = P; m->head.cout = C;
    name2(m, "head", x0, x0);
    const float* x = x0.p;
    const char* gn[2] = {"g1", "g2"};
    const char* rn[2] = {"r1", "r2"};
    const double real = (double)Creal * Creal / ((double)C * C);       // direct-conv FLOPs of the REAL channels
    auto plan = [&](int ci, const ConvIO& io) {
        const double before = m->flops;
        plan_conv(m, ci, n, h, w, io, C);
        m->convs[ci].flops *= real;
        m->flops = before + m->convs[ci].flops;
        push_conv(m, ci);
    };
    for (int s = 0; s < 2; ++s) {
        ConvIO a{}, b{};
        a.in_packed = b.in_packed = P; a.out_packed = b.out_packed = P; a.state_packed = b.state_packed = P;
        for (int p = 0; p < 2; ++p) {
            a.in0[p] = x; a.in1[p] = hs[s].p; a.out[p] = hr.p; a.state[p] = hs[s].p; a.aux0[p] = z.p;
            b.in0[p] = x; b.in1[p] = hr.p; b.out[p] = hs[s].p; b.state[p] = hs[s].p; b.aux0[p] = z.p;
        }
        plan(conv_index(m, std::string(gn[s]) + ".zr"), a);
        plan(conv_index(m, std::string(gn[s]) + ".out"), b);
        name2(m, "h" + std::to_string(s), hs[s], hs[s]);
        ConvIO c1{}, c2{};
        c1.in_packed = c2.in_packed = P; c1.out_packed = c2.out_packed = P; c2.res_packed = P;
        for (int p = 0; p < 2; ++p) {
            c1.in0[p] = hs[s].p; c1.out[p] = t.p;
            c2.in0[p] = t.p; c2.out[p] = r[s].p; c2.residual[p] = hs[s].p;
        }
        plan(conv_index(m, std::string(rn[s]) + ".conv1"), c1);
        plan(conv_index(m, std::string(rn[s]) + ".conv2"), c2);
        name2(m, "res" + std::to_string(s), r[s], r[s]);
        x = r[s].p;
    }
    m->pred_x[0] = m->pred_x[1] = x;
    m->pred_skip[0] = m->pred_skip[1] = nullptr;
    m->pred_c = C;
    m->pred_x_packed = P; m->pred_skip_packed = 0;
    try_fuse_pred(m, conv_index(m, "r2.conv2"), nullptr, false);
    return EVR_OK;
}

int plan_spade(evr_model* m, hipStream_t stream) {
    const int n = m->n_seq, hp = m->hp, wp = m->wp;
    const int P = m->packed ? m->fmt : 0;
    int rc;
    EVR_REQUIRE(hp % 4 == 0 && wp % 4 == 0, "SPADE-E2VID: padded size %dx%d not a multiple of 4", wp, hp);
    DevTensor xpad, xorg, xorgh, head;
    if ((rc = alloc(m, &xpad, n, m->desc.num_bins, hp, wp, stream))) return rc;          // planar [n,B,hp,wp]
    if ((rc = alloc(m, &xorg, n, 3, hp, wp, stream))) return rc;                          // planar [n,3,hp,wp]
    if ((rc = alloc(m, &xorgh, n, 3, hp / 2, wp / 2, stream))) return rc;
    if ((rc = alloc(m, &head, n, hp, wp, 32, stream, P))) return rc;
    m->sp_xpad = xpad.p; m->sp_xorg = xorg.p; m->sp_xorg_half = xorgh.p;
    name2(m, "head", head, head);
    m->head.out = head.p; m->head.out_packed = P;
    m->head.wfrag = P ? m->d_head_wfrag : nullptr;

    const float* x[2] = {head.p, head.p};
    int h = hp, w = wp;
    DevTensor hs[4][2];      // hidden states (new-after-parity-p at index [i][p])
    auto add_rec = [&](const std::string& nm, int idx, int cin_unused, int cout, int stride) -> int {
        (void)cin_unused;
        DevTensor cv;
        int r;
        if ((r = alloc(m, &cv, n, h / stride, w / stride, cout, stream, P))) return r;
        ConvIO io{};
        io.in_packed = P; io.out_packed = P;
        io.in0[0] = x[0]; io.in0[1] = x[1]; io.in1[0] = io.in1[1] = nullptr; io.out[0] = io.out[1] = cv.p;
        const int ci = conv_index(m, nm + ".conv");
        plan_conv(m, ci, n, h, w, io, cout); push_conv(m, ci);
        h /= stride; w /= stride;
        DevTensor hb[2], cb;
        if ((r = alloc(m, &hb[0], n, h, w, cout, stream, P))) return r;
        if ((r = alloc(m, &hb[1], n, h, w, cout, stream, P))) return r;
        if ((r = alloc(m, &cb, n, h, w, cout, stream))) return r;
        ConvIO q{};
        q.in_packed = P; q.out_packed = P;
        for (int p = 0; p < 2; ++p) { q.in0[p] = cv.p; q.in1[p] = hb[p].p; q.out[p] = hb[1 - p].p; q.state[p] = cb.p; }
        const int ri = conv_index(m, nm + ".rec");
        plan_conv(m, ri, n, h, w, q, cout); push_conv(m, ri);
        x[0] = hb[1].p; x[1] = hb[0].p;
        hs[idx][0] = hb[1]; hs[idx][1] = hb[0];
        name2(m, "h" + std::to_string(idx), hb[1], hb[0]);
        name2(m, "c" + std::to_string(idx), cb, cb);
        return EVR_OK;
    };
    if ((rc = add_rec("rec0", 0, 32, 64, 1))) return rc;
    if ((rc = add_rec("rec1", 1, 64, 128, 2))) return rc;
    if ((rc = add_rec("rec2", 2, 128, 256, 2))) return rc;
    int last = -1;
    for (int i = 0; i < 2; ++i) {
        const std::string rn = "res" + std::to_string(i);
        DevTensor t, o;
        if ((rc = alloc(m, &t, n, h, w, 256, stream, P))) return rc;
        if ((rc = alloc(m, &o, n, h, w, 256, stream, P))) return rc;
        ConvIO a{}, b{};
        a.in_packed = b.in_packed = P; a.out_packed = b.out_packed = P; b.res_packed = P;
        for (int p = 0; p < 2; ++p) { a.in0[p] = x[p]; a.out[p] = t.p; b.in0[p] = t.p; b.out[p] = o.p; b.residual[p] = x[p]; }
        const int c1 = conv_index(m, rn + ".conv1"), c2 = conv_index(m, rn + ".conv2");
        plan_conv(m, c1, n, h, w, a, 256); push_conv(m, c1);
        plan_conv(m, c2, n, h, w, b, 256); push_conv(m, c2);
        x[0] = x[1] = o.p;
        name2(m, rn, o, o);
        last = c2;
    }
    // up0(x + x2, x_org): the skip sum rides on res1.conv2's epilogue (model_util-style fusion)
    for (int p = 0; p < 2; ++p) { m->convs[last].args[p].post_add = hs[2][p].p; m->convs[last].args[p].padd_packed = P; }
    m->sp_seg.clear();
    for (int i = 0; i < 2; ++i) {
        const std::string un = "up" + std::to_string(i);
        const int C = (i == 0) ? 128 : 64;
        DevTensor xn, actv, gb, u;
        if ((rc = alloc(m, &xn, n, 2 * h, 2 * w, C, stream))) return rc;                   // PLAIN: read by spade_apply
        ConvIO a{};
        a.in_packed = P; a.out_packed = false;
        for (int p = 0; p < 2; ++p) { a.in0[p] = x[p]; a.out[p] = xn.p; }
        const int ci = conv_index(m, un + ".conv");
        plan_conv(m, ci, n, h, w, a, C); push_conv(m, ci);
        h *= 2; w *= 2;
        const float* seg = m->sp_xorg;
        if (h != hp) {
            EVR_REQUIRE(2 * h == hp && 2 * w == wp, "SPADE-E2VID: unexpected decoder resolution");
            Step s; s.kind = ST_SP_NEAREST; m->steps.push_back(s);
            seg = m->sp_xorg_half;
        }
        if ((rc = alloc(m, &actv, n, h, w, 64, stream, P))) return rc;
        HeadArgs ha; memset(&ha, 0, sizeof(ha));
        ha.vox = seg; ha.n = n; ha.B = 3; ha.H = h; ha.W = w; ha.hp = h; ha.wp = w; ha.k = 3; ha.cout = 64;
        ha.wgt = m->sp_seg_dw[i]; ha.bias = m->sp_seg_db[i]; ha.out = actv.p; ha.relu = 1; ha.out_packed = P;
        m->sp_seg.push_back(ha);
        { Step s; s.kind = ST_SP_SEG; s.conv = i; m->steps.push_back(s); }
        if ((rc = alloc(m, &gb, n, h, w, 2 * C, stream))) return rc;
        ConvIO g{};
        g.in_packed = P; g.out_packed = false;
        for (int p = 0; p < 2; ++p) { g.in0[p] = actv.p; g.out[p] = gb.p; }
        const int gi = conv_index(m, un + ".gb");
        plan_conv(m, gi, n, h, w, g, 2 * C); push_conv(m, gi);
        if ((rc = alloc(m, &u, n, h, w, C, stream, P))) return rc;
        { Step s; s.kind = ST_SP_APPLY; s.a[0] = s.a[1] = xn.p; s.b[0] = s.b[1] = gb.p; s.out = u.p; s.h = h; s.w = w; s.c = C;
          s.out_packed = P; s.a_packed = P;          // a_packed: format of the fused skip operand
          s.conv = i;                                // skip = hidden state of rec(1 - i)
          m->steps.push_back(s); }
        m->flops += 2.0 * n * h * w * 9.0 * 3 * 64;
        x[0] = x[1] = u.p;
        name2(m, un, u, u);
    }
    // skip operands of the two spade_apply steps (parity-dependent hidden states): kept in the model, looked up by step
    m->sp_skip[0][0] = hs[1][0].p; m->sp_skip[0][1] = hs[1][1].p;     // up0 output + x1
    m->sp_skip[1][0] = hs[0][0].p; m->sp_skip[1][1] = hs[0][1].p;     // up1 output + x0
    {   // up2: RecurrentConvLayer(64 -> 32, stride 1)
        DevTensor cv;
        if ((rc = alloc(m, &cv, n, h, w, 32, stream, P))) return rc;
        ConvIO io{};
        io.in_packed = P; io.out_packed = P;
        io.in0[0] = x[0]; io.in0[1] = x[1]; io.in1[0] = io.in1[1] = nullptr; io.out[0] = io.out[1] = cv.p;
        const int ci = conv_index(m, "up2.conv");
        plan_conv(m, ci, n, h, w, io, 32); push_conv(m, ci);
        DevTensor hb[2], cb;
        if ((rc = alloc(m, &hb[0], n, h, w, 32, stream, P))) return rc;
        if ((rc = alloc(m, &hb[1], n, h, w, 32, stream, P))) return rc;
        if ((rc = alloc(m, &cb, n, h, w, 32, stream))) return rc;
        ConvIO q{};
        q.in_packed = P; q.out_packed = P;
        for (int p = 0; p < 2; ++p) { q.in0[p] = cv.p; q.in1[p] = hb[p].p; q.out[p] = hb[1 - p].p; q.state[p] = cb.p; }
        const int ri = conv_index(m, "up2.rec");
        plan_conv(m, ri, n, h, w, q, 32); push_conv(m, ri);
        name2(m, "h3", hb[1], hb[0]); name2(m, "c3", cb, cb);
        m->pred_x[0] = hb[1].p; m->pred_x[1] = hb[0].p;
    }
    memset(&m->sp_pred, 0, sizeof(m->sp_pred));
    m->sp_pred.head = head.p; m->sp_pred.x_packed = P; m->sp_pred.head_packed = P;
    m->sp_pred.n = n; m->sp_pred.hp = hp; m->sp_pred.wp = wp; m->sp_pred.wgt = m->d_sp_pred_w;
    for (int k = 0; k < 3; ++k) m->sp_pred.bias[k] = m->sp_pred_b[k];
    m->sp_pred.prev = m->sp_xorg; m->sp_pred.H = m->H; m->sp_pred.W = m->W; m->sp_pred.iy0 = m->iy0; m->sp_pred.ix0 = m->ix0;
    m->pred_c = 32; m->pred_fused_conv = -1;
    m->flops += 2.0 * n * hp * wp * 32.0 * 2;     // conv_img has 3 outputs (the generic account adds one)
    return EVR_OK;
}

int plan_etnet(evr_model* m, hipStream_t stream) {
    const int n = m->n_seq, hp = m->hp, wp = m->wp;
    const int P = m->packed ? m->fmt : 0;
    int rc;
    EVR_REQUIRE(hp % 8 == 0 && wp % 8 == 0, "ET-Net: padded size %dx%d not a multiple of 8", wp, hp);
    DevTensor head;
    if ((rc = alloc(m, &head, n, hp, wp, 32, stream, P))) return rc;
    name2(m, "head", head, head);
    m->head.out = head.p; m->head.out_packed = P;
    m->head.wfrag = P ? m->d_head_wfrag : nullptr;
    const float* x[2] = {head.p, head.p};
    int h = hp, w = wp;
    DevTensor blk[3][2];
    for (int i = 0; i < 3; ++i) {       // DownsampleConv: conv k5 s2 + ConvLSTM (as the UNet encoders)
        const int cout = 64 << i;
        const std::string en = "enc" + std::to_string(i);
        DevTensor cv;
        if ((rc = alloc(m, &cv, n, h / 2, w / 2, cout, stream, P))) return rc;
        ConvIO io{};
        io.in_packed = P; io.out_packed = P;
        io.in0[0] = x[0]; io.in0[1] = x[1]; io.in1[0] = io.in1[1] = nullptr; io.out[0] = io.out[1] = cv.p;
        const int ci = conv_index(m, en + ".conv");
        plan_conv(m, ci, n, h, w, io, cout); push_conv(m, ci);
        h /= 2; w /= 2;
        DevTensor hb[2], cb;
        if ((rc = alloc(m, &hb[0], n, h, w, cout, stream, P))) return rc;
        if ((rc = alloc(m, &hb[1], n, h, w, cout, stream, P))) return rc;
        if ((rc = alloc(m, &cb, n, h, w, cout, stream))) return rc;
        ConvIO r{};
        r.in_packed = P; r.out_packed = P;
        for (int p = 0; p < 2; ++p) { r.in0[p] = cv.p; r.in1[p] = hb[p].p; r.out[p] = hb[1 - p].p; r.state[p] = cb.p; }
        const int ri = conv_index(m, en + ".rec");
        plan_conv(m, ri, n, h, w, r, cout); push_conv(m, ri);
        x[0] = hb[1].p; x[1] = hb[0].p;
        blk[i][0] = hb[1]; blk[i][1] = hb[0];
        name2(m, "h" + std::to_string(i), hb[1], hb[0]);
        name2(m, "c" + std::to_string(i), cb, cb);
    }
    const int th = hp / 8, tw = wp / 8, L = th * tw;

    // sine position table (position_encoding.py:14-23: float64 numpy, cast to float32)
    {
        std::vector<float> pos((size_t)L * 256);
        for (int l = 0; l < L; ++l)
            for (int j = 0; j < 256; ++j) {
                const double ang = (double)l / std::pow(10000.0, 2.0 * (j / 2) / 256.0);
                pos[(size_t)l * 256 + j] = (float)((j & 1) ? std::cos(ang) : std::sin(ang));
            }
        // (its own allocation, NOT alloc(): the same-shape path of evr_model_reset_states zeroes every entry of m->allocs,
        // and a zeroed table would silently drop the position term from the second sequence of a dataset on)
        float* pt = nullptr;
        EVR_HIP(hipMalloc((void**)&pt, pos.size() * sizeof(float)));
        m->shape_consts.push_back(pt);
        EVR_HIP(hipMemcpyAsync(pt, pos.data(), pos.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        EVR_HIP(hipStreamSynchronize(stream));       // `pos` is a local
        m->et_pos = pt;
    }
    auto tok = [&](DevTensor* t, int c, bool packed) { return alloc(m, t, n, L, 1, c, stream, packed); };
    DevTensor Wd[3], HS[3], HC[3], Ta, Tb, Tm, Tm2, XA, MEMN, QKV, CQ, CKV, AO, FF, SP;
    for (int s2 = 0; s2 < 3; ++s2) { if ((rc = tok(&Wd[s2], 256, false))) return rc; if ((rc = tok(&HS[s2], 256, false))) return rc; if ((rc = tok(&HC[s2], 256, false))) return rc; }
    if ((rc = tok(&Ta, 256, false))) return rc; if ((rc = tok(&Tb, 256, false))) return rc;
    if ((rc = tok(&Tm, 256, false))) return rc; if ((rc = tok(&Tm2, 256, false))) return rc;
    if ((rc = tok(&XA, 256, P))) return rc; if ((rc = tok(&MEMN, 256, P))) return rc;
    if ((rc = tok(&QKV, 768, false))) return rc; if ((rc = tok(&CQ, 256, false))) return rc; if ((rc = tok(&CKV, 512, false))) return rc;
    if ((rc = tok(&AO, 256, P))) return rc; if ((rc = tok(&FF, 1024, P))) return rc; if ((rc = tok(&SP, 256, false))) return rc;

    auto ln = [&](int idx, const float* in, float* out, int out_packed) {
        Step s; s.kind = ST_LN; s.conv = idx; s.a[0] = s.a[1] = in; s.out = out; s.out_packed = out_packed; m->steps.push_back(s);
    };
    // linear layer = 1x1 conv over [n, L, 1, C]: in (PACKED when P) -> out [+ post_add]
    auto lin = [&](const std::string& nm, const float* in, float* out, int cout_total, bool out_packed, const float* add) {
        const int ci = conv_index(m, nm);
        ConvIO io{};
        io.in_packed = P; io.out_packed = out_packed;
        for (int p = 0; p < 2; ++p) { io.in0[p] = in; io.in1[p] = nullptr; io.out[p] = out; io.post_add[p] = add; }
        io.padd_packed = false;
        plan_conv(m, ci, n, L, 1, io, cout_total); push_conv(m, ci);
    };
    auto attn = [&](const float* q, int ldq, int qo, const float* k, int ldk, int ko, const float* v, int ldv, int vo, float* out) {
        AttnArgs a; memset(&a, 0, sizeof(a));
        a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.qo = qo; a.ko = ko; a.vo = vo;
        a.n = n; a.Lq = L; a.Lk = L; a.heads = 8; a.out = out; a.out_packed = P;
        Step s; s.kind = ST_ATTN; s.conv = (int)m->et_attn.size(); m->et_attn.push_back(a); m->steps.push_back(s);
        m->flops += 2.0 * 2.0 * n * (double)L * L * 256;
    };
    for (int sc = 0; sc < 3; ++sc) {
        // ---- words + position embedding (u_trans.py:93-104) ----
        if (sc == 0) {
            Step s; s.kind = ST_ADDPOS; s.a[0] = blk[2][0].p; s.a[1] = blk[2][1].p; s.a_packed = P; s.out = Wd[0].p; m->steps.push_back(s);
        } else {
            const std::string nm = sc == 1 ? "split1" : "split2";
            const int src = 2 - sc, ks = sc == 1 ? 2 : 4;
            const int ci = conv_index(m, nm);
            ConvIO io{};
            io.in_packed = P; io.out_packed = false;
            for (int p = 0; p < 2; ++p) { io.in0[p] = blk[src][p].p; io.in1[p] = nullptr; io.out[p] = SP.p; }
            plan_conv(m, ci, n, th * ks, tw * ks, io, 256); push_conv(m, ci);
            Step s; s.kind = ST_ADDPOS; s.a[0] = s.a[1] = SP.p; s.a_packed = 0; s.out = Wd[sc].p; m->steps.push_back(s);
        }
        // ---- encoder: 3 pre-norm layers (transformer_encoder.py:64-76) ----
        const float* cur = Wd[sc].p;
        for (int l = 0; l < 3; ++l) {
            const std::string nm = "te" + std::to_string(sc) + "." + std::to_string(l);
            const int lb = sc * 14 + l * 2;
            float* outp = (l == 2) ? HS[sc].p : ((l & 1) ? Tb.p : Ta.p);
            ln(lb + 0, cur, XA.p, P);
            lin(nm + ".qkv", XA.p, QKV.p, 768, false, nullptr);
            attn(QKV.p, 768, 0, QKV.p, 768, 256, QKV.p, 768, 512, AO.p);
            lin(nm + ".out", AO.p, Tm.p, 256, false, cur);
            ln(lb + 1, Tm.p, XA.p, P);
            lin(nm + ".ff1", XA.p, FF.p, 1024, P, nullptr);
            lin(nm + ".ff2", FF.p, outp, 256, false, Tm.p);
            cur = outp;
        }
    }
    for (int sc = 0; sc < 3; ++sc) {
        // ---- decoder: tgt = hs[sc], memory = hs[0], hs[0], hs[1] (u_trans.py:106-108; transformer_decoder.py:66-84) ----
        const float* mem = HS[sc == 2 ? 1 : 0].p;
        const float* cur = HS[sc].p;
        for (int l = 0; l < 2; ++l) {
            const std::string nm = "td" + std::to_string(sc) + "." + std::to_string(l);
            const int lb = sc * 14 + 6 + l * 4;
            float* outp = (l == 1) ? HC[sc].p : Ta.p;
            ln(lb + 0, cur, XA.p, P);
            lin(nm + ".qkv", XA.p, QKV.p, 768, false, nullptr);
            attn(QKV.p, 768, 0, QKV.p, 768, 256, QKV.p, 768, 512, AO.p);
            lin(nm + ".out", AO.p, Tm.p, 256, false, cur);                 // tgt2
            ln(lb + 1, Tm.p, XA.p, P);
            ln(lb + 2, mem, MEMN.p, P);
            lin(nm + ".cq", XA.p, CQ.p, 256, false, nullptr);
            lin(nm + ".ckv", MEMN.p, CKV.p, 512, false, nullptr);
            attn(CQ.p, 256, 0, CKV.p, 512, 0, CKV.p, 512, 256, AO.p);
            lin(nm + ".cout", AO.p, Tm2.p, 256, false, Tm.p);              // tgt4
            ln(lb + 3, Tm2.p, XA.p, P);
            lin(nm + ".ff1", XA.p, FF.p, 1024, P, nullptr);
            lin(nm + ".ff2", FF.p, outp, 256, false, Tm2.p);
            cur = outp;
        }
    }
    // ---- hs_trans = mean of the six token sets -> [n, h/8, w/8, 256] (u_trans.py:112-113) ----
    DevTensor hm;
    if ((rc = alloc(m, &hm, n, th, tw, 256, stream, P))) return rc;
    m->et_mean_in[0] = HS[0].p; m->et_mean_in[1] = HS[1].p; m->et_mean_in[2] = HS[2].p;
    m->et_mean_in[3] = HC[0].p; m->et_mean_in[4] = HC[1].p; m->et_mean_in[5] = HC[2].p;
    { Step s; s.kind = ST_MEAN6; s.out = hm.p; s.out_packed = P; s.h = L; m->steps.push_back(s); }
    name2(m, "hs_trans", hm, hm);
    // ---- UpsampleConv decoders with skip sums (u_trans.py:116-117) ----
    const float* y[2] = {hm.p, hm.p};
    int ypk = P;
    h = th; w = tw;
    for (int i = 0; i < 3; ++i) {
        const int cin = 256 >> i, cout = 128 >> i;
        DevTensor up, o;
        if ((rc = alloc(m, &up, n, 2 * h, 2 * w, cin, stream, P))) return rc;
        Step s; s.kind = ST_UPSAMPLE; s.out = up.p; s.h = h; s.w = w; s.c = cin; s.a_packed = ypk; s.b_packed = P; s.out_packed = P;
        for (int p = 0; p < 2; ++p) { s.a[p] = y[p]; s.b[p] = blk[2 - i][p].p; }
        m->steps.push_back(s);
        h *= 2; w *= 2;
        if ((rc = alloc(m, &o, n, h, w, cout, stream, P))) return rc;
        ConvIO a{};
        a.in_packed = P; a.out_packed = P;
        for (int p = 0; p < 2; ++p) { a.in0[p] = up.p; a.out[p] = o.p; }
        const int di = conv_index(m, "dec" + std::to_string(i));
        plan_conv(m, di, n, h, w, a, cout); push_conv(m, di);
        y[0] = y[1] = o.p; ypk = P;
        name2(m, "dec" + std::to_string(i), o, o);
    }
    m->pred_x[0] = y[0]; m->pred_x[1] = y[1];
    m->pred_skip[0] = m->pred_skip[1] = head.p;
    m->pred_c = 32; m->pred_x_packed = P; m->pred_skip_packed = P;
    DevTensor hdot;
    if ((rc = alloc(m, &hdot, n, m->hp, m->wp, 1, stream))) return rc;
    try_fuse_pred(m, conv_index(m, "dec2"), head.p, P, hdot.p);
    return EVR_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" int evr_model_create(const evr_model_desc* desc, const evr_tensor* tensors, int n_tensors, evr_model** out) {
    EVR_REQUIRE(desc && out && (tensors || n_tensors == 0), "evr_model_create: null argument");
    *out = nullptr;
    evr_model* m = new evr_model();
    m->desc = *desc;
    for (int i = 0; i < n_tensors; ++i) {
        HostTensor t; t.data = tensors[i].data_host; t.ndim = tensors[i].ndim;
        if (!tensors[i].name || t.ndim < 0 || t.ndim > 4) { delete m; set_error("evr_model_create: bad tensor #%d", i); return EVR_ERR_INVALID; }
        for (int k = 0; k < t.ndim; ++k) t.shape[k] = tensors[i].shape[k];
        m->sd[tensors[i].name] = t;
    }
    // mx6 (f16 + MX-fp6, P6 tensors; the default): P6 groups carry their own scale and are written WHOLE -- by matrix-core epilogues
    // and by the bilinear upsampling kernel's group form.  That covers UNetRecurrent with ConvLSTM blocks, transposed-conv or
    // upsample-conv decoders, BatchNorm or no norm, and the 5-bin k5 32-channel head: the E2VID / E2VID+ / SSL-E2VID checkpoints'
    // layouts (BASELINE configurations 2 and 5).  Layouts with 4-channel producers of packed tensors (the dynamic decoder,
    // InstanceNorm, ConvGRU's epilogue, SPADE, ET-Net, FireNet) keep the f16 + MX-fp8 mode.
    m->arith = arith_mode();
    // per-model override (evr_model_desc::reserved[2] = mode + 1): the drop-in loop builds an exact-fp32 twin of a model whose
    // activations left the split format's range and re-runs the sequence on it (evreal_amd/eval.py), in the same process
    if (desc->reserved[2] != 0) {
        const int want = desc->reserved[2] - 1;
        if (want != 0 && want != 2 && want != 3 && want != 4) { delete m; set_error("evr_model_create: reserved[2] = %d is not an arithmetic mode + 1", desc->reserved[2]); return EVR_ERR_INVALID; }
        m->arith = want;
    }
    if (m->arith == 4) {
        const evr_model_desc& d = *desc;
        const bool ok = d.arch == EVR_ARCH_UNET_RECURRENT && d.recurrent_block == EVR_REC_CONVLSTM && !(d.reserved[1] & 1) &&
                        d.norm != EVR_NORM_IN && d.base_num_channels == 32 && d.kernel_size == 5 && d.num_bins == 5 && use_group_store();
        if (!ok) m->arith = 2;
    }
    int rc;
    if (desc->arch == EVR_ARCH_UNET_RECURRENT) rc = build_unet(m);
    else if (desc->arch == EVR_ARCH_FIRENET_LEGACY || desc->arch == EVR_ARCH_FIRENET) rc = build_firenet(m);
    else if (desc->arch == EVR_ARCH_SPADE_E2VID) rc = build_spade(m);
    else if (desc->arch == EVR_ARCH_ETNET) rc = build_etnet(m);
    else { set_error("evr_model_create: unknown arch %d", desc->arch); rc = EVR_ERR_UNSUPPORTED; }
    m->sd.clear();   // host pointers are only valid during this call
    if (rc) { delete m; return rc; }
    if (hipMalloc((void**)&m->d_sat, (m->convs.size() + 1) * sizeof(unsigned)) != hipSuccess ||
        hipMemset(m->d_sat, 0, (m->convs.size() + 1) * sizeof(unsigned)) != hipSuccess) { delete m; set_error("evr_model_create: counter allocation failed"); return EVR_ERR_HIP; }
    *out = m;
    return EVR_OK;
}

extern "C" int evr_model_destroy(evr_model* m) {
    delete m;
    return EVR_OK;
}

extern "C" int evr_model_release_shape(evr_model* m) {
    EVR_REQUIRE(m, "evr_model_release_shape: null model");
    m->release_shape();
    m->H = m->W = 0; m->frame = 0;
    return EVR_OK;
}

extern "C" int evr_model_reset_states(evr_model* m, int n_seq, int H, int W, evr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EVR_REQUIRE(m && n_seq >= 1 && H >= 1 && W >= 1, "evr_model_reset_states: bad arguments");
    if (m->n_seq == n_seq && m->H == H && m->W == W && !m->allocs.empty()) {
        // same shape: zero everything that carries state (all activations are cheap to clear too)
        for (auto& pr : m->allocs) EVR_HIP(hipMemsetAsync(pr.first, 0, pr.second, stream));
        m->frame = 0;
        return EVR_OK;
    }
    EVR_HIP(hipStreamSynchronize(stream));
    m->release_shape();
    m->n_seq = n_seq; m->H = H; m->W = W; m->frame = 0; m->flops = 0.0;
    m->packed = false; m->fmt = packed_fmt(m->arith);
    const bool firenet = m->desc.arch == EVR_ARCH_FIRENET_LEGACY || m->desc.arch == EVR_ARCH_FIRENET;
    const bool fire_padded = firenet && m->fire_C != m->desc.base_num_channels;
    {   // every tensor between the matrix-core convolutions is packed when ALL of them run a split arithmetic
        m->packed = !m->convs.empty();
        for (const auto& c : m->convs) if (!c.x3) m->packed = false;
        if (m->packed) m->fmt = packed_fmt(m->convs[0].x3);      // (FireNet's 16-channel layers: H2 whatever the global mode)
    }
    // CropParameters (utils/util.py:30-59)
    const int f = 1 << m->desc.pad_multiple_log2;
    m->hp = (H + f - 1) / f * f; m->wp = (W + f - 1) / f * f;
    m->pad_top = (m->hp - H + 1) / 2; m->pad_left = (m->wp - W + 1) / 2;     // ceil(0.5*(crop - size))
    m->iy0 = m->hp / 2 - H / 2; m->ix0 = m->wp / 2 - W / 2;                  // cy - floor(H/2)
    if (m->desc.arch == EVR_ARCH_UNET_RECURRENT)
        EVR_REQUIRE(m->hp % (1 << m->desc.num_encoders) == 0 && m->wp % (1 << m->desc.num_encoders) == 0,
                    "padded size %dx%d not divisible by 2^num_encoders", m->wp, m->hp);
    memset(&m->head, 0, sizeof(m->head));
    m->head.n = n_seq; m->head.B = m->desc.num_bins; m->head.H = H; m->head.W = W; m->head.hp = m->hp; m->head.wp = m->wp;
    m->head.pad_top = m->pad_top; m->head.pad_left = m->pad_left; m->head.k = m->desc.kernel_size;
    m->head.cout = m->desc.base_num_channels; m->head.wgt = m->d_head_w; m->head.bias = m->d_head_b; m->head.relu = 1;
    m->head.sat = m->d_sat ? m->d_sat + m->convs.size() : nullptr;
    m->head.wfrag_scale = std::ldexp(1.0f, -m->head_wfrag_e); m->head.wfrag_inv_scale = std::ldexp(1.0f, m->head_wfrag_e);
    const bool spade = m->desc.arch == EVR_ARCH_SPADE_E2VID;
    if (spade) {      // the head convolution reads the explicit padded copy (spade.hip): no padding of its own
        m->head.H = m->hp; m->head.W = m->wp; m->head.pad_top = 0; m->head.pad_left = 0;
    }
    int rc = (m->desc.arch == EVR_ARCH_UNET_RECURRENT) ? plan_unet(m, stream) : spade ? plan_spade(m, stream)
           : (m->desc.arch == EVR_ARCH_ETNET) ? plan_etnet(m, stream) : plan_firenet(m, stream);
    if (rc) { m->release_shape(); return rc; }
    m->flops += 2.0 * n_seq * m->hp * m->wp * (double)m->desc.num_bins * m->desc.kernel_size * m->desc.kernel_size * m->desc.base_num_channels;
    m->flops += 2.0 * n_seq * m->hp * m->wp * (double)(fire_padded ? m->desc.base_num_channels : m->pred_c);
    {   // split-K partial sums of this model's under-filled launches (conv.h KSPLIT_WS_BYTES): owned by the shape -- one device, one
        // stream at a time -- freed with it, never zeroed (every split launch writes what its epilogue kernel reads).  Only the split
        // arithmetics have kernels that split (the exact-fp32 twin never does), and only launches of <= 192 tiles of 128 pixels do;
        // a failed allocation degrades to "never split" (the launchers treat a null workspace that way) instead of failing the plan
        bool may_split = false;
        for (auto& c : m->convs)
            if (c.x3 && (int64_t)c.args[0].n * c.args[0].hm * c.args[0].wm <= 192LL * 128) may_split = true;
        float* kws = nullptr;
        if (may_split) {
            if (hipMalloc((void**)&kws, KSPLIT_WS_BYTES) != hipSuccess) { (void)hipGetLastError(); kws = nullptr; }
            else m->shape_consts.push_back(kws);
        }
        for (auto& c : m->convs) { c.args[0].ksplit_ws = kws; c.args[1].ksplit_ws = kws; }
    }
    // upload the launch plans (a failure here leaves no half-built shape behind: the next reset re-plans)
    std::vector<ConvArgs> all;
    for (auto& c : m->convs) { c.arg_slot = (int)all.size(); all.push_back(c.args[0]); all.push_back(c.args[1]); }
    hipError_t he = hipMalloc((void**)&m->d_args, all.size() * sizeof(ConvArgs));
    if (he == hipSuccess) he = hipMemcpyAsync(m->d_args, all.data(), all.size() * sizeof(ConvArgs), hipMemcpyHostToDevice, stream);
    if (he == hipSuccess) he = hipStreamSynchronize(stream);
    if (he != hipSuccess) { m->release_shape(); return hip_fail(he, "upload of the launch plans", __FILE__, __LINE__); }
    return EVR_OK;
}

extern "C" int evr_model_step(evr_model* m, const float* vox, const double* stats, float* img, unsigned flags, evr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EVR_REQUIRE(m && vox && img, "evr_model_step: null argument");
    EVR_REQUIRE(m->n_seq > 0, "evr_model_step: call evr_model_reset_states first");
    EVR_REQUIRE(!(flags & 1u) || stats, "evr_model_step: normalization requested without stats");
    const int p = (int)(m->frame & 1);
    int rc;
    HeadArgs ha = m->head;
    ha.vox = vox; ha.stats = (flags & 1u) ? stats : nullptr; ha.group_store = use_group_store();
    const bool spade = m->desc.arch == EVR_ARCH_SPADE_E2VID;
    if (spade) {
        // Unet6.forward (spade_e2v.py:139-151): cropper.pad made explicit; on a sequence's first frame the first three
        // channels of the padded input are min/max-normalised IN PLACE (the head conv sees them) and become x_org
        EVR_REQUIRE(!(flags & 1u), "SPADE-E2VID: event-tensor normalization is off in its method config (not fused here)");
        if ((rc = launch_spade_pad(vox, m->sp_xpad, m->n_seq, m->desc.num_bins, m->H, m->W, m->hp, m->wp, m->pad_top, m->pad_left, stream))) return rc;
        if (m->frame == 0 && (rc = launch_spade_first(m->sp_xpad, m->sp_xorg, m->n_seq, m->desc.num_bins, m->hp, m->wp, stream))) return rc;
        ha.vox = m->sp_xpad; ha.stats = nullptr;
    }
    // per-layer event timing (evr_model_profile_*): conv layers by name; every other launch under its step kind
    // (ids convs.size() + kind; "head" = convs.size() + 64, "pred" = + 65), selected by the same substring filter
    static const char* const kind_names[] = {"head", "conv", "upsample", "add", "pred", "ctx_down", "ctx_conv", "dynamic_filter", "to_packed",
                                             "spade_nearest", "spade_seg", "spade_apply", "instnorm", "layernorm", "attention", "add_pos", "mean6"};
    auto bracket_begin = [&](int id, const char* name, evr_model::ProfPair& pp) -> bool {
        if (!m->prof_on || std::string(name).find(m->prof_filter) == std::string::npos) return false;
        pp = evr_model::ProfPair{id, nullptr, nullptr};
        if (hipEventCreate(&pp.a) != hipSuccess || hipEventCreate(&pp.b) != hipSuccess) return false;
        (void)hipEventRecord(pp.a, stream);
        return true;
    };
    auto bracket_end = [&](evr_model::ProfPair& pp) { (void)hipEventRecord(pp.b, stream); m->prof_pending.push_back(pp); };
    const int nconv = (int)m->convs.size();
    {
        evr_model::ProfPair pp{};
        const bool on = bracket_begin(nconv + 64, "head", pp);
        if ((rc = launch_head_conv(ha, stream))) return rc;
        if (on) bracket_end(pp);
    }
    for (const Step& s : m->steps) {
        evr_model::ProfPair spp{};
        const bool son = (s.kind != ST_CONV) && bracket_begin(nconv + (int)s.kind, kind_names[(int)s.kind], spp);
        switch (s.kind) {
            case ST_CONV: {
                const Conv& c = m->convs[s.conv];
                const bool prof = m->prof_on && c.name.find(m->prof_filter) != std::string::npos;
                evr_model::ProfPair pp{s.conv, nullptr, nullptr};
                if (prof) {
                    EVR_HIP(hipEventCreate(&pp.a)); EVR_HIP(hipEventCreate(&pp.b));
                    EVR_HIP(hipEventRecord(pp.a, stream));
                }
                if ((rc = launch_conv_igemm(c.args[p], m->d_args + c.arg_slot + p, c.kc, c.wm, c.nb, stream, s.conv == m->pred_fused_conv ? img : nullptr))) return rc;
                if (prof) { EVR_HIP(hipEventRecord(pp.b, stream)); m->prof_pending.push_back(pp); }
                if (m->gate_event && c.name == m->gate_layer) EVR_HIP(hipEventRecord(m->gate_event, stream));
                break;
            }
            case ST_UPSAMPLE:
                if ((rc = launch_upsample2x_sum(s.a[p], s.b[p], s.out, m->n_seq, s.h, s.w, s.c, s.a_packed, s.b_packed, s.out_packed, stream))) return rc;
                break;
            case ST_ADD:
                if ((rc = launch_add(s.a[p], s.b[p], s.out, (int64_t)m->n_seq * s.h * s.w * s.c, s.out_packed, stream))) return rc;
                break;
            case ST_CTX: {
                CtxArgs ca = m->ctx;
                ca.vox = vox; ca.stats = (flags & 1u) ? stats : nullptr;
                if ((rc = launch_ctx_down(ca, stream))) return rc;
                break;
            }
            case ST_CTXCONV:
                if ((rc = launch_head_conv(m->ctxconv, stream))) return rc;
                break;
            case ST_TOPACKED:
                if ((rc = launch_to_packed(s.out, s.out, (int64_t)m->n_seq * s.h * s.w * s.c, stream, m->fmt))) return rc;
                break;
            case ST_DYN:
                if ((rc = launch_dynamic_filter(s.a[p], s.b[p], m->d_bases, s.out, m->n_seq, s.h, s.w, s.c, stream, s.out_packed))) return rc;
                break;
            case ST_INORM:
                if ((rc = launch_instnorm(s.a[p], s.b[p], nullptr, s.out, m->n_seq, s.h * s.w, s.c, s.b_packed, 0, s.out_packed, stream))) return rc;
                break;
            case ST_LN:
                if ((rc = launch_layernorm256(s.a[p], m->et_ln[s.conv].first, m->et_ln[s.conv].second, s.out, (int64_t)m->n_seq * (m->hp / 8) * (m->wp / 8), s.out_packed, stream))) return rc;
                break;
            case ST_ATTN:
                if ((rc = launch_attention(m->et_attn[s.conv], stream))) return rc;
                break;
            case ST_ADDPOS:
                if ((rc = launch_add_pos(s.a[p], m->et_pos, s.out, m->n_seq, (m->hp / 8) * (m->wp / 8), s.a_packed, stream))) return rc;
                break;
            case ST_MEAN6:
                if ((rc = launch_mean6(m->et_mean_in, s.out, (int64_t)m->n_seq * s.h, s.out_packed, stream))) return rc;
                break;
            case ST_SP_NEAREST:
                if ((rc = launch_nearest_half(m->sp_xorg, m->sp_xorg_half, m->n_seq * 3, m->hp, m->wp, stream))) return rc;
                break;
            case ST_SP_SEG:
                if ((rc = launch_head_conv(m->sp_seg[s.conv], stream))) return rc;
                break;
            case ST_SP_APPLY:
                if ((rc = launch_spade_apply(s.a[p], s.b[p], m->sp_skip[s.conv][p], s.out, (int64_t)m->n_seq * s.h * s.w, s.c, s.a_packed, s.out_packed, stream))) return rc;
                break;
            default: break;
        }
        if (son) bracket_end(spp);
    }
    PredArgs pa{};
    pa.x = m->pred_x[p]; pa.skip = m->pred_skip[p]; pa.n = m->n_seq; pa.hp = m->hp; pa.wp = m->wp; pa.c = m->pred_c;
    pa.wgt = m->d_pred_w; pa.bias = m->pred_b; pa.sigmoid = m->desc.final_activation == EVR_ACT_SIGMOID;
    pa.H = m->H; pa.W = m->W; pa.iy0 = m->iy0; pa.ix0 = m->ix0; pa.img = img;
    pa.x_packed = m->pred_x_packed; pa.skip_packed = m->pred_skip_packed;
    if (spade) {
        SpadePredArgs sa = m->sp_pred;
        sa.x = m->pred_x[p]; sa.img = img;
        if ((rc = launch_spade_pred(sa, stream))) return rc;
    } else if (m->pred_fused_conv < 0) {
        evr_model::ProfPair pp{};
        const bool on = bracket_begin(nconv + 65, "pred", pp);
        if ((rc = launch_pred(pa, stream))) return rc;
        if (on) bracket_end(pp);
    }
    m->frame++;
    return EVR_OK;
}

extern "C" int evr_model_read_tensor(evr_model* m, const char* name, float* dst, int64_t dst_elems, int64_t* n_out, evr_stream_t stream) {
    EVR_REQUIRE(m && name, "evr_model_read_tensor: null argument");
    EVR_REQUIRE(m->frame > 0, "evr_model_read_tensor: no frame has run yet");
    const int p = (int)((m->frame - 1) & 1);
    auto it = m->named[p].find(name);
    if (it == m->named[p].end()) { set_error("evr_model_read_tensor: unknown tensor '%s'", name); return EVR_ERR_INVALID; }
    const DevTensor& t = it->second;
    const int cv = t.c_valid ? t.c_valid : t.c;
    const int64_t numel = (int64_t)t.n * t.h * t.w * cv;
    if (n_out) *n_out = numel;
    if (!dst) return EVR_OK;
    EVR_REQUIRE(dst_elems >= numel, "evr_model_read_tensor: destination holds %lld elements, need %lld", (long long)dst_elems, (long long)numel);
    return launch_nhwc_to_nchw(t.p, dst, t.n, t.h, t.w, cv, t.packed ? m->fmt : 0, (hipStream_t)stream, t.c);
}

extern "C" int evr_model_set_gate(evr_model* m, const char* layer, evr_event_t ev) {
    EVR_REQUIRE(m != nullptr, "evr_model_set_gate: null model");
    if (!layer || !ev) { m->gate_layer.clear(); m->gate_event = nullptr; return EVR_OK; }
    EVR_REQUIRE(conv_index(m, layer) >= 0, "evr_model_set_gate: the model has no layer '%s'", layer);
    m->gate_layer = layer; m->gate_event = (hipEvent_t)ev;
    return EVR_OK;
}

extern "C" double evr_model_flops_per_step(const evr_model* m) { return m ? m->flops : 0.0; }

// arithmetic mode of the model's 32-channel-chunk convolutions (conv.h arith_mode): 0 fp32, 2 f16 + MX-fp8, 3 three f16 products,
// 4 f16 + MX-fp6 -- what EVR_ARITH asked for, narrowed to what the layout supports (evr_model_create)
// (FireNet's unpadded 16-channel layers run three f16 products on H2 tensors whatever the global mode -- finish_conv: when every
// convolution of the model agrees on a mode, THAT is the one reported)
extern "C" int evr_model_arith(const evr_model* m) {
    if (!m) return -1;
    int x3 = 0;
    for (const auto& c : m->convs) {
        if (!c.x3) continue;                      // (a VALU / fp32 side layer, e.g. a 1x1 prediction conv, does not decide the mode)
        if (x3 && c.x3 != x3) return m->arith;
        x3 = c.x3;
    }
    return x3 ? x3 : (m->convs.empty() ? m->arith : 0);
}

// Range guard of the packed activation formats (packed.h sat_note): how many output runs (4 or 16 channels of one pixel)
// of the matrix-core producers left the format's exact range since the counters were last cleared, and in which layer most.
// PACKED (default arithmetic) keeps only the f16 half of such values (2^-12 relative); H2 (EVR_ARITH=h3) CLAMPS them at
// +-4094.  Synchronises the stream.  Zero means the arithmetic's error analysis held for every frame so far.
extern "C" int evr_model_saturation(evr_model* m, int64_t* runs_host, char* worst_layer, size_t worst_len, int clear, evr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EVR_REQUIRE(m && runs_host, "evr_model_saturation: null argument");
    std::vector<unsigned> h(m->convs.size() + 1, 0u);
    EVR_HIP(hipMemcpyAsync(h.data(), m->d_sat, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    if (clear) EVR_HIP(hipMemsetAsync(m->d_sat, 0, h.size() * sizeof(unsigned), stream));
    EVR_HIP(hipStreamSynchronize(stream));
    int64_t total = 0; size_t worst = 0;
    for (size_t i = 0; i < h.size(); ++i) { total += h[i]; if (h[i] > h[worst]) worst = i; }
    *runs_host = total;
    if (worst_layer && worst_len) snprintf(worst_layer, worst_len, "%s", total == 0 ? "" : (worst < m->convs.size() ? m->convs[worst].name.c_str() : "head"));
    return EVR_OK;
}

// The same counters without a host synchronisation: an asynchronous copy on `stream` into caller-owned (pinned) host memory; the
// caller sums them once an event recorded behind this call has completed.  How the drop-in frame loop polls every chunk of frames
// before it books the chunk's scores and files, without stalling its two-deep pipeline (evreal_amd/eval.py).
extern "C" int evr_model_saturation_async(evr_model* m, unsigned* counters_host, int max_counters, int* n_counters, evr_stream_t stream_) {
    EVR_REQUIRE(m && n_counters, "evr_model_saturation_async: null argument");
    const int n = (int)m->convs.size() + 1;
    *n_counters = n;
    if (!counters_host) return EVR_OK;           // (size query)
    EVR_REQUIRE(max_counters >= n, "evr_model_saturation_async: %d counters, room for %d", n, max_counters);
    EVR_HIP(hipMemcpyAsync(counters_host, m->d_sat, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    return EVR_OK;
}

extern "C" int evr_model_profile_enable(evr_model* m, const char* filter) {
    EVR_REQUIRE(m != nullptr, "evr_model_profile_enable: null model");
    for (auto& pp : m->prof_pending) { (void)hipEventDestroy(pp.a); (void)hipEventDestroy(pp.b); }
    m->prof_pending.clear();
    m->prof_ms.assign(m->convs.size() + 80, 0.0);
    m->prof_n.assign(m->convs.size() + 80, 0);
    m->prof_on = filter != nullptr;
    m->prof_filter = filter ? filter : "";
    return EVR_OK;
}

extern "C" int evr_model_profile_read(evr_model* m, int max_layers, char* names, double* ms, double* flops_per_launch,
                                      int64_t* launches, int* n_layers, evr_stream_t stream) {
    EVR_REQUIRE(m && names && ms && flops_per_launch && launches && n_layers, "evr_model_profile_read: null argument");
    EVR_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (m->prof_ms.size() != m->convs.size() + 80) { m->prof_ms.assign(m->convs.size() + 80, 0.0); m->prof_n.assign(m->convs.size() + 80, 0); }
    for (auto& pp : m->prof_pending) {
        float t = 0.f;
        EVR_HIP(hipEventElapsedTime(&t, pp.a, pp.b));
        m->prof_ms[pp.conv] += t; m->prof_n[pp.conv] += 1;
        (void)hipEventDestroy(pp.a); (void)hipEventDestroy(pp.b);
    }
    m->prof_pending.clear();
    int k = 0;
    static const char* const kind_names[] = {"head", "conv", "upsample", "add", "pred", "ctx_down", "ctx_conv", "dynamic_filter", "to_packed",
                                             "spade_nearest", "spade_seg", "spade_apply", "instnorm", "layernorm", "attention", "add_pos", "mean6"};
    const size_t nconv = m->convs.size();
    for (size_t i = 0; i < m->prof_n.size() && k < max_layers; ++i) {
        if (m->prof_n[i] == 0) continue;
        const char* nm = i < nconv ? m->convs[i].name.c_str() : (i == nconv + 64 ? "head" : i == nconv + 65 ? "pred" : (i - nconv < 17 ? kind_names[i - nconv] : "?"));
        snprintf(names + (size_t)k * 64, 64, "%s", nm);
        double fl = 0.0;
        if (i < nconv) fl = m->convs[i].flops;
        else if (i == nconv + 64) fl = 2.0 * m->n_seq * m->hp * m->wp * (double)m->desc.num_bins * m->desc.kernel_size * m->desc.kernel_size * m->desc.base_num_channels;
        else if (i == nconv + (size_t)ST_DYN) fl = m->dyn_flops;
        else if (i == nconv + (size_t)ST_ATTN && !m->et_attn.empty()) {      // every attention launch: S = Q K^T and O = P V over 8 heads of 32
            const AttnArgs& a0 = m->et_attn[0];
            fl = 2.0 * 2.0 * a0.n * (double)a0.Lq * a0.Lk * 32.0 * a0.heads;
        }
        ms[k] = m->prof_ms[i]; flops_per_launch[k] = fl; launches[k] = m->prof_n[i];
        ++k;
    }
    *n_layers = k;
    return EVR_OK;
}

// ------------------------------------------------------------------------------------------------
// the PACKED activation codec on the host (include/evreal_hip.h): conv.h's pack_split_act, the twin of packed.h
extern "C" int evr_split_pack(const float* src, float* dst, int64_t n) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_split_pack: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    pack_split_act(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}

// the device codec (packed.h through to_packed_kernel): src/dst device pointers, dst may equal src
extern "C" int evr_split_pack_device(const float* src, float* dst, int64_t n, evr_stream_t stream) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_split_pack_device: n = %lld must be a multiple of 16", (long long)n);
    return launch_to_packed(src, dst, n, (hipStream_t)stream, 1);
}

extern "C" int evr_h2_pack_device(const float* src, float* dst, int64_t n, evr_stream_t stream) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_h2_pack_device: n = %lld must be a multiple of 16", (long long)n);
    return launch_to_packed(src, dst, n, (hipStream_t)stream, 2);
}

extern "C" int evr_p6_pack_device(const float* src, float* dst, int64_t n, evr_stream_t stream) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_p6_pack_device: n = %lld must be a multiple of 16", (long long)n);
    return launch_to_packed(src, dst, n, (hipStream_t)stream, 3);
}

extern "C" int evr_split_unpack(const float* src, float* dst, int64_t n) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_split_unpack: n = %lld must be a multiple of 16", (long long)n);
    unpack_split_act(src, dst, (size_t)n);
    return EVR_OK;
}
