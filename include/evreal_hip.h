/* evreal_hip.h -- C ABI of libevreal_hip.so, the MI355X (gfx950) hot path of EVREAL.
 *
 * The reference (ercanburak/EVREAL) is pure Python and has no FFI of its own; the boundary a
 * maintainer binds is its Python plugin surface.  Each entry point below replaces
 * one stretch of that surface and cites it (paths relative to the reference root).  The Python
 * host in evreal_amd/ binds these with ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - plain C types only; every pointer is CALLER-OWNED DEVICE memory unless suffixed _host;
 *   - every call takes a hipStream_t (passed as void*) and is asynchronous on it;
 *   - return 0 on success, a negative evr_status on failure; evr_last_error() gives a
 *     thread-local message; nothing throws across the boundary;
 *   - opaque handles own persistent device state and are freed by the matching _destroy;
 *     one handle is used from one host thread at a time;
 *   - tensors are dense fp32, layouts as in the reference (NCHW) at the boundary.
 */
#ifndef EVREAL_HIP_H
#define EVREAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* evr_stream_t; /* hipStream_t */
typedef void* evr_event_t;  /* hipEvent_t */

enum evr_status {
    EVR_OK = 0,
    EVR_ERR_INVALID = -1,   /* bad argument */
    EVR_ERR_HIP = -2,       /* a HIP runtime call failed */
    EVR_ERR_WORKSPACE = -3, /* workspace too small */
    EVR_ERR_UNSUPPORTED = -4,
    EVR_ERR_MISSING_TENSOR = -5
};

/* Message for the last failure on this thread ("" if none). */
const char* evr_last_error(void);
/* ABI version (major*1000 + minor).  1003 (round 6): evr_model_release_shape, evr_png_* (native PNG writer pool).  1002 (round 5): evr_model_desc.reserved[2] (per-model arithmetic), evr_model_saturation_async.
 * 1001 (round 4): evr_percentile_normalize rejects a NULL workspace (size it with
 * evr_percentile_normalize_workspace_bytes); evr_model_arith reports the mode the convolutions actually run (FireNet's 16-channel
 * layers: h3 whatever EVR_ARITH says). */
int evr_version(void);
/* Device facts used by bench.py's roofline block: CU count, clock (MHz), name. */
int evr_device_info(int device, int* n_cu, int* clock_mhz, char* name_out, size_t name_len);
/* A stream restricted to n_cus compute units spread evenly over the XCDs (from_top: the highest-numbered CUs of each XCD
 * instead of the lowest).  The reference is single-stream (eval.py:227: one synchronize per frame); evreal_amd runs the
 * evaluation half of a frame (eval.py:234-238) on such a stream beside the next frame's reconstruction. */
int evr_stream_create_cu_masked(int device, int n_cus, int from_top, evr_stream_t* out);
int evr_stream_destroy(evr_stream_t stream);
/* HIP events for cross-stream ordering INSIDE a model step (evr_model_set_gate below records one after a named layer);
 * the reference has a single stream and one synchronize per frame (eval.py:227). */
int evr_event_create(evr_event_t* out);
int evr_event_destroy(evr_event_t ev);
int evr_stream_wait_event(evr_stream_t stream, evr_event_t ev);

/* ----------------------------------------------------------------------------------------------
 * Events -> voxel grid.  Replaces utils/event_utils.py:27-59 (events_to_voxel_torch) and :4-24
 * (events_to_image_torch) as called from dataset.py:205-216 (MemMapDataset.get_voxel_grid), for
 * many windows per launch.
 *
 *   x, y, t, p     fp32 [n_events_total]: the four arrays MemMapDataset.__getitem__ builds
 *                  (dataset.py:48-57): x,y pixel coordinates, t = float32(ts - ts[window start]),
 *                  p = polarity weight (+-1).  Events are time-ordered inside a window.
 *   win_offsets    int64 [n_windows+1] (device): window w = events [win_offsets[w], win_offsets[w+1]);
 *                  an empty window yields zeros (dataset.py:200-203).
 *   out            fp32 [n_windows, B, H, W]; every cell is written (zero fill included).
 *   stats          optional (may be NULL): double [n_windows, 3] = {sum, sum of squares, nnz} of
 *                  each window's voxel grid, the reductions eval.py:402-405 needs.
 * Results are BIT-IDENTICAL to the reference's CPU path: per-cell adds happen in event order.
 * Events whose pixel falls outside [0,W)x[0,H) are dropped and counted (evr_voxelize_dropped).
 * workspace: caller-owned device memory of evr_voxelize_workspace_bytes(); its first 256 bytes must be ZERO the first
 * time it is used (hipMemset once after allocation) -- the calls themselves launch no memset.
 */
size_t evr_voxelize_workspace_bytes(int64_t n_events_total, int n_windows, int B, int H, int W);
int evr_voxelize(const float* x, const float* y, const float* t, const float* p,
                 const int64_t* win_offsets, int n_windows, int64_t n_events_total,
                 int B, int H, int W, float* out, double* stats,
                 void* workspace, size_t workspace_bytes, evr_stream_t stream);
/* Reads (synchronously) the out-of-range-event counter the last evr_voxelize on this workspace
 * left behind.  The reference raises from index_put_ in that case. */
int evr_voxelize_dropped(const void* workspace, int64_t* n_dropped_host, evr_stream_t stream);
/* The same counter accumulated over EVERY call since the workspace header was zeroed: a host loop polls it once per
 * sequence / batch instead of once per launch (evreal_amd raises when it is non-zero, like index_put_ would). */
int evr_voxelize_dropped_total(const void* workspace, int64_t* n_dropped_host, evr_stream_t stream);

/* Raw-format variant: events straight from the memmaps (dataset.py:222-228 + :53-57 fused):
 * xy int16 [n,2], ts float64 [n] (absolute seconds), pol uint8 {0,1}.  Same output. */
int evr_voxelize_raw(const int16_t* xy, const double* ts, const uint8_t* pol,
                     const int64_t* win_offsets, int n_windows, int64_t n_events_total,
                     int B, int H, int W, float* out, double* stats,
                     void* workspace, size_t workspace_bytes, evr_stream_t stream);

/* General window form for a whole resident sequence: window w = events [win_begin[w], win_end[w]) of the
 * stream; windows may overlap (k_events / t_seconds with a sliding window, dataset.py:104-130) or be empty.
 * rec_base[w] = exclusive prefix sum of the window lengths (where the window's records go in the
 * workspace); n_window_events = total of the window lengths (sizes the workspace). */
int evr_voxelize_raw_windows(const int16_t* xy, const double* ts, const uint8_t* pol,
                             const int64_t* win_begin, const int64_t* win_end, const int64_t* rec_base,
                             int n_windows, int64_t n_window_events, int B, int H, int W, float* out,
                             double* stats, void* workspace, size_t workspace_bytes, evr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Event-tensor normalization.  Replaces eval.py:398-410 (normalize_event_tensor), applied per
 * window: over non-zeros mean = sum/nnz, std = sqrt(sumsq/nnz - mean^2) clamped to 1e-6,
 * out = mask*(v-mean)/std; identity when nnz == 0.  In place on vox [n, B, H, W].
 * stats: optional double [n,3] from evr_voxelize (skips the reduction pass); NULL -> computed here
 * into workspace (needs n*3*8 bytes).
 */
int evr_event_tensor_normalize(float* vox, int n, int B, int H, int W, const double* stats,
                               void* workspace, size_t workspace_bytes, evr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Recurrent-network inference.  Replaces get_model_from_checkpoint_path/load_model
 * (eval.py:109-115,124-158), cropper.pad / model(voxel) / cropper.crop (eval.py:226-230,
 * utils/util.py:30-59) and model.reset_states() (eval.py:197) for
 *   EVR_ARCH_UNET_RECURRENT  model/model.py:108-144 E2VIDRecurrent over model/unet.py:85-143
 *                            (E2VID, E2VID+, SSL-E2VID layouts: BN or no norm, transposed or
 *                            bilinear-upsample decoders, ConvLSTM or ConvGRU, optional sigmoid)
 *   EVR_ARCH_FIRENET_LEGACY  model/legacy.py:32-111,155-187 (the "FireNet" method)
 *   EVR_ARCH_FIRENET         model/model.py:147-190 (the "FireNet+" method)
 *   EVR_ARCH_SPADE_E2VID     model/spade_e2v.py:113-179 (Unet6: full-resolution ConvLSTMs, pixel-shuffle decoders
 *                            with SPADE normalisation conditioned on the previous reconstruction, 3-channel head)
 *   EVR_ARCH_ETNET           model/eitr/ (EITR / mls_tpa): ConvLSTM encoder, three token scales through pre-norm
 *                            transformer encoders and decoders (8 heads, d = 256), bilinear-upsample decoders
 * Weights are handed over as the reference's own state_dict: names + host fp32 arrays.
 */
enum evr_arch { EVR_ARCH_UNET_RECURRENT = 0, EVR_ARCH_FIRENET_LEGACY = 1, EVR_ARCH_FIRENET = 2, EVR_ARCH_SPADE_E2VID = 3,
                EVR_ARCH_ETNET = 4 };
enum evr_norm { EVR_NORM_NONE = 0, EVR_NORM_BN = 1,
                EVR_NORM_IN = 2 /* submodules.py:22-23,160-162: running-statistics InstanceNorm in the conv layers (folded),
                                   true InstanceNorm2d inside the residual blocks */ };
enum evr_recurrent { EVR_REC_CONVLSTM = 0, EVR_REC_CONVGRU = 1 };
enum evr_activation { EVR_ACT_NONE = 0, EVR_ACT_SIGMOID = 1 };

typedef struct evr_model_desc {
    int arch;                /* evr_arch */
    int num_bins;            /* input channels */
    int base_num_channels;
    int num_encoders;        /* UNet only; also drives the crop/pad size (utils/util.py:41-48) */
    int num_residual_blocks;
    int kernel_size;         /* head/encoder/decoder kernel (5 for E2VID, 3 for FireNet) */
    int norm;                /* evr_norm */
    int use_upsample_conv;   /* 0: ConvTranspose2d decoders, 1: bilinear x2 + conv */
    int recurrent_block;     /* evr_recurrent */
    int final_activation;    /* evr_activation */
    int pad_multiple_log2;   /* cropper's num_encoders (FireNet legacy: 4, FireNet+: 0) */
    int reserved[5];         /* reserved[2]: arithmetic of THIS model's convolutions, as mode + 1 (1 exact fp32, 3 "mx", 4 "h3", 5 "mx6");
                              *   0 = what EVR_ARITH / EVR_FP32 select for the process.  The reference computes in fp32
                              *   (model/submodules.py:227-245): the exact-fp32 twin is what a sequence is re-run on when its
                              *   activations leave a split format's range (evr_model_saturation);
                              * reserved[1] bit 0: use_dynamic_decoder (HyperE2VID, model/submodules.py:100-127);
                              * reserved[0] bit 0: debug -- keep every intermediate readable by
                              * evr_model_read_tensor (otherwise the last decoder's NHWC output is never
                              * stored: the prediction layer is fused into its epilogue) */
} evr_model_desc;

typedef struct evr_tensor {
    const char* name;        /* state_dict key */
    const float* data_host;  /* fp32, contiguous, host memory */
    int ndim;
    int64_t shape[4];
} evr_tensor;

typedef struct evr_model evr_model;

int evr_model_create(const evr_model_desc* desc, const evr_tensor* tensors, int n_tensors,
                     evr_model** out);
int evr_model_destroy(evr_model* m);
/* (Re)allocate activations/state for n_seq sequences of H x W frames and zero the recurrent
 * state (model.reset_states(), eval.py:197). */
int evr_model_reset_states(evr_model* m, int n_seq, int H, int W, evr_stream_t stream);
/* Give the shape-dependent device memory (activations, recurrent state, launch plans) back without destroying the model: weights
 * stay resident and the next evr_model_reset_states plans again.  The caller has synchronised every stream that ran the model.
 * (ABI 1003: the drop-in frees a saturated model's buffers before its exact-fp32 twin allocates its own.) */
int evr_model_release_shape(evr_model* m);
/* One frame for each of the n_seq sequences: vox [n_seq, B, H, W] (unpadded) -> img
 * [n_seq, 1, H, W] (cropped).  Zero padding to a multiple of 2^pad_multiple_log2 and the centre
 * crop happen inside (utils/util.py:41-59).  flags: bit 0 = apply event-tensor normalization
 * (eval.py:222-223) to vox on the fly using `stats` (double [n_seq,3]). */
int evr_model_step(evr_model* m, const float* vox, const double* stats, float* img, unsigned flags,
                   evr_stream_t stream);
/* Debug/parity access: copy a named internal activation or state (NCHW fp32) to device memory.
 * Names: "head", "enc{i}.conv", "h{i}", "c{i}", "res{i}", "dec{i}".  Returns element count in *n. */
int evr_model_read_tensor(evr_model* m, const char* name, float* dst, int64_t dst_elems,
                          int64_t* n_out, evr_stream_t stream);
/* Direct-convolution FLOPs (2*MAC) of one evr_model_step at the current shape. */
/* From now on every evr_model_step records `ev` on its stream right after launching the layer called `layer` ("enc0.rec",
 * "g1.out", ... -- the names evr_model_profile_read reports); NULL layer or event: off.  Lets a second stream's work (the
 * evaluation of the previous frame) start at a chosen point of the next frame instead of at its beginning. */
int evr_model_set_gate(evr_model* model, const char* layer, evr_event_t ev);
double evr_model_flops_per_step(const evr_model* m);
/* arithmetic of the model's 32-channel-chunk convolutions: 0 exact fp32, 2 f16 + MX-fp8 ("mx"), 3 three f16 products ("h3"),
 * 4 f16 + MX-fp6 ("mx6") -- the mode EVR_ARITH selects, narrowed to what the layout supports ("mx6" needs a layout whose packed
 * tensors are all written as whole 16-channel groups: ConvLSTM UNets with transposed or upsample-conv decoders; others run "mx") */
int evr_model_arith(const evr_model* m);
/* Range guard of the packed activation formats the split arithmetic modes store between layers: number of output runs
 * (4 or 16 channels of one pixel) that left the format's exact range since the last clear, and the layer with most of them
 * (name copied into worst_layer).  0 = every frame so far stayed inside the arithmetic's error analysis; otherwise rerun
 * with EVR_ARITH=h3 (fp32-grade, range +-4094) or EVR_FP32=1.  The reference runs fp32 (model/submodules.py:227-245) and
 * has no such limit.  Synchronises the stream. */
int evr_model_saturation(evr_model* model, int64_t* runs_host, char* worst_layer, size_t worst_len, int clear, evr_stream_t stream);
/* The same per-layer counters (cumulative since the last clear) WITHOUT a synchronisation: copied asynchronously on `stream` to
 * counters_host (caller-owned, pinned host memory, room for max_counters unsigned); *n_counters = how many there are
 * (counters_host == NULL: size query only).  Read them once an event recorded behind this call has completed.  evreal_amd's frame
 * loop polls every chunk of frames this way BEFORE it books the chunk, and re-runs the sequences on an exact-fp32 twin of the
 * model (reserved[2] = 1) when any counter is non-zero: out-of-range activations never reach a score or an output file. */
int evr_model_saturation_async(evr_model* model, unsigned* counters_host, int max_counters, int* n_counters, evr_stream_t stream);
/* Per-layer timing for the roofline block of bench.py.  While enabled, evr_model_step brackets every
 * convolution launch whose layer name contains `filter` ("" = all layers) with HIP events on the launch
 * stream; filter == NULL disables.  evr_model_profile_read synchronises the stream, then returns per
 * layer: name (64 bytes each), summed elapsed milliseconds, direct-conv FLOPs per launch, launch count. */
int evr_model_profile_enable(evr_model* m, const char* filter);
int evr_model_profile_read(evr_model* m, int max_layers, char* names, double* ms, double* flops_per_launch,
                           int64_t* launches, int* n_layers, evr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Post-processing.  Replaces post_process_normalization (eval.py:380-395) + normalize
 * (utils/eval_utils.py:15-35): img <- (img - P_qlo) / (P_qhi - P_qlo) with numpy's default linear
 * percentile; do_exp applies exp() first ('exprobust').  Per image of [n, H, W], in place.
 */
size_t evr_percentile_normalize_workspace_bytes(int n, int H, int W);
int evr_percentile_normalize(float* img, int n, int H, int W, float q_lo, float q_hi, int do_exp,
                             void* workspace, size_t workspace_bytes, evr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Histogram equalisation of the tracker.  Replaces EvalMetricsTracker.histogram_equalization
 * (utils/eval_metrics.py:326-350), applied to the clipped image and reference before the metrics:
 *   EVR_HISTEQ_GLOBAL  skimage.exposure.equalize_hist            (:327-331)
 *   EVR_HISTEQ_LOCAL   skimage.filters.rank.equalize, disk(55)   (:332-339)
 *   EVR_HISTEQ_CLAHE   cv2.createCLAHE(2.0, (8, 8)).apply        (:340-345)
 * img: [n, H, W] fp32 in [0, 1], in place.  PARITY UNPINNED (scikit-image / OpenCV are not available offline): the
 * kernels restate the published algorithms and are checked against oracle/histeq.py.
 */
enum evr_histeq { EVR_HISTEQ_NONE = 0, EVR_HISTEQ_GLOBAL = 1, EVR_HISTEQ_LOCAL = 2, EVR_HISTEQ_CLAHE = 3 };
size_t evr_hist_equalize_workspace_bytes(int n, int H, int W, int mode);
int evr_hist_equalize(float* img, int n, int H, int W, int mode, void* workspace, size_t workspace_bytes,
                      evr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Per-frame metrics.  Replaces EvalMetricsTracker.update's clip (utils/eval_metrics.py:253-255),
 * MseMetric.calculate (:82-84) and SsimMetric.calculate (:95-97; gaussian_weights=True, sigma=1.5,
 * use_sample_covariance=False, data_range=1.0).  img, ref: [n, H, W]; out: double [n, 2] = {mse, ssim}.
 * which: bit 0 = mse, bit 1 = ssim.  clip: clamp both inputs to [0,1] first.
 */
int evr_metrics(const float* img, const float* ref, int n, int H, int W, unsigned which, int clip,
                double* out, void* workspace, size_t workspace_bytes, evr_stream_t stream);
size_t evr_metrics_workspace_bytes(int n, int H, int W);

/* ----------------------------------------------------------------------------------------------
 * LPIPS (AlexNet, v0.1).  Replaces PyIqaMetricFactory.get_metric('lpips') (utils/eval_metrics.py:110-156):
 * gray images are replicated to 3 channels (cv2torch(num_ch=3), eval_utils.py:46-54) and scored in batches.
 * tensors: the metric's state_dict with pyiqa's names ("net.slice1.0.weight" ... "net.slice5.10.bias",
 * "lin0.model.1.weight" ... "lin4.model.1.weight").  img, ref: [n,H,W] in [0,1] (clip: clamp first);
 * out: double [n].  PARITY UNPINNED: pyiqa and its downloaded weights are unavailable offline.
 * The handle owns its feature buffers: sized for the largest n seen at the current H x W (they only grow; a new H x W
 * re-allocates), with the launch plans of the last eight distinct n cached -- a call whose n was seen before, or is below
 * the capacity, neither synchronises nor allocates.  One handle is used from one host thread and one stream at a time.
 */
typedef struct evr_lpips evr_lpips;
int evr_lpips_create(const evr_tensor* tensors, int n_tensors, evr_lpips** out);
int evr_lpips_destroy(evr_lpips* m);
int evr_lpips_forward(evr_lpips* m, const float* img, const float* ref, int n, int H, int W, int clip,
                      double* out, evr_stream_t stream);
double evr_lpips_flops(const evr_lpips* m);

/* ----------------------------------------------------------------------------------------------
 * Colour reconstruction (ColorNet, model/model.py:46-105; merge utils/color_utils.py:53-88).
 * evr_bayer_split: vox [n,B,H,W] -> out [4n,B,H/2,W/2], the R,G,B,W Bayer sub-lattices of each sequence
 *   (model.py:54-57), so the four colour streams run as extra sequences of one batched evr_model_step.
 * evr_color_merge: planes [n,4,H/2,W/2] (R,G,B,W reconstructions, float) + gray [n,H,W] -> bgr_out uint8 [n,H,W,3]:
 *   uint8 truncation (model.py:101), x2 bilinear, Bayer origin shifts, G/W mean, Lab merge with the gray L.
 *   The uint8 planes are pinned against the reference; the merge is floating-point (OpenCV's fixed-point tables are
 *   unavailable offline) -> UNPINNED, may differ by a few LSB.
 */
int evr_bayer_split(const float* vox, int n, int B, int H, int W, float* out, evr_stream_t stream);
int evr_color_merge(const float* planes, const float* gray, int n, int H, int W, unsigned char* bgr_out,
                    evr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Split storage format (host utilities, no GPU): the default arithmetic mode keeps every activation tensor that feeds
 * the matrix cores in 64-byte groups of 16 values: 16 IEEE-half 'hi' (round-to-nearest-even of the value, saturating at
 * +-65504), then 16 OCP e4m3 'lo8' = RNE((value - hi) * 2^12), then 16 OCP e4m3 'x8' = RNE(value) (both saturating at
 * +-448); value ~ hi + lo8 * 2^-12 to 2^-16 relative.  A product is hi_x * hi_w on the f16 matrix path plus the
 * MX-scaled fp8 corrections lo8_x * w8 + x8 * lo8_w (csrc/conv.h).  evr_model_read_tensor decodes the format; these two
 * functions expose the codec itself so it can be checked on a CPU-only host.  n must be a multiple of 16; host pointers.
 */
int evr_split_pack(const float* src, float* dst, int64_t n);
int evr_split_unpack(const float* src, float* dst, int64_t n);
/* the same packing done by the device code the kernels use (src / dst device pointers, in place allowed): lets a test pin the
 * hardware conversions (f16 RNE, OCP e4m3 RNE, saturation) against the host codec bit for bit */
int evr_split_pack_device(const float* src, float* dst, int64_t n, evr_stream_t stream);
/* host-only hooks for the CPU test-suite: the weight form of the split format -- per 16 values f16 hi | e4m3 RNE(w * 2^e) |
 * e4m3 RNE((w - hi) * 2^(e + 12)), e = the largest exponent with max|w| * 2^e <= 224, returned in *exponent -- and the
 * constants of n / d = (umulhi(n, mul) + n) >> shift (n < 2^31) the launch plans carry for the kernels' pixel decode */
int evr_split_pack_weights(const float* src, float* dst, int64_t n, int* exponent);
/* The H2 storage format of the fp32-grade arithmetic mode (EVR_ARITH=h3): every 16 values -> 16 IEEE halves hi = RNE(v 2^e)
 * (saturating) | 16 halves lo = RNE(v 2^e - hi).  Activations use the fixed exponent evr_h2_act_exponent(); weights a per-tensor
 * exponent that brings max|w| to [2^13, 2^14) (returned).  Host codec + the device twin, as for the default format above. */
int evr_h2_pack(const float* src, float* dst, int64_t n);
int evr_h2_pack_weights(const float* src, float* dst, int64_t n, int* exponent);
int evr_h2_unpack(const float* src, float* dst, int64_t n, int exponent);
int evr_h2_pack_device(const float* src, float* dst, int64_t n, evr_stream_t stream);
int evr_h2_act_exponent(void);
/* The P6 storage format of the f16 + MX-fp6 arithmetic mode (EVR_ARITH=mx6): every 16 values -> 16 IEEE halves hi = RNE(v) |
 * 32 e2m3 codes (6 bits, element j at bits 6j of bytes 32-55): element 2i = v_i / S, element 2i + 1 = (v_i - hi_i) 2^11 / S,
 * S = 2^(floor(log2 max|v|) - 2) the group's own scale | its E8M0 byte (byte 56).  Weights: values times 2^e (max|w| 2^e in
 * [2^13, 2^14), e returned), the two elements of a pair swapped, the scale byte lowered by 11.  evr_p6_unpack decodes an
 * ACTIVATION tensor (hi + residual).  Host codec + the device twin (v_cvt_scalef32_2xpk16_fp6_f32), as for the formats above. */
int evr_p6_pack(const float* src, float* dst, int64_t n);
int evr_p6_pack_weights(const float* src, float* dst, int64_t n, int* exponent);
int evr_p6_unpack(const float* src, float* dst, int64_t n);
int evr_p6_pack_device(const float* src, float* dst, int64_t n, evr_stream_t stream);
int evr_fastdiv_magic(unsigned d, unsigned* mul, unsigned* shift);

/* ---- native PNG writers (round 6, ABI 1003) -----------------------------------------------------------------------------------
 * Replaces the reference's per-frame cv2.imwrite (utils/eval_utils.py:80-84; `save_images` is on in config/eval/std.json:9): a pool
 * of host threads encodes 8-bit gray / RGB PNGs (filter 0, one zlib stream of the given level; 0 = stored) and writes them as
 * <folder>/frame_%010d.png.  `frames_host` ([n,H,W] or [n,H,W,3] uint8, host memory) is copied before submit returns; wait blocks
 * until everything submitted is on disk and fails (message: the first failure) if a file could not be written. */
typedef struct evr_png_pool evr_png_pool;
int evr_png_pool_create(int n_threads, int zlib_level, evr_png_pool** out);
int evr_png_pool_submit(evr_png_pool* pool, const char* folder, const int64_t* indices, int n, const unsigned char* frames_host,
                        int H, int W, int channels, int64_t frame_stride /* bytes between frames; 0 = dense */);
int evr_png_pool_wait(evr_png_pool* pool, int64_t* n_written);
int evr_png_pool_destroy(evr_png_pool* pool);

#ifdef __cplusplus
}
#endif
#endif /* EVREAL_HIP_H */
