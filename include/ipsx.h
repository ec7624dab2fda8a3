/*
 * ipsx.h - C ABI of libipsx.so: the MI355X (gfx950) implementation of the
 * Iterative-Patch-Selection no-grad hot path of benbergner/ips.
 *
 * Everything here is plain C: device pointers, sizes, an opaque stream handle
 * (a hipStream_t passed as void*; NULL = the null stream).  No torch types.  All
 * functions enqueue asynchronously on `stream`, never synchronise, never allocate
 * device memory (workspaces are caller-owned) and return 0 on success or a
 * negative IPSX_E* code; ipsx_last_error() gives the message for the calling
 * thread.  Buffers are contiguous fp32 (int64 for indices) unless stated.
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to the reference repo benbergner/ips).
 *
 * Arithmetic contract (restated on the CPU in oracle/ips_oracle.cpp):
 *   - every contraction (convolution taps x channels, Linear rows, q.k, attn.v)
 *     is ONE fp32 fused-multiply-add chain, which is what
 *     v_mfma_f32_32x32x2_f32 computes bit for bit;
 *   - convolution K index is tap-major: k = (ky*KW + kx)*C_in + c;
 *   - chains that run on the matrix cores (convolutions, projector Linear, K and V
 *     projections) visit every aligned group of 8 k in the order 0,4,1,5,2,6,3,7
 *     (the MFMA consumes k in (lane-half 0, lane-half 1) pairs and a lane half
 *     fetches 4 consecutive k with one 16-byte load); all others ascend;
 *   - eval BatchNorm is the affine y = fma(acc, alpha, shift) with
 *     alpha = gamma * (1/sqrt(var+eps)), shift = beta - mean*alpha;
 *   - exp() is ipsx's own fma polynomial (identical bits on host and device); softmax weights are
 *     e * (1 / den): ONE IEEE division per (head, token) row, a multiplication per candidate (round 5);
 *   - row sums (softmax denominators, the transformer's LayerNorm moments) are 64 strided
 *     partial sums combined by an xor butterfly (the wavefront reduction order);
 *   - the projector (ipsx_projector*) evaluates Linear(LayerNorm(x)) with the LayerNorm FOLDED
 *     into the epilogue: acc = the fma chain of the RAW row, t = fma(-mean, colsum[o], acc),
 *     u = t * rstd, y = relu(fma(u, alpha, shift')); the row's moments are eight chains - chain
 *     (h, j) over x[8g + 4h + j], g ascending: exactly what lane (row, half h) of the GEMM's
 *     operand stream holds - folded ((c0+c1)+(c2+c3)) per half, half 0 + half 1;
 *     mean = sum / F, var = fma(-mean, mean, sumsq / F) clamped at 0, rstd = 1 / sqrt(var + eps).
 *     Round 6: a row with var * 17 < sumsq / F (mean^2 / var > 16 - both forms above cancel there) is CENTRED, as
 *     nn.LayerNorm itself does: over d = x - mean the same eight chains sum d and d * d, mean' = mean + E[d],
 *     var = fma(-E[d], E[d], E[d^2]) clamped at 0; acc = the chain over (x - mean') * w, t = acc.  Its statistics read
 *     (mean', -rstd): the SIGN of rstd marks the row.  Well-conditioned rows never take this path (bits unchanged).
 */
#ifndef IPSX_H
#define IPSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version = 100 * major + minor.  The MAJOR number changes whenever the signature or meaning of an exported
 * function changes incompatibly or an export is removed (a caller built against another major must refuse to run:
 * `ipsx_version() / 100 != IPSX_VERSION / 100`); the minor number counts compatible additions.
 *   1.xx  rounds 1-2
 *   2.00  round 3: ipsx_scan / ipsx_scan_range / ipsx_topm take (workspace, workspace_bytes) in front of `stream`,
 *         ipsx_scan_persistent takes ready_per_image, ipsx_projector_stats_publish removed
 *   2.01  round 4 (additions only): ipsx_aggregate_packed, ipsx_set_persistent_wait_ms, ipsx_conv2d_wgrad_nhwc*,
 *         ipsx_pack_conv_weight_strided, ipsx_conv2d_affine_to_nhwc, ipsx_conv2d_lds_nhwc*;
 *         ipsx_projector_stream accepts
 *         short_first <= -3 (guided tile sizes)
 *   2.02  round 4: ipsx_scan_persistent_on, ipsx_scan_persistent_groupable
 *   2.03  round 4: ipsx_scan_persistent_ws, ipsx_scan_range_if_ws (persistent loops for candidate sets beyond the LDS)
 *   3.00  round 5: struct ipsx_conv grew by `colsum` (every struct that embeds it moved) and the projector's arithmetic
 *         contract changed (LayerNorm folded into the epilogue: ipsx_projector* need lin->colsum, ipsx_projector_stats
 *         returns the moments in the operand-stream order); ipsx_weight_colsum added; the two stream kernels publish their
 *         last rows themselves (one more control word each: ipsx_*_stream_ctl_words), ipsx_ips_finish, ipsx_ips_call_run /
 *         ipsx_ips_call_elapsed added
 *   3.01  round 5 (additions only): ipsx_pack_conv_weights_batch, ipsx_conv2d_lds_nhwc_stats (+ _slabs),
 *         ipsx_bn_train_forward_partials, ipsx_stem7x7s2_nhwc (+ _supported), ipsx_conv2d_dgrad_s2_lds_nhwc (+ _supported),
 *         ipsx_maxpool_3x3s2_bwd_nhwc (+ _supported)
 *   3.02  round 6 (additions): ipsx_projector_stream_ctl_zero_words (the stream's control words were re-ordered: what must be
 *         zero comes first); ipsx_scan_workgroups_per_image - candidate sets beyond the LDS with 8 heads and one token run
 *         as a team of workgroups per image; ipsx_scan_workspace_bytes grew for those shapes */
#define IPSX_VERSION 302

#define IPSX_OK            0
#define IPSX_EINVAL       -1      /* bad argument / unsupported shape */
#define IPSX_EHIP         -2      /* HIP runtime error (launch failed ...) */
#define IPSX_EWORKSPACE   -3      /* workspace too small */

int ipsx_version(void);
const char* ipsx_last_error(void);
/* number of GPUs visible, and whether device `dev` is a gfx950 part (1/0) */
int ipsx_device_count(void);
int ipsx_device_is_gfx950(int dev);

/* ------------------------------------------------------------------ encoder
 * Replaces IPSNet.encoder as built by get_conv_patch_enc
 * (architecture/ips_net.py:17-52; called at :209, :227, :273).            */

/* elements of the packed form of an OIHW weight (C_out padded to 32, K to 8) */
size_t ipsx_packed_conv_weight_elems(int c_out, int c_in, int kh, int kw);

/* OIHW fp32 -> MFMA B-operand stream [C_out/32][K/8][64 lanes][4], tap-major K;
 * element j of lane l holds k = 8*group + 4*(l>>5) + j, output channel 32*tile + (l&31) */
int ipsx_pack_conv_weight(const float* w_oihw, int c_out, int c_in, int kh, int kw,
                          float* packed, void* stream);
/* The same from a strided view: element (n, c, ky, kx) of the convolution's weight is w[base + n*s_out + c*s_in + ky*s_ky +
 * kx*s_kx] (element strides, any sign) - a channels-last tensor as it lies, or, with base at the last tap, negated tap
 * strides and swapped channel strides, the weights rotated by 180 degrees and transposed that make ipsx_conv2d_affine_nhwc
 * on dy the data gradient of the convolution. */
int ipsx_pack_conv_weight_strided(const float* w, int64_t base, int c_out, int c_in, int kh, int kw, int64_t s_out,
                                  int64_t s_in, int64_t s_ky, int64_t s_kx, float* packed, void* stream);
/* Up to 32 of those as ONE launch (the training step re-packs every convolution's weights, in forward and in data-gradient
 * form, after each optimizer step: training/iterative.py:155-163).  Same bits as the single calls. */
typedef struct ipsx_pack_job {
    const float* w; int64_t base; int c_out, c_in, kh, kw; int64_t s_out, s_in, s_ky, s_kx; float* packed;
} ipsx_pack_job;
int ipsx_pack_conv_weights_batch(const ipsx_pack_job* jobs, int n_jobs, void* stream);

/* bf16 variant for the reduced-precision trunk: [C_out/32][K/16][64 lanes][8 bf16], round to nearest even */
size_t ipsx_packed_conv_weight_bf16_bytes(int c_out, int c_in, int kh, int kw);
int ipsx_pack_conv_weight_bf16(const float* w_oihw, int c_out, int c_in, int kh, int kw,
                               void* packed, void* stream);

/* eval-mode BatchNorm (nn.BatchNorm2d/1d with running stats) -> per-channel affine.
 * lin_bias (or NULL): bias of a Linear feeding the BatchNorm, folded in as
 * shift = fma(lin_bias, alpha, shift)                                        */
int ipsx_bn_affine(const float* gamma, const float* beta, const float* mean,
                   const float* var, const float* lin_bias, float eps, int c,
                   float* alpha, float* shift, void* stream);

/* split variant for precision 2 ("fp32x3"): every weight as three bf16 terms hi + mid + lo (exact),
 * [C_out/32][K/16][plane][64 lanes][8 bf16]; three times the bf16 stream                            */
size_t ipsx_packed_conv_weight_x3_bytes(int c_out, int c_in, int kh, int kw);
int ipsx_pack_conv_weight_x3(const float* w_oihw, int c_out, int c_in, int kh, int kw,
                             void* packed, void* stream);

/* stem of the split trunks (precision 1: planes = 1, precision 2: planes = 3): the (c_out, 1, 7, 7) weights as the
 * B-operand stream of a contraction whose K is laid out as k = 8 ky + kx (zero weights for ky = 7 and kx = 7),
 * [C_out/32][4][plane][64 lanes][8 bf16]; goes into the stem's ipsx_conv.w_packed_bf16                       */
size_t ipsx_packed_stem_weight_split_bytes(int c_out, int planes);
int ipsx_pack_stem_weight_split(const float* w_oihw, int c_out, int planes, void* packed, void* stream);

typedef struct ipsx_conv {
    int c_in, c_out, kh, kw, stride, pad;
    const float* w_packed;        /* ipsx_pack_conv_weight output            */
    const float* alpha;           /* c_out, BatchNorm scale (or NULL = 1)    */
    const float* shift;           /* c_out, BatchNorm shift / bias (or NULL) */
    const void* w_packed_bf16;    /* bf16 operand stream for ipsx_trunk.precision: ipsx_pack_conv_weight_bf16
                                     (precision 1) or ipsx_pack_conv_weight_x3 (precision 2); NULL = fp32 only */
    const float* colsum;          /* c_out, sum over c_in of the weights (ipsx_weight_colsum): the projector's Linear
                                     only (ipsx_projector*: the folded LayerNorm's mean term); NULL elsewhere      */
} ipsx_conv;

/* one residual block: BasicBlock (n_conv = 2) or Bottleneck (n_conv = 3)     */
typedef struct ipsx_block {
    int n_conv;
    ipsx_conv conv[3];
    int has_down;                 /* 1x1 strided projection on the shortcut  */
    ipsx_conv down;
} ipsx_block;

/* the nn.Sequential of ips_net.py:35-50: stem conv 7x7/2 + BN + ReLU,
 * max-pool 3x3/2, residual blocks, global average pool                      */
typedef struct ipsx_trunk {
    int c_in, h, w;               /* patch shape (C, h, w)                    */
    ipsx_conv stem;
    int n_block;
    const ipsx_block* blocks;     /* HOST array of n_block descriptors        */
    int precision;                /* 0 = fp32 (exact, default); 1 = bf16 operands / fp32 accumulate in the
                                     residual stages; 2 = "fp32x3": every fp32 operand split exactly into three
                                     bf16 terms, the six significant products on the bf16 matrix pipe, fp32
                                     accumulate - fp32-grade accuracy, not bit-identical to precision 0.
                                     1 and 2: fused 1x32x32 trunk only, need w_packed_bf16                */
    int patch_dtype;              /* storage type of the `patches` argument of the encode calls: 0 = float32
                                     (default), 1 = bfloat16, 2 = float16 (BASELINE configs[4]: half-precision patch
                                     storage; precision 1 or 2 only - the exact path reads float32)        */
} ipsx_trunk;

/* y = act(affine(conv(x)) [+ residual]); x (n,c_in,h,w), y (n,c_out,ho,wo) NCHW */
int ipsx_conv2d_affine(const ipsx_conv* cv, const float* x, const float* residual,
                       float* y, int64_t n, int h, int w, int relu, void* stream);
/* x NCHW as above, residual / y channels-last (n,ho,wo,c_out): the stems (1 or 3 input channels) in front of the
 * channels-last layers */
int ipsx_conv2d_affine_to_nhwc(const ipsx_conv* cv, const float* x, const float* residual,
                               float* y, int64_t n, int h, int w, int relu, void* stream);
/* the same on channels-last activations: x (n,h,w,c_in), residual / y (n,ho,wo,c_out); C_in % 32 == 0.
 * This is the fast layer-by-layer path (16-byte operand loads); a Linear over rows is h = w = 1. */
int ipsx_conv2d_affine_nhwc(const ipsx_conv* cv, const float* x, const float* residual,
                            float* y, int64_t n, int h, int w, int relu, void* stream);
/* The fused trunk's stages as stand-alone plain convolutions of channels-last maps (training step: forward and data
 * gradient on the maps of 32-px patches, each layer's input read once and LDS-resident for all taps): 64 -> 64 3x3 on 8x8,
 * 128 -> 128 3x3 on 4x4, 64 -> 128 3x3 / 2 and 1x1 / 2 from 8x8 (ipsx_conv2d_lds_nhwc_supported).  cv: kernel, stride, pad and
 * w_packed (ipsx_pack_conv_weight[_strided]); alpha / shift are ignored. */
int ipsx_conv2d_lds_nhwc_supported(int c_in, int c_out, int k, int stride, int pad, int h, int w);
int ipsx_conv2d_lds_nhwc(const ipsx_conv* cv, const float* x, float* y, int64_t n, int h, int w, void* stream);
/* The same with the BatchNorm BATCH STATISTICS of the output taken off the accumulators (training step: conv -> bn of a
 * torchvision BasicBlock, architecture/ips_net.py:273 under net.train()): partial[slab][0 | 1][C_out] = sum (y - shift[c]),
 * sum (y - shift[c])^2 over the slab's output rows, slab = 4 patches, ipsx_conv2d_lds_nhwc_stats_slabs(n) of them; shift: a
 * per-channel constant near the mean (the BatchNorm's running mean; NULL = 0).  ipsx_bn_train_forward_partials combines them
 * (fp64, slab order: deterministic) and applies the BatchNorm - the reduction pass over y is gone.  partial NULL: plain. */
/* The 1 -> 64 channel 7x7 / 2 stem on 32 x 32 patches as a stand-alone layer, channels-last output (n, 16, 16, 64): the
 * training step's first convolution (ips_net.py:29-31 under net.train()) on the matrix cores - the fused trunk's stem
 * without its BatchNorm / ReLU / max-pool epilogue; same bits as ipsx_conv2d_affine_to_nhwc without affine. */
int ipsx_stem7x7s2_nhwc_supported(int c_in, int c_out, int kh, int kw, int stride, int pad, int h, int w);
int ipsx_stem7x7s2_nhwc(const ipsx_conv* conv, const float* x, float* y, int64_t n, const float* shift, float* partial,
                        void* stream);          /* shift / partial: as ipsx_conv2d_lds_nhwc_stats (slabs of 4 patches); NULL: none */
/* The data gradient of the 32-px trunk's strided convolution (64 -> 128 channels, 3x3 / 2, 8x8 -> 4x4 maps; the arguments of
 * _supported describe that FORWARD convolution): dy (n, 4, 4, 128) -> dx (n, 8, 8, 64), channels-last, taken by parity class
 * of the input pixel - nine taps instead of the thirty-six of a stride-1 convolution over dy spread on a zero map.
 * w_packed_dgrad: ipsx_pack_conv_weight_strided of the weights rotated by 180 degrees and transposed (128 -> 64, 3x3). */
int ipsx_conv2d_dgrad_s2_lds_nhwc_supported(int c_in, int c_out, int k, int stride, int pad, int h, int w);
int ipsx_conv2d_dgrad_s2_lds_nhwc(const float* w_packed_dgrad, int k, const float* dy, float* dx, int64_t n, void* stream);
                                  /* k = 3 (pad 1), or k = 1: the 1x1 / 2 projection beside it (pad 0; three of four input pixels
                                   * receive zeros) */
int64_t ipsx_conv2d_lds_nhwc_stats_slabs(int64_t n);
int ipsx_conv2d_lds_nhwc_stats(const ipsx_conv* conv, const float* x, float* y, int64_t n, int h, int w, const float* shift,
                               float* partial, void* stream);
/* Weight gradient of that convolution for the training step (reference: loss.backward() of training/iterative.py:157-163
 * through the BasicBlocks of architecture/ips_net.py:264-283):
 *   dw[co][ky][kx][ci] = sum over (img, oy, ox) of dy[img,oy,ox,co] * x[img, stride*oy + ky - pad, stride*ox + kx - pad, ci]
 * x (n,h,w,c_in) and dy (n,ho,wo,c_out) channels-last, dw in the memory order of a channels-last weight tensor
 * ((c_out, c_in, kh, kw) with strides (kh*kw*c_in, 1, kw*c_in, c_in)).  fp32 MFMA, reduction over the pixels split across
 * workgroups and added in a fixed order (deterministic).  c_out a multiple of 64, c_in a multiple of 64 or 1 (the 7x7 stem)
 * (ipsx_conv2d_wgrad_nhwc_supported).
 * The data gradient is ipsx_conv2d_affine_nhwc itself: on dy with the weights rotated by 180 degrees and transposed. */
int ipsx_conv2d_wgrad_nhwc_supported(int c_in, int c_out, int kh, int kw, int stride, int pad);
size_t ipsx_conv2d_wgrad_nhwc_workspace_bytes(int64_t n, int c_in, int c_out, int kh, int kw);
int ipsx_conv2d_wgrad_nhwc(const float* x, const float* dy, int64_t n, int h, int w, int c_in, int c_out, int kh, int kw,
                           int stride, int pad, float* dw, void* workspace, size_t workspace_bytes, void* stream);
int ipsx_maxpool_3x3s2_nhwc(const float* x, float* y, int64_t n, int c, int h, int w, void* stream);
/* Its backward pass for the training step (16 x 16 maps, C % 32 == 0): dx (n, 16, 16, C) from x and dy (n, 8, 8, C), the
 * gradient of a window to its first maximum in row-major order - ATen's max_pool2d_with_indices_backward bit for bit, without
 * an index tensor. */
int ipsx_maxpool_3x3s2_bwd_nhwc_supported(int c, int h, int w);
int ipsx_maxpool_3x3s2_bwd_nhwc(const float* x, const float* dy, float* dx, int64_t n, int c, int h, int w, void* stream);
/* (n,hw,c) -> (n,c) */
int ipsx_avgpool_nhwc(const float* x, float* y, int64_t n, int c, int hw, void* stream);
/* nn.MaxPool2d(3, 2, 1) */
int ipsx_maxpool_3x3s2(const float* x, float* y, int64_t n, int c, int h, int w, void* stream);
/* nn.AdaptiveAvgPool2d(1): (n,c,hw) -> (n,c) */
int ipsx_avgpool(const float* x, float* y, int64_t n, int c, int hw, void* stream);

size_t ipsx_trunk_workspace_bytes(const ipsx_trunk* t, int64_t n_patch);
/* name of the kernel family ipsx_trunk_encode will use for this trunk (static string) */
const char* ipsx_trunk_kernel(const ipsx_trunk* t);
/* patches (n_patch, c_in, h, w) -> emb (n_patch, D); picks the fused LDS-resident
 * kernel when the trunk matches it, the layer-by-layer kernels otherwise       */
int ipsx_trunk_encode(const ipsx_trunk* t, const float* patches, int64_t n_patch,
                      float* emb, void* workspace, size_t workspace_bytes, void* stream);

/* Encode the patches patches[index[0..n_index)] (int32 indices, device) -> emb (n_index, D): lets a caller
 * encode a strided subset (e.g. columns lo..hi of a (B, N, ...) tensor) without copying it.  Fused trunk only. */
int ipsx_trunk_encode_indexed(const ipsx_trunk* t, const float* patches, const int32_t* index,
                              int64_t n_index, float* emb, void* stream);

/* Same result as ipsx_trunk_encode, with exact blank-patch deduplication (all-zero patches share one
 * embedding in eval mode; ~93 % of Megapixel-MNIST patches): only the non-blank patches and one blank
 * are encoded, everything on the device.  Fused 1x32x32 trunk only.  n_encoded (device int32, or NULL)
 * receives the number of patches actually encoded.                                              */
size_t ipsx_trunk_dedup_workspace_bytes(const ipsx_trunk* t, int64_t n_patch);
int ipsx_trunk_encode_dedup(const ipsx_trunk* t, const float* patches, int64_t n_patch, float* emb,
                            void* workspace, size_t workspace_bytes, int32_t* n_encoded, void* stream);
/* Same, with the per-patch flags (1 = has a non-zero element) already known, e.g. from
 * ipsx_patchify_sparse: the pass that reads every patch to find the blank ones is skipped.        */
int ipsx_trunk_encode_dedup_flagged(const ipsx_trunk* t, const float* patches, int64_t n_patch,
                                    const int32_t* nonblank, float* emb, void* workspace,
                                    size_t workspace_bytes, int32_t* n_encoded, void* stream);

/* ---------------------------------------------------------- image -> patches
 * The step that feeds ips(): replaces, for a whole batch on the device,
 *   img.unfold(1, ph, sh).unfold(2, pw, sw).permute(1, 2, 0, 3, 4).reshape(-1, C, ph, pw)
 * of data/megapixel_mnist/mnist_dataset.py:44-51 and data/traffic/traffic_dataset.py:336-343.
 * img (b,c,h,w) -> patches (b, ny*nx, c, ph, pw), ny = (h-ph)/sh+1, nx = (w-pw)/sw+1, patch order
 * row-major over (py, px).  ipsx_patchify_count returns ny*nx (0 for an invalid geometry).         */
int64_t ipsx_patchify_count(int h, int w, int ph, int pw, int sh, int sw);
int ipsx_patchify(const float* img, int b, int c, int h, int w, int ph, int pw, int sh, int sw,
                  float* patches, void* stream);
/* Megapixel-MNIST's on-disk form straight to patches (mnist_dataset.py:34-51): image i owns the
 * non-zeros [offsets[i], offsets[i+1]) of (index, value); index is the flat position in an (h, w, c)
 * canvas (the reference's _img_shape), value its pixel.  patches is zeroed and filled; nonblank
 * (b*ny*nx int32, or NULL) is set to 1 for every patch that received a non-zero value.  Indices
 * outside the canvas are skipped (the host validates them before upload; numpy would raise).
 * offsets, index and value are device pointers; nnz = offsets[b].                                  */
int ipsx_patchify_sparse(const int64_t* index, const float* value, const int64_t* offsets, int64_t nnz,
                         int b, int c, int h, int w, int ph, int pw, int sh, int sw,
                         float* patches, int32_t* nonblank, void* stream);

/* Replaces IPSNet.encoder as built by get_projector (ips_net.py:54-60):
 * ReLU(BN1d(Linear(LayerNorm_noaffine(x)))); x (n,f) -> out (n,d).
 * `lin` is the Linear packed as a 1x1 conv (c_in=f, c_out=d; f % 32 == 0) whose alpha/shift
 * hold the BatchNorm affine with the Linear bias folded into shift
 * (shift' = fma(bias, alpha, shift)) and whose colsum holds ipsx_weight_colsum of the weights.
 * The LayerNorm is folded into the GEMM's epilogue (see "Arithmetic contract") from a (mean, rstd)
 * pass; workspace: ipsx_projector_workspace_bytes(n).                                              */
/* cs[o] = sum over c of w[o][c] of a Linear's (c_out, c_in) fp32 weights: ascending in float64, rounded once */
int ipsx_weight_colsum(const float* w, int c_out, int c_in, float* colsum, void* stream);
int ipsx_projector(const ipsx_conv* lin, const float* x, int64_t n, float ln_eps,
                   float* out, void* workspace, size_t workspace_bytes, void* stream);
size_t ipsx_projector_workspace_bytes(int64_t n);      /* 8 bytes per row: (mean, rstd); rstd < 0: a centred row (above) */
/* the two halves of ipsx_projector, for callers that compute the row statistics (n x 2 floats: mean, rstd) ahead of
 * the GEMM - e.g. on another stream, beside the GEMM of an earlier slab (the pass is HBM-bound, the GEMM MFMA-bound) */
int ipsx_projector_stats(const float* x, int64_t n, int f, float ln_eps, float* stats, void* stream);
int ipsx_projector_apply(const ipsx_conv* lin, const float* x, int64_t n, const float* stats, float* out, void* stream);
/* ipsx_projector_apply that also does what ipsx_publish_rows(ready, value) does, at its start: the launches enqueued before
 * it (the logits of the previous slab) have completed, which is all a publication says - one launch less per slab in a
 * pipeline that feeds ipsx_scan_persistent. */
int ipsx_projector_apply_publish(const ipsx_conv* lin, const float* x, int64_t n, const float* stats, float* out,
                                 int32_t* ready, int32_t value, void* stream);

/* ONE image's patches through the fused 1x32x32 trunk AND their logits as ONE persistent launch that feeds
 * ipsx_scan_persistent patch by patch (reference: self.encoder(...) chunk by chunk in IPSNet.ips, architecture/ips_net.py:
 * 213-241, and transformer.py:71-83 for the logits of emb + pos): resident workgroups pull two patches at a time, encode
 * them (out: emb (n_patch, 128)), compute their logits against the folded query v_packed from emb + pos (pos: (n_patch,
 * 128) or NULL; out: logits (n_patch, r)) and advance *ready past every completed pair in order.  Bit-identical to
 * ipsx_trunk_encode + ipsx_logits.  ctl: ipsx_trunk_stream_ctl_words(n_patch) int32 words ZEROED before every call.
 * workgroups <= 0: a few more than compute units (one that finds no unit at once starts late or finds nothing left);
 * quad_pulls: a workgroup's first quad_pulls pulls are four patches (the trunk's full rate), the rest two (the finer
 * deal); < 0: as many as leave a round of two-patch pulls.                                                              */
size_t ipsx_trunk_stream_ctl_words(int64_t n_patch);
int ipsx_trunk_stream_supported(const ipsx_trunk* t, int d, int r);
int ipsx_trunk_stream(const ipsx_trunk* t, const float* patches, int64_t n_patch, float* emb, const float* pos,
                      const float* v_packed, int r, float* logits, int32_t* ctl, int32_t* ready, int workgroups,
                      int quad_pulls, void* stream);

/* The projector AND the logits of one slide as ONE persistent launch that feeds ipsx_scan_persistent row by row
 * (reference: the projector of architecture/ips_net.py:60-66 applied chunk by chunk in IPSNet.ips :213-241, and
 * transformer.py:71-83 for the logits).  Resident workgroups pull 64-row tiles off a counter; a tile's Linear (its rows'
 * LayerNorm moments come off the GEMM's own operand registers: every row is read once) + folded LayerNorm + BatchNorm +
 * ReLU (out: emb, (n, 512)), and logits against the folded query v_packed
 * (ipsx_fold_query; r = h * n_token <= 32; out: logits (n, r)) are computed in place, and `*ready` - the progress word
 * of the selection loop - is advanced past every completed 32-row unit in order.  Several slides: x holds them one
 * after the other (n rows in all, slide_rows each, a multiple of 32), ready[s] is slide s's word, and the stream of
 * tiles runs across the slides' ends.  Bit-identical to ipsx_projector_stats + ipsx_projector_apply + ipsx_logits.  ctl: ipsx_projector_stream_ctl_words(n) int32 words, the first ipsx_projector_stream_ctl_zero_words(n) of them ZEROED before every call (3.02; the rest are hand-over accumulators, written before they are read).
 * workgroups <= 0: 7 of every 8 compute units (the loop's unit stays free); short_first >= 0: that many workgroups
 * start with a 32-row tile, -1: half of them (completions then do not come in bursts), -2: every tile is 32 rows (a
 * steady supply at 12 % less throughput), -3 - (head + 8 tail), head < 8: every workgroup's first `head` pulls and the
 * last `tail` x workgroups units are 32-row tiles (early first rows, an even end; head = 0: half the workgroups start
 * short as with -1), and the few units left over when every workgroup has had its whole share go out as four column
 * quarters each (the logits' accumulators pass from quarter to quarter: the same bits).  ipsx_projector_stream_supported: 1x1 Linear with
 * 512 outputs and its column sums (lin->colsum), c_in % 32 == 0, 64 <= n < 2^31.                    */
size_t ipsx_projector_stream_ctl_words(int64_t n);
size_t ipsx_projector_stream_ctl_zero_words(int64_t n);
int ipsx_projector_stream_supported(const ipsx_conv* lin, int64_t n, int r);
int ipsx_projector_stream(const ipsx_conv* lin, const float* x, int64_t n, int64_t slide_rows, float ln_eps, float* emb,
                          const float* v_packed, int r, float* logits, int32_t* ctl, int32_t* ready,
                          int workgroups, int short_first, void* stream);

/* ------------------------------------------------------------------- scorer
 * Replaces MultiHeadCrossAttention.get_attn + ScaledDotProductAttention.
 * compute_attn + Transformer.get_scores (architecture/transformer.py:29-34,
 * 71-83, 143-148) and IPSNet.score_and_select (ips_net.py:136-155).          */

/* qs[t][o] = (sum_c q[t][c]*wq[o][c]) / temperature; q (T,D), wq (H*Dk,D)
 * (transformer.py:76 and the division of :31)                               */
int ipsx_query_proj(const float* q, const float* wq, float temperature,
                    int n_token, int d, int hdk, float* qs, void* stream);

/* Attention logits before the softmax (transformer.py:77,31: matmul(q / temperature, k_w(x)^T)):
 *   logit[n][h*T + t] = sum_j qs[t][h,j] * sum_c W_k[h*dk + j][c] * x[n][c],   x = emb (+ pos)
 * evaluated with the query folded into the key weights (the two sums commute; this order is the arithmetic
 * contract, restated by the oracle):  v = ipsx_fold_query(qs, wk_packed)  once per call, then
 * logit[n][r] = sum_c x[n][c] * v[r][c]  on the matrix cores - H*T*D multiply-adds per patch instead of
 * H*Dk*D, so the kernel is bound by reading the embeddings.
 * emb (b,n,d) with batch stride emb_bstride (floats); pos (b|1,n,d) or NULL, pos_bstride = 0 broadcasts one
 * (n,d) table over the batch; wk_packed = ipsx_pack_conv_weight of k_w.weight viewed as (H*Dk, D, 1, 1);
 * v_packed: ipsx_folded_query_elems(h, n_token, d) floats; r = h * n_token; logits (b,n,r).            */
size_t ipsx_folded_query_elems(int h, int n_token, int d);
int ipsx_fold_query(const float* qs, const float* wk_packed, int h, int dk, int n_token, int d,
                    float* v_packed, void* stream);
int ipsx_logits(const float* emb, int64_t emb_bstride,
                const float* pos, int64_t pos_bstride,
                const float* v_packed,
                int b, int64_t n, int d, int r,
                float* logits, int64_t logits_bstride, void* stream);

/* BASELINE configs[4]: the same logits on the bf16 matrix pipe - x = emb (+ pos) rounded to bfloat16, the folded query
 * rounded to bfloat16 (ipsx_fold_query_bf16: ipsx_folded_query_bf16_bytes(h, n_token, d) bytes, from the float32 folded
 * query's inputs), float32 accumulation.  No reference behaviour to match: tolerance-tested against ipsx_logits.   */
size_t ipsx_folded_query_bf16_bytes(int h, int n_token, int d);
int ipsx_fold_query_bf16(const float* qs, const float* wk_packed, int h, int dk, int n_token, int d,
                         void* v_packed_bf16, void* stream);
int ipsx_logits_bf16(const float* emb, int64_t emb_bstride, const float* pos, int64_t pos_bstride,
                     const void* v_packed_bf16, int b, int64_t n, int d, int r,
                     float* logits, int64_t logits_bstride, void* stream);

/* Order of equal scores in the top-M steps of ipsx_scan / ipsx_scan_range / ipsx_topm (process-wide).
 * 1 (default) = the reference's: torch.topk on CPU returns what libstdc++'s nth_element + sort (or
 * partial_sort when 64*M <= L) leave behind (ATen/native/TopKImpl.h:45-68, reference call site
 * ips_net.py:148); those routines are replayed on the device - by ipsx_topm whenever two of the first M+1
 * ranked scores are equal; by the selection loops (ipsx_scan*, round 5) when two NEIGHBOURS among the first
 * M+1 canonical ranks have equal scores AND bit-identical logit rows (duplicated patches: they tie in the
 * reference's arithmetic as well, so its order there is torch's) - two different rows whose scores collide in
 * the last bit of THIS arithmetic keep the canonical order (in the reference's arithmetic they are an ulp apart).
 * 2 = the loops replay on every tie too (rounds 1-4).
 * 0 = canonical: score descending, earlier candidate position first (no sequential step).
 * Returns the previous mode; any other argument only queries.                                       */
int ipsx_set_tie_order(int mode);

/* The chunk loop of IPSNet.ips (ips_net.py:213-241) on cached logits
 * (b, n, h*T): memory = first m patches; for every chunk of `i` further
 * patches: candidates = memory ++ chunk, per-(h,t) softmax over candidates,
 * mean over h then t, keep top m (score descending, ties: earlier candidate
 * position first).  One workgroup per image, one launch for the whole loop.
 * mem_idx (b,m) int64 = selected patch indices in final score order;
 * mem_score (b,m) or NULL; tie_flag (b) int32 or NULL: set when two candidates
 * straddling a top-m boundary had bit-equal scores (the reference's own order is
 * then implementation-defined, ips_net.py:199).                               */
int ipsx_scan(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
              int64_t* mem_idx, float* mem_score, int32_t* tie_flag,
              void* workspace, size_t workspace_bytes, void* stream);

/* Candidate sets that do not fit one compute unit's LDS - m + i above ~4,000, up to 16,384; the reference's shipped
 * CAMELYON configuration is m = i = 5000 (config/camelyon_config.yml:35-36, torch.topk over 10,000 candidates,
 * ips_net.py:148) - keep only the ranking in LDS and need a caller-owned device workspace of this many bytes (0 = none
 * needed, workspace may be NULL).  ipsx_scan / ipsx_scan_range return IPSX_EWORKSPACE when it is missing or too small. */
size_t ipsx_scan_workspace_bytes(int b, int m, int i, int h, int n_token);
/* Workgroups (= compute units) the loop of ONE image occupies for a call of b images: 1 for every LDS-resident shape; the
 * shapes above with 8 heads and one token - the shipped CAMELYON configuration - run as a TEAM of up to 8 workgroups per
 * image that share the element-wise passes and pre-sort their shares of the ranking (csrc/scan_large_team.h; same results).
 * A caller that launches a persistent producer beside persistent loops leaves b x this many units free (3.02). */
int ipsx_scan_workgroups_per_image(int b, int m, int i, int h, int n_token);

/* Iterations [it_begin, it_end) of the same loop.  it_begin = 0 starts from the first m patches, it_begin > 0
 * resumes from the state a previous call left in mem_idx; only logits rows below m + it_end*i are read, so a
 * range can run (on another stream) while later patches are still being encoded.  tie_flag is only ever SET.  */
int ipsx_scan_range(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                    int64_t it_begin, int64_t it_end, int64_t* mem_idx, float* mem_score,
                    int32_t* tie_flag, void* workspace, size_t workspace_bytes, void* stream);

/* The whole loop as ONE launch that may start before any logits exist (overlap with the encoder without re-launching
 * per part): the kernel waits until *ready (device int32, written with ipsx_publish_rows on another stream after the
 * kernels that produced the rows) says the rows it is about to read are in memory.  The wait is bounded: 50 ms (see
 * ipsx_set_persistent_wait_ms) WITHOUT PROGRESS of any of the call's progress words; on a timeout, or when *ready is set
 * negative, the kernel ends and sets bit 0 of *status (results are then invalid - ipsx_scan_range_if behind it redoes the
 * loop in the same call; bit 1 = the kernel is resident).  ready_per_image != 0: `ready` is an array of b words and image k follows ready[k] - a
 * producer that works through the images one after the other publishes each image's rows as it goes, and image k's loop
 * runs beside the production of image k + 1.
 * Shapes: ipsx_scan_persistent_supported() != 0.     */
int ipsx_scan_persistent_supported(int m, int i, int h, int n_token);
int ipsx_scan_persistent(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                         int64_t* mem_idx, float* mem_score, int32_t* tie_flag, const int32_t* ready,
                         int32_t ready_per_image, int32_t* status, void* stream);
/* ipsx_scan_persistent with FEWER resident workgroups than images (0 < workgroups < b; otherwise one per image):
 * workgroup w runs the loops of images w, w + workgroups, ... one after the other - for a producer that works through
 * the images in order and needs longer for an image than its loop does (the projector of 16 CAMELYON slides: two
 * resident loops keep up, and 14 more compute units stay the projector's).  Shapes: ipsx_scan_persistent_groupable(). */
int ipsx_scan_persistent_groupable(int m, int i, int h, int n_token);
int ipsx_scan_persistent_on(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                            int64_t* mem_idx, float* mem_score, int32_t* tie_flag, const int32_t* ready,
                            int32_t ready_per_image, int32_t* status, int workgroups, void* stream);
/* The persistent loop and its conditional recovery launch for EVERY shape ipsx_scan covers: candidate sets beyond the LDS
 * (ipsx_scan_workspace_bytes() > 0: the reference's shipped CAMELYON M = I = 5000) take the workspace of ipsx_scan; for
 * the other shapes these are ipsx_scan_persistent_on / ipsx_scan_range_if and the workspace is ignored (2.03). */
int ipsx_scan_persistent_ws(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                            int64_t* mem_idx, float* mem_score, int32_t* tie_flag, const int32_t* ready,
                            int32_t ready_per_image, int32_t* status, int workgroups, void* workspace,
                            size_t workspace_bytes, void* stream);
int ipsx_scan_range_if_ws(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                          int64_t it_begin, int64_t it_end, int64_t* mem_idx, float* mem_score,
                          int32_t* tie_flag, const int32_t* cond, int32_t cond_mask, void* workspace,
                          size_t workspace_bytes, void* stream);
int ipsx_publish_rows(int32_t* ready, int32_t value, void* stream);
/* longest wait of a persistent loop (and of ipsx_scan_gate) without progress, in milliseconds (1 .. 20000; default 50);
 * returns the previous value, ms <= 0 only queries */
int ipsx_set_persistent_wait_ms(int ms);
/* ipsx_scan_range that runs only when (*cond & cond_mask) != 0, tested ON THE DEVICE by every workgroup as it starts (no
 * host synchronisation): enqueued behind ipsx_scan_persistent with cond = its status word and mask 1, it redoes the loop
 * with plain launches in the very call whose persistent loop gave up waiting, and costs one empty launch otherwise.
 * Shapes of the LDS-resident loops (ipsx_scan_workspace_bytes() == 0). */
int ipsx_scan_range_if(const float* logits, int b, int64_t n, int m, int i, int h, int n_token,
                       int64_t it_begin, int64_t it_end, int64_t* mem_idx, float* mem_score,
                       int32_t* tie_flag, const int32_t* cond, int32_t cond_mask, void* stream);
/* ipsx_logits for rows [0, n) and, in the SAME launch, the LayerNorm row moments (ipsx_projector_stats) of stats_n rows
 * of stats_x - the next slab the projector is about to take: two short latency-bound launches of a slab-by-slab producer
 * of ipsx_scan_persistent as one. */
int ipsx_logits_stats(const float* emb, int64_t emb_bstride, const float* pos, int64_t pos_bstride,
                      const float* v_packed, int b, int64_t n, int d, int r, float* logits, int64_t logits_bstride,
                      const float* stats_x, int64_t stats_n, int stats_f, float ln_eps, float* stats_out, void* stream);
/* holds `stream` (one waiting thread, bounded) until the persistent scan that owns `status` is resident: enqueue it on
 * the producing stream right after launching the scan, so that the producers do not take the compute units first */
int ipsx_scan_gate(const int32_t* status, void* stream);

/* Transformer.get_scores on arbitrary embeddings x (b,l,d) -> scores (b,l);
 * attn (b,h,T,l) optionally written too (get_attn).                          */
int ipsx_scores(const float* x, const float* wk_packed, const float* qs,
                int b, int l, int d, int h, int dk, int n_token,
                float* scores, float* attn, void* workspace, size_t workspace_bytes,
                void* stream);
size_t ipsx_scores_workspace_bytes(int b, int l, int d, int h, int n_token);

/* torch.topk(scores, m, dim=-1)[1] (ips_net.py:148): (b,l) -> (b,m) int64, l <= 16,384.  Rows beyond the LDS
 * (l above ~8,000) need ipsx_topm_workspace_bytes(b, l, m) bytes of device workspace (0 = none, NULL allowed). */
int ipsx_topm(const float* scores, int b, int l, int m, int64_t* top_idx,
              int32_t* tie_flag, void* workspace, size_t workspace_bytes, void* stream);
size_t ipsx_topm_workspace_bytes(int b, int l, int m);

/* ------------------------------------------------------------------- gather
 * Replaces the torch.gather calls of ips_net.py:245-250: dst[b][j] = src[b][idx[b][j]]
 * for rows of row_bytes bytes (multiple of 4).  src_bstride_rows = rows between
 * batches of src (0 = one table shared by all batches).                       */
int ipsx_gather_rows(const void* src, const int64_t* idx, void* dst,
                     int b, int64_t n_rows, int m, int64_t row_bytes,
                     int64_t src_bstride_rows, void* stream);

/* The end of IPSNet.ips (architecture/ips_net.py:243-250: mem_patch = patches[mem_idx], mem_pos = pos_enc[mem_idx]) in ONE
 * launch for calls whose selection ran as a resident loop: both gathers (rows of 16-byte units; pos may be NULL;
 * *_bstride_rows = rows per image, 0 = one table for every image), a fresh copy of the loop's (b, m) index buffer, and -
 * status_host != NULL - the loop's status word stored to its pinned host mirror (read by the caller one call later). */
int ipsx_ips_finish(const void* patches, int64_t patch_row_bytes, int64_t patch_bstride_rows, int64_t n_rows,
                    const void* pos, int64_t pos_row_bytes, int64_t pos_bstride_rows, const int64_t* mem_idx, int b, int m,
                    void* mem_patch, void* mem_pos, int64_t* mem_idx_out, const int32_t* status, int32_t* status_host,
                    void* stream);

/* ONE IPSNet.ips call whose selection loop is resident (architecture/ips_net.py:169-262 for one image on the fused trunk,
 * or for feature slides through the projector), enqueued by ONE library call: fill of the control words, the loop on
 * `side_stream` (ipsx_scan_persistent_ws), the gate, the producer on `stream` (ipsx_trunk_stream when `trunk` is set -
 * b == 1 -, ipsx_projector_stream when `lin` is set), the conditional recovery launch (ipsx_scan_range_if_ws) and the end
 * of the call (ipsx_ips_finish), with the two cross-stream hand-overs between them.  Same kernels, same results as the
 * entry points called one by one; the host's share of a call drops from ~12 calls to one, and nothing of the caller's
 * runtime (an interpreter's collector or allocator) sits between the launch of the loop and the launch of the producer it
 * waits for - a host thread the OS deschedules there still can: the loop's wait is bounded and the call then redoes it.
 *   words: words_total int32 = tie flags [b] | progress words [b] | status | the producer's control words
 *          (ipsx_trunk_stream_ctl_words / ipsx_projector_stream_ctl_words); zeroed by the call (what has to be: see ipsx_projector_stream_ctl_zero_words).
 *   timing_slot in [0, 64): the producer's launch is bracketed by a library-owned HIP event pair; ipsx_ips_call_elapsed
 *          (slot, &ms) reads it once the stream has been synchronised.  -1: no events.
 * The two hand-over events are the library's, one pair per device; the call enqueues under a per-device lock, so calls
 * from several threads (on streams of their own) do not mix their hand-overs.                                          */
typedef struct ipsx_ips_call {
    int b; int64_t n; int m, i, h, n_token;
    float* logits; int64_t* mem_idx; int32_t* words; int64_t words_total;
    int loops; void* scan_workspace; size_t scan_workspace_bytes;
    const ipsx_trunk* trunk; const float* pos; int quad_pulls;
    const ipsx_conv* lin; float ln_eps; int short_first;
    const void* x; float* emb; const float* v_packed; int r; int workgroups;
    const void* src; int64_t src_row_bytes, src_bstride_rows;
    const void* pos_table; int64_t pos_row_bytes, pos_bstride_rows;
    void* mem_patch; void* mem_pos; int64_t* mem_idx_out; int32_t* status_host;
    int timing_slot;
    void* stream; void* side_stream;
} ipsx_ips_call;
int ipsx_ips_call_run(const ipsx_ips_call* c);
int ipsx_ips_call_elapsed(int slot, float* ms);

/* --------------------------------------------------------------- aggregation
 * Replaces Transformer.forward (transformer.py:85-109,122-132,150-152) and the
 * task heads (ips_net.py:72-81,157-166) in eval / no-grad mode.               */
typedef struct ipsx_transf {
    int n_token, h, d, dk, dv, d_inner;
    const float *q, *wq, *wk, *wv, *fc;          /* (T,D) (HDk,D) (HDk,D) (HDv,D) (D,HDv) */
    const float *ln1_g, *ln1_b;                  /* LayerNorm after attention, eps 1e-6   */
    const float *w1, *b1, *w2, *b2;              /* (Dinner,D) (Dinner) (D,Dinner) (D)    */
    const float *ln2_g, *ln2_b;
    float temperature, ln_eps;
} ipsx_transf;

size_t ipsx_aggregate_workspace_bytes(const ipsx_transf* t, int b, int m);
/* x (b,m,d) -> out (b,T,d) */
int ipsx_aggregate(const ipsx_transf* t, const float* x, int b, int m, float* out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* The same with operands the caller keeps between calls (they depend on the parameters only): vq_packed = ipsx_fold_query
 * of (q, wq, wk), wv_packed = ipsx_pack_conv_weight(wv, h*dv, d, 1, 1); either may be NULL (then it is made here, in the
 * workspace, as ipsx_aggregate does).  Saves four small launches per call in an evaluation loop. */
int ipsx_aggregate_packed(const ipsx_transf* t, const float* vq_packed, const float* wv_packed, const float* x, int b,
                          int m, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* out[b][c] = act(w[c,:] . emb[b][token,:] + bias[c]); act 0 = softmax, 1 = sigmoid */
int ipsx_head(const float* emb, int b, int n_token, int d, int token,
              const float* w, const float* bias, int n_class, int act,
              float* out, void* stream);

/* ------------------------------------------------------- training step (with-grad forward, SURVEY 8 f N-b)
 * Training-mode BatchNorm2d of the ResNet trunk with its residual add and ReLU, forward and backward, on channels-last
 * activations x (rows, c), rows = patches * pixels.  Replaces, under net.train() with autograd
 * (training/iterative.py:158-163 -> ips_net.py:273), the bn / += identity / relu kernels of a torchvision BasicBlock:
 *   forward : y = [relu]( (x - mean_batch) * invstd_batch * gamma + beta [+ residual] ); running_mean / running_var are
 *             updated in place with `momentum` (unbiased variance), save_mean / save_invstd (c floats) go to backward;
 *   backward: g = dy masked by y > 0 (when relu); dbeta = sum g; dgamma = sum g * xhat;
 *             dx = gamma * invstd * (g - dbeta / rows - xhat * dgamma / rows); dresidual = g (when not NULL).
 * c / 4 must be a power of two <= 256 (ipsx_bn_train_supported); workspace: ipsx_bn_train_workspace_floats floats.
 * fp32 results to rounding of torch.nn.functional.batch_norm (another summation order), deterministic.          */
int ipsx_bn_train_supported(int64_t rows, int c);
size_t ipsx_bn_train_workspace_floats(int64_t rows, int c);
int ipsx_bn_train_forward(const float* x, const float* residual, int64_t rows, int c, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                          int relu, float* y, float* save_mean, float* save_invstd, float* workspace, void* stream);
/* ipsx_bn_train_forward without its reduction pass: the per-slab partial sums come from the convolution that produced x
 * (ipsx_conv2d_lds_nhwc_stats), taken around `shift` (may alias running_mean: it is read before the update). */
int ipsx_bn_train_forward_partials(const float* x, const float* residual, int64_t rows, int c, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                   int relu, float* y, float* save_mean, float* save_invstd, const float* partial,
                                   int64_t slabs, const float* shift, void* stream);
int ipsx_bn_train_backward(const float* dy, const float* y, const float* x, int64_t rows, int c, const float* gamma,
                           const float* save_mean, const float* save_invstd, int relu, float* dx, float* dresidual,
                           float* dgamma, float* dbeta, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IPSX_H */
