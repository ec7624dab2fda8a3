// trunk.hip - walks the nn.Sequential of IPSNet.get_conv_patch_enc (reference
// architecture/ips_net.py:35-50) over a batch of patches: stem conv 7x7/2 + BN +
// ReLU, max-pool 3x3/2, residual blocks, global average pool.
//
// Host-side runtime only: it sequences the kernels of conv.hip (or the fused
// LDS-resident kernel of fused_trunk.hip when the trunk matches it) on the
// caller's stream, in chunks of patches so the activation workspace stays bounded.

#include <algorithm>
#include <cstdlib>

#include "ipsx_common.h"

namespace ipsx {

// conv.hip
int conv2d_affine_impl(const ipsx_conv* cv, const float* x, const float* residual, float* y, int64_t n, int h,
                       int w, int relu, int out_nhwc, void* stream);
// fused_stage.hip: the leading 64 -> 64 BasicBlocks on a small map, LDS-resident (50-px patches: 13x13)
int fused_stage64_blocks(const ipsx_block* blocks, int n_block, int h, int w);
int fused_stage64(const ipsx_block* blocks, int n_block, const float* x, float* y, int64_t n, int h, int w, hipStream_t s);
int fused_stem_pool50(const ipsx_trunk* t, const float* patches, float* y, int64_t n, hipStream_t s);   // 1 = ran, 0 = other shape
bool fused_stem_pool50_covers(const ipsx_trunk* t);
bool fused_stem_pool100x3_covers(const ipsx_trunk* t);
// fused_trunk.hip
bool fused_trunk_supported(const ipsx_trunk* t);
int fused_trunk_encode(const ipsx_trunk* t, const float* patches, int64_t n, float* emb, hipStream_t s);
int fused_trunk_stream(const ipsx_trunk* t, const float* patches, int64_t n, float* emb, const float* pos, const float* v_packed,
                       int r, float* logits, int32_t* ctl, int32_t* ready, int workgroups, int quad_pulls, hipStream_t s);

struct TrunkGeom {
    size_t max_elems;   // largest per-patch activation (floats) of any layer
    int d_out;
};

static int trunk_geom(const ipsx_trunk* t, TrunkGeom* g) {
    IPSX_REQUIRE(t && t->n_block >= 0 && (t->n_block == 0 || t->blocks), "trunk: missing blocks");
    IPSX_REQUIRE(t->c_in > 0 && t->h > 0 && t->w > 0, "trunk: bad patch shape %dx%dx%d", t->c_in, t->h, t->w);
    IPSX_REQUIRE(t->stem.c_in == t->c_in, "trunk: stem expects %d channels, patches have %d", t->stem.c_in, t->c_in);
    int h = conv_out(t->h, t->stem.kh, t->stem.stride, t->stem.pad);
    int w = conv_out(t->w, t->stem.kw, t->stem.stride, t->stem.pad);
    IPSX_REQUIRE(h > 0 && w > 0, "trunk: patch too small");
    int c = t->stem.c_out;
    size_t mx = (size_t)c * h * w;
    h = conv_out(h, 3, 2, 1); w = conv_out(w, 3, 2, 1);
    for (int b = 0; b < t->n_block; ++b) {
        const ipsx_block& B = t->blocks[b];
        IPSX_REQUIRE(B.n_conv == 2 || B.n_conv == 3, "trunk: block %d has %d convs", b, B.n_conv);
        int ch = h, cw = w, cc = c;
        for (int j = 0; j < B.n_conv; ++j) {
            const ipsx_conv& cv = B.conv[j];
            IPSX_REQUIRE(cv.c_in == cc, "trunk: block %d conv %d expects %d channels, gets %d", b, j, cv.c_in, cc);
            IPSX_REQUIRE(cv.c_in % 32 == 0, "trunk: block %d conv %d has %d input channels (need a multiple of 32)", b, j, cv.c_in);
            ch = conv_out(ch, cv.kh, cv.stride, cv.pad); cw = conv_out(cw, cv.kw, cv.stride, cv.pad);
            IPSX_REQUIRE(ch > 0 && cw > 0, "trunk: feature map vanished in block %d", b);
            cc = cv.c_out;
            mx = std::max(mx, (size_t)cc * ch * cw);
        }
        if (B.has_down) {
            IPSX_REQUIRE(B.down.c_in == c && B.down.c_out == cc, "trunk: block %d shortcut shape", b);
            IPSX_REQUIRE(conv_out(h, B.down.kh, B.down.stride, B.down.pad) == ch, "trunk: block %d shortcut size", b);
        } else {
            IPSX_REQUIRE(cc == c && ch == h && cw == w, "trunk: block %d needs a projection shortcut", b);
        }
        h = ch; w = cw; c = cc;
    }
    g->max_elems = mx;
    g->d_out = c;
    return IPSX_OK;
}

// patches per chunk: 4 activation buffers of chunk*max_elems floats, <= 24 GiB in all.  Sized for 288 GB of HBM: the
// deep layers of a trunk have few output pixels per patch, and a chunk has to be large for THEM to fill 256 CUs
// (traffic signs, 512 channels at 4x4: 768 patches are 384 workgroups - fewer than the GPU runs at once)
// The budget is the smallest of: IPSX_TRUNK_WORKSPACE_MB (default 24576), 40 % of the device memory that is free right
// now (other processes / the training step's activations live there too).  ipsx_trunk_encode itself sizes its chunks
// from the workspace it is GIVEN, so a caller may hand over less than ipsx_trunk_workspace_bytes proposes.
static size_t trunk_budget() {
    size_t budget = (size_t)24 << 30;
    if (const char* e = getenv("IPSX_TRUNK_WORKSPACE_MB")) {
        const long long mb = atoll(e);
        if (mb > 0) budget = (size_t)mb << 20;
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) budget = std::min(budget, free_b / 10 * 4);
    else (void)hipGetLastError();
    return budget;
}

static int64_t chunk_for(const TrunkGeom& g, int64_t n, size_t bytes) {
    int64_t cap = (int64_t)(bytes / (4 * g.max_elems * sizeof(float)));
    return std::min<int64_t>(n, cap);
}

static int64_t trunk_chunk(const TrunkGeom& g, int64_t n) {
    return std::max<int64_t>(chunk_for(g, n, trunk_budget()), 1);
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API size_t ipsx_trunk_workspace_bytes(const ipsx_trunk* t, int64_t n_patch) {
    TrunkGeom g;
    if (trunk_geom(t, &g) != IPSX_OK || n_patch <= 0) return 0;
    if (fused_trunk_supported(t)) return 0;
    return (size_t)trunk_chunk(g, n_patch) * g.max_elems * sizeof(float) * 4;
}

IPSX_API const char* ipsx_trunk_kernel(const ipsx_trunk* t) {
    if (t && fused_trunk_supported(t))
        return t->precision == 2 ? "fused_trunk_x3_kernel" : (t->precision == 1 ? "fused_trunk_bf16_kernel" : "fused_trunk_kernel");
    if (t && t->n_block >= 2 && fused_stem_pool50_covers(t) &&
        fused_stage64_blocks(t->blocks, t->n_block, 13, 13) > 0)          // the reference's shipped 50-px Megapixel-MNIST trunk
        return "stem_pool50_kernel + fused_stage64_kernel (layer1, LDS-resident) + conv_nhwc_kernel (layer2, layer by layer)";
    if (t && fused_stem_pool100x3_covers(t))                               // the traffic-sign trunk: fused stem + pool, then layer by layer
        return "stem_pool100x3_kernel + conv_nhwc_kernel (layer by layer)";
    return "conv_nhwc_kernel (layer by layer)";
}

IPSX_API int ipsx_trunk_encode(const ipsx_trunk* t, const float* patches, int64_t n_patch, float* emb,
                               void* workspace, size_t workspace_bytes, void* stream) {
    TrunkGeom g;
    IPSX_TRY(trunk_geom(t, &g));
    IPSX_REQUIRE(patches && emb && n_patch >= 0, "trunk_encode: bad arguments");
    IPSX_REQUIRE(t->precision == 0 || fused_trunk_supported(t), "trunk_encode: the bf16 / fp32x3 paths exist for the fused 1x32x32 trunk only");
    IPSX_REQUIRE(t->patch_dtype == 0 || (fused_trunk_supported(t) && t->precision != 0),
                 "trunk_encode: half-precision patch storage exists for the fused 1x32x32 trunk at precision 1 / 2 only");
    if (n_patch == 0) return IPSX_OK;
    if (fused_trunk_supported(t)) return fused_trunk_encode(t, patches, n_patch, emb, as_stream(stream));

    const int64_t chunk = workspace ? chunk_for(g, n_patch, workspace_bytes) : 0;      // chunks fit what the caller gave
    if (chunk < 1)
        return fail(IPSX_EWORKSPACE, "trunk_encode: workspace %zu B < %zu B (one patch)", workspace_bytes,
                    g.max_elems * sizeof(float) * 4);
    const size_t buf_elems = (size_t)chunk * g.max_elems;
    float* buf[4];
    for (int i = 0; i < 4; ++i) buf[i] = static_cast<float*>(workspace) + i * buf_elems;
    const size_t patch_elems = (size_t)t->c_in * t->h * t->w;

    for (int64_t p0 = 0; p0 < n_patch; p0 += chunk) {
        const int64_t n = std::min(chunk, n_patch - p0);
        int h = conv_out(t->h, t->stem.kh, t->stem.stride, t->stem.pad);
        int w = conv_out(t->w, t->stem.kw, t->stem.stride, t->stem.pad);
        int c = t->stem.c_out;
        // stem reads the NCHW patches and writes channels-last; everything after it is channels-last
        const int fused_stem = fused_stem_pool50(t, patches + p0 * patch_elems, buf[1], n, as_stream(stream));
        if (fused_stem < 0) return IPSX_EHIP;
        if (!fused_stem) {
            IPSX_TRY(conv2d_affine_impl(&t->stem, patches + p0 * patch_elems, nullptr, buf[0], n, t->h, t->w, 1, 1, stream));
            IPSX_TRY(ipsx_maxpool_3x3s2_nhwc(buf[0], buf[1], n, c, h, w, stream));
        }
        h = conv_out(h, 3, 2, 1); w = conv_out(w, 3, 2, 1);
        int cur = 1;                                   // buf[cur] holds the block input
        int b_first = 0;
        if (c == 64) {                                 // layer1 on a small map: all of its convolutions in one LDS-resident kernel
            const int nf = fused_stage64_blocks(t->blocks, t->n_block, h, w);
            if (nf > 0) {
                IPSX_TRY(fused_stage64(t->blocks, nf, buf[1], buf[0], n, h, w, as_stream(stream)));
                cur = 0;
                b_first = nf;
            }
        }
        for (int b = b_first; b < t->n_block; ++b) {
            const ipsx_block& B = t->blocks[b];
            // free buffers: the three that are not `cur`
            int fr[3], k = 0;
            for (int i = 0; i < 4; ++i) if (i != cur) fr[k++] = i;
            const float* src = buf[cur];
            int ch = h, cw = w, si = -1;
            for (int j = 0; j < B.n_conv - 1; ++j) {   // conv -> BN -> ReLU
                const ipsx_conv& cv = B.conv[j];
                const int di = (si == fr[0]) ? fr[1] : fr[0];
                IPSX_TRY(ipsx_conv2d_affine_nhwc(&cv, src, nullptr, buf[di], n, ch, cw, 1, stream));
                ch = conv_out(ch, cv.kh, cv.stride, cv.pad); cw = conv_out(cw, cv.kw, cv.stride, cv.pad);
                src = buf[di]; si = di;
            }
            const ipsx_conv& last = B.conv[B.n_conv - 1];
            const float* shortcut = buf[cur];
            if (B.has_down) {                          // 1x1 strided conv + BN on the identity path
                IPSX_TRY(ipsx_conv2d_affine_nhwc(&B.down, buf[cur], nullptr, buf[fr[2]], n, h, w, 0, stream));
                shortcut = buf[fr[2]];
            }
            const int oi = (si == fr[0]) ? fr[1] : fr[0];
            // last conv -> BN -> += identity -> ReLU
            IPSX_TRY(ipsx_conv2d_affine_nhwc(&last, src, shortcut, buf[oi], n, ch, cw, 1, stream));
            h = conv_out(ch, last.kh, last.stride, last.pad); w = conv_out(cw, last.kw, last.stride, last.pad);
            c = last.c_out;
            cur = oi;
        }
        IPSX_TRY(ipsx_avgpool_nhwc(buf[cur], emb + (size_t)p0 * g.d_out, n, c, h * w, stream));
    }
    return IPSX_OK;
}

// One image: trunk AND logits of its patches as ONE persistent launch that feeds ipsx_scan_persistent patch by patch
// (fused_trunk_stream_kernel).  ctl: ipsx_trunk_stream_ctl_words(n) int32 words ZEROED by the caller before every call.
IPSX_API size_t ipsx_trunk_stream_ctl_words(int64_t n_patch) { return n_patch > 0 ? (size_t)ipsx::cdiv(n_patch, 2) + 3 : 0; }   // (+ the exit counter)

IPSX_API int ipsx_trunk_stream_supported(const ipsx_trunk* t, int d, int r) {
    return t && ipsx::fused_trunk_supported(t) && t->precision == 0 && t->patch_dtype == 0 && d == 128 && r >= 1 && r <= 32 ? 1 : 0;
}

IPSX_API int ipsx_trunk_stream(const ipsx_trunk* t, const float* patches, int64_t n_patch, float* emb, const float* pos,
                               const float* v_packed, int r, float* logits, int32_t* ctl, int32_t* ready, int workgroups,
                               int quad_pulls, void* stream) {
    IPSX_REQUIRE(t && patches && emb && v_packed && logits && ctl && ready && n_patch > 0, "trunk_stream: bad arguments");
    IPSX_REQUIRE(ipsx_trunk_stream_supported(t, 128, r), "trunk_stream: the fused fp32 1x32x32 trunk with 128 features and at most 32 logits per patch");
    return ipsx::fused_trunk_stream(t, patches, n_patch, emb, pos, v_packed, r, logits, ctl, ready, workgroups, quad_pulls,
                                    ipsx::as_stream(stream));
}
