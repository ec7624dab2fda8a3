// bn_train.hip - training-mode BatchNorm2d fused with the residual add and the ReLU of a ResNet BasicBlock, forward and
// backward, on channels-last activations (rows = patches x pixels, C contiguous).
//
// Reference: training/iterative.py:158-163 runs `net(mem_patch, mem_pos)` with autograd under net.train(), where
// architecture/ips_net.py:273 sends the M selected patches through the ResNet trunk with BatchNorm in batch-statistics
// mode (torchvision BasicBlock: conv - bn - relu - conv - bn - (+identity) - relu).  On stock ROCm ops every bn / add /
// relu is a kernel of its own in both directions (~25 % of the step's device time at the MNIST configuration, more than
// the convolutions); here a BatchNorm with its add and ReLU is two memory passes forward (batch moments; normalise +
// add + ReLU) and two backward (d_gamma / d_beta; dx and the residual's gradient).
//
// Numerics: fp32 like torch.nn.functional.batch_norm; the moments are accumulated around a per-channel shift (row 0 of
// the activation) in fp32 per slab of rows and combined over the slabs in fp64 in a fixed order - deterministic, and
// free of the E[x^2] - E[x]^2 cancellation.  Results agree with the stock path to fp32 rounding (tests/test_hip_train.py),
// they are not bit-identical to it (a different summation order), which the training path never was across devices.

#include "ipsx_common.h"
#include "ipsx_math.h"

namespace ipsx {

constexpr int BN_MAX_SLABS = 512;

struct BnShape {
    long long rows;
    int c, cg;           // cg = C / 4 float4 columns (a power of two <= 256)
    int slabs;
    long long slab;      // rows per slab
};

__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Two per-channel sums over a slab of rows -> partial[slab][0|1][C].  Threads: (row lane, float4 column); the row lanes
// are combined through LDS in ascending order.
//   BWD = false: sum (x - k), sum (x - k)^2          with k = x[row 0] (the same shift in every slab)
//   BWD = true : sum g, sum g * xhat                 with g = dy (masked by y > 0 under ReLU), xhat = (x - mean) * invstd
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                        const float4* __restrict__ y, const float4* __restrict__ mean,
                                                        const float4* __restrict__ invstd, BnShape s, int relu,
                                                        float* __restrict__ partial) {
    __shared__ float4 red[2][256];
    const int tid = threadIdx.x, c4 = tid & (s.cg - 1), rl = tid / s.cg, nrl = 256 / s.cg;
    const long long r0 = (long long)blockIdx.x * s.slab, r1 = r0 + s.slab < s.rows ? r0 + s.slab : s.rows;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    const float4 k = BWD ? mean[c4] : x[c4];
    const float4 is = BWD ? invstd[c4] : make_float4(1.f, 1.f, 1.f, 1.f);
    // four rows per trip, every load issued before the first add (the pass is latency-bound otherwise)
    for (long long r = r0 + rl; r < r1; r += 4ll * nrl) {
        float4 xv[4], gv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long ru = r + (long long)u * nrl;
            const bool in = ru < r1;
            const long long i = (in ? ru : r) * s.cg + c4;
            xv[u] = x[i];
            if (BWD) {
                gv[u] = dy[i];
                if (relu) yv[u] = y[i];
                if (!in) gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (!in) {
                xv[u] = k;                       // contributes (k - k) = 0
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 d = f4_sub(xv[u], k);
            if (!BWD) {
                a = f4_add(a, d);
                b = f4_add(b, f4_mul(d, d));
            } else {
                float4 g = gv[u];
                if (relu) {
                    g.x = yv[u].x > 0.f ? g.x : 0.f; g.y = yv[u].y > 0.f ? g.y : 0.f;
                    g.z = yv[u].z > 0.f ? g.z : 0.f; g.w = yv[u].w > 0.f ? g.w : 0.f;
                }
                a = f4_add(a, g);
                b = f4_add(b, f4_mul(g, f4_mul(d, is)));
            }
        }
    }
    red[0][tid] = a;
    red[1][tid] = b;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < nrl; ++j) {
            a = f4_add(a, red[0][j * s.cg + c4]);
            b = f4_add(b, red[1][j * s.cg + c4]);
        }
        float4* p = reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * 2 * s.c);
        p[c4] = a;
        p[s.cg + c4] = b;
    }
}

// Combine the slabs (fp64, ascending slab order inside 64 slab lanes, then the lanes in ascending order); one workgroup
// per 4 channels.
//   BWD = false: mean, invstd (+ running statistics, momentum update with the unbiased variance as torch does)
//   BWD = true : d_beta = sum g, d_gamma = sum g * xhat
// (x: the shift the partial sums were taken around, per channel - row 0 of the activation, or, for partial sums that come
//  from a convolution's epilogue, the running mean itself: no __restrict__ on the two, the read precedes the update)
template <bool BWD>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, const float* x,
                                                          BnShape s, float eps, float momentum,
                                                          float* running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ double red[2][64][4];
    const int cl = threadIdx.x & 3, gl = threadIdx.x >> 2, c = blockIdx.x * 4 + cl;
    double a = 0.0, b = 0.0;
    if (c < s.c)
        for (int g = gl; g < s.slabs; g += 64) {
            a += (double)partial[(size_t)g * 2 * s.c + c];
            b += (double)partial[(size_t)g * 2 * s.c + s.c + c];
        }
    red[0][gl][cl] = a;
    red[1][gl][cl] = b;
    __syncthreads();
    if (gl != 0 || c >= s.c) return;
    for (int j = 1; j < 64; ++j) { a += red[0][j][cl]; b += red[1][j][cl]; }
    if (BWD) {
        out0[c] = (float)b;          // d_gamma
        out1[c] = (float)a;          // d_beta
    } else {
        const double n = (double)s.rows, m1 = a / n;
        const double mean = (x ? (double)x[c] : 0.0) + m1;
        double var = b / n - m1 * m1;
        var = var > 0.0 ? var : 0.0;
        out0[c] = (float)mean;
        out1[c] = 1.0f / __builtin_sqrtf((float)var + eps);
        if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) {
            const double unbiased = s.rows > 1 ? var * n / (n - 1.0) : var;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// y = [relu]( (x - mean) * invstd * gamma + beta [+ residual] )
__global__ __launch_bounds__(256) void bn_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ res,
                                                       const float4* __restrict__ mean, const float4* __restrict__ invstd,
                                                       const float4* __restrict__ gamma, const float4* __restrict__ beta,
                                                       long long total4, int cg, int relu, float4* __restrict__ y) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i & (cg - 1));
    const float4 xv = x[i], m = mean[c4], is = invstd[c4], g = gamma[c4], b = beta[c4];
    float4 v = f4_add(f4_mul(f4_mul(f4_sub(xv, m), is), g), b);
    if (res) v = f4_add(v, res[i]);
    if (relu) {
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
    }
    y[i] = v;
}

// dx = gamma * invstd * (g - d_beta / n - xhat * d_gamma / n);  d_residual = g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                           const float4* __restrict__ x, const float4* __restrict__ mean,
                                                           const float4* __restrict__ invstd, const float4* __restrict__ gamma,
                                                           const float4* __restrict__ dgamma, const float4* __restrict__ dbeta,
                                                           long long total4, int cg, int relu, float inv_n,
                                                           float4* __restrict__ dx, float4* __restrict__ dres) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i & (cg - 1));
    float4 g = dy[i];
    if (relu) {
        const float4 yv = y[i];
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
        g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    }
    if (dres) dres[i] = g;
    const float4 is = invstd[c4];
    const float4 xh = f4_mul(f4_sub(x[i], mean[c4]), is);
    const float4 sc = f4_mul(gamma[c4], is);
    const float4 dg = dgamma[c4], db = dbeta[c4];
    float4 v;
    v.x = sc.x * (g.x - db.x * inv_n - xh.x * (dg.x * inv_n));
    v.y = sc.y * (g.y - db.y * inv_n - xh.y * (dg.y * inv_n));
    v.z = sc.z * (g.z - db.z * inv_n - xh.z * (dg.z * inv_n));
    v.w = sc.w * (g.w - db.w * inv_n - xh.w * (dg.w * inv_n));
    dx[i] = v;
}

static bool bn_shape(int64_t rows, int c, BnShape* s) {
    if (rows <= 0 || c < 4 || c % 4 != 0) return false;
    const int cg = c / 4;
    if (cg > 256 || (cg & (cg - 1)) != 0) return false;
    s->rows = rows; s->c = c; s->cg = cg;
    // slabs of at least one sweep of the workgroup's row lanes x 4, at most BN_MAX_SLABS of them
    const int64_t sweep = (int64_t)(256 / cg) * 4;
    int64_t slabs = cdiv(rows, sweep);
    if (slabs > BN_MAX_SLABS) slabs = BN_MAX_SLABS;
    s->slab = cdiv(rows, slabs);
    s->slabs = (int)cdiv(rows, s->slab);
    return true;
}

}  // namespace ipsx

using namespace ipsx;

IPSX_API int ipsx_bn_train_supported(int64_t rows, int c) {
    BnShape s;
    return bn_shape(rows, c, &s) ? 1 : 0;
}

IPSX_API size_t ipsx_bn_train_workspace_floats(int64_t rows, int c) {
    BnShape s;
    if (!bn_shape(rows, c, &s)) return 0;
    return (size_t)s.slabs * 2 * c;
}

IPSX_API int ipsx_bn_train_forward(const float* x, const float* residual, int64_t rows, int c, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, int relu, float* y, float* save_mean, float* save_invstd,
                                   float* workspace, void* stream) {
    BnShape s;
    IPSX_REQUIRE(bn_shape(rows, c, &s), "bn_train_forward: rows = %lld, C = %d (C / 4 must be a power of two <= 256)",
                 (long long)rows, c);
    IPSX_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && workspace, "bn_train_forward: bad arguments");
    hipStream_t st = as_stream(stream);
    bn_reduce_kernel<false><<<dim3(s.slabs), dim3(256), 0, st>>>(reinterpret_cast<const float4*>(x), nullptr, nullptr,
                                                                 nullptr, nullptr, s, 0, workspace);
    bn_finalize_kernel<false><<<dim3((unsigned)cdiv(c, 4)), dim3(256), 0, st>>>(workspace, x, s, eps, momentum,
                                                                                  running_mean, running_var, save_mean,
                                                                                  save_invstd);
    const long long total4 = (long long)rows * s.cg;
    bn_apply_kernel<<<dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, st>>>(
        reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(residual),
        reinterpret_cast<const float4*>(save_mean), reinterpret_cast<const float4*>(save_invstd),
        reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(beta), total4, s.cg, relu,
        reinterpret_cast<float4*>(y));
    return launched("bn_train_forward");
}

IPSX_API int ipsx_bn_train_forward_partials(const float* x, const float* residual, int64_t rows, int c, const float* gamma,
                                            const float* beta, float eps, float momentum, float* running_mean,
                                            float* running_var, int relu, float* y, float* save_mean, float* save_invstd,
                                            const float* partial, int64_t slabs, const float* shift, void* stream) {
    BnShape s;
    IPSX_REQUIRE(bn_shape(rows, c, &s), "bn_train_forward_partials: rows = %lld, C = %d (C / 4 must be a power of two <= 256)",
                 (long long)rows, c);
    IPSX_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && partial && slabs > 0 && slabs < (1 << 30),
                 "bn_train_forward_partials: bad arguments");
    hipStream_t st = as_stream(stream);
    s.slabs = (int)slabs;                                   // the producer's slabs (bn_finalize_kernel only counts them)
    bn_finalize_kernel<false><<<dim3((unsigned)cdiv(c, 4)), dim3(256), 0, st>>>(partial, shift, s, eps, momentum, running_mean,
                                                                                  running_var, save_mean, save_invstd);
    const long long total4 = (long long)rows * s.cg;
    bn_apply_kernel<<<dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, st>>>(
        reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(residual),
        reinterpret_cast<const float4*>(save_mean), reinterpret_cast<const float4*>(save_invstd),
        reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(beta), total4, s.cg, relu,
        reinterpret_cast<float4*>(y));
    return launched("bn_train_forward_partials");
}

IPSX_API int ipsx_bn_train_backward(const float* dy, const float* y, const float* x, int64_t rows, int c,
                                    const float* gamma, const float* save_mean, const float* save_invstd, int relu,
                                    float* dx, float* dresidual, float* dgamma, float* dbeta, float* workspace,
                                    void* stream) {
    BnShape s;
    IPSX_REQUIRE(bn_shape(rows, c, &s), "bn_train_backward: rows = %lld, C = %d (C / 4 must be a power of two <= 256)",
                 (long long)rows, c);
    IPSX_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace && (y || !relu),
                 "bn_train_backward: bad arguments");
    hipStream_t st = as_stream(stream);
    bn_reduce_kernel<true><<<dim3(s.slabs), dim3(256), 0, st>>>(
        reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y),
        reinterpret_cast<const float4*>(save_mean), reinterpret_cast<const float4*>(save_invstd), s, relu, workspace);
    bn_finalize_kernel<true><<<dim3((unsigned)cdiv(c, 4)), dim3(256), 0, st>>>(workspace, nullptr, s, 0.f, 0.f, nullptr,
                                                                                 nullptr, dgamma, dbeta);
    const long long total4 = (long long)rows * s.cg;
    bn_bwd_apply_kernel<<<dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, st>>>(
        reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(x),
        reinterpret_cast<const float4*>(save_mean), reinterpret_cast<const float4*>(save_invstd),
        reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(dgamma),
        reinterpret_cast<const float4*>(dbeta), total4, s.cg, relu, 1.0f / (float)rows, reinterpret_cast<float4*>(dx),
        reinterpret_cast<float4*>(dresidual));
    return launched("bn_train_backward");
}
