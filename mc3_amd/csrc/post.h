// post.h -- everything around the chain update: exchange helpers, standalone likelihoods,
// posterior assembly, initial population, checkpoint.
#pragma once
#include "core.h"

namespace mc3hip {

// after the all-gather: the other ranks' rows of ring slot `roff` go to their place in Z
// (refresh ctl[0] + roff); own rows are already there
__global__ void ring_in_kernel(const DevProblem P, const int roff)
{
    const size_t zs = (size_t)P.zs;
    const double *src = P.xring + (size_t)roff * P.nchains * zs;
    double *dst = P.Z + (size_t)(P.M0 + (P.ctl[0] + roff) * P.nchains) * zs;
    const int nf = P.nfree;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.nchains * nf;
         i += gridDim.x * blockDim.x) {
        const int row = i / nf, k = i - row * nf;
        if (row < P.chain0 || row >= P.chain0 + P.nlocal)
            dst[(size_t)row * zs + k] = src[(size_t)row * zs + k];
    }
}

// log_post of this rank's chains over refreshes [R0, R0+nref) -> dense [nref][nlocal]
// (send buffer of ONE all-gather), and back from the gathered [nranks][nref][nlocal]
__global__ void logpost_pack_kernel(const DevProblem P, long long R0, long long nref,
                                    double *dense)
{
    const long long total = nref * P.nlocal;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / P.nlocal, lc = i - r * P.nlocal;
        dense[i] = P.log_post[P.M0 + (R0 + r) * P.nchains + P.chain0 + lc];
    }
}

__global__ void logpost_unpack_kernel(const DevProblem P, long long R0, long long nref,
                                      const double *gathered)
{
    const long long total = nref * P.nchains;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / P.nchains, ch = i - r * P.nchains;
        const long long rank = ch / P.nlocal, lc = ch - rank * P.nlocal;
        if (rank != P.chain0 / P.nlocal)
            P.log_post[P.M0 + (R0 + r) * P.nchains + ch] =
                gathered[(rank * nref + r) * P.nlocal + lc];
    }
}

template <int MT>
__global__ void eval_model_kernel(const DevProblem P, const double *params, double *out)
{
    __shared__ double s_p[MC3HIP_MAX_PARAMS];
    for (int q = threadIdx.x; q < P.nparams; q += blockDim.x)
        s_p[q] = params[q];
    __syncthreads();
    Model<MT> model;
    model.init(P, s_p);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.ndata;
         i += gridDim.x * blockDim.x)
        out[i] = model.at(i);
}

// standalone D4 transform (ms.dwt_daub4): one workgroup, global-memory ping-pong
__global__ void daub4_kernel(double *a, double *tmp, int n, int inverse)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    if (!inverse) {
        for (int nn = n; nn >= 4; nn >>= 1) {
            const int nh = nn >> 1;
            for (int i = tid; i < nh; i += nt) {
                double s, d;
                d4_pair(a[2 * i], a[2 * i + 1], a[(2 * i + 2) & (nn - 1)],
                        a[(2 * i + 3) & (nn - 1)], s, d);
                tmp[i] = s;
                tmp[i + nh] = d;
            }
            __syncthreads();
            for (int i = tid; i < nn; i += nt)
                a[i] = tmp[i];
            __syncthreads();
        }
    } else {
        for (int nn = 4; nn <= n; nn <<= 1) {                        // wavelet.h:37-43
            const int nh = nn >> 1;
            for (int i = tid; i < nh; i += nt) {
                const int im = (i + nh - 1) & (nh - 1);
                const double sp = a[im], dp = a[im + nh], sc = a[i], dc = a[i + nh];
                double e = D4C2 * sp;
                e = mc3_fma(D4C1, dp, e);
                e = mc3_fma(D4C0, sc, e);
                e = mc3_fma(D4C3, dc, e);
                double o = D4C3 * sp;
                o = mc3_fma(-D4C0, dp, o);
                o = mc3_fma(D4C1, sc, o);
                o = mc3_fma(-D4C2, dc, o);
                tmp[2 * i] = e;
                tmp[2 * i + 1] = o;
            }
            __syncthreads();
            for (int i = tid; i < nn; i += nt)
                a[i] = tmp[i];
            __syncthreads();
        }
    }
}

// ---- standalone likelihood kernels: the public functions of the reference's
// _chisq / _dwt extension modules on arbitrary model arrays --------------------
// residuals, src_c/_chisq.c:37-79: (model-data)/uncert, then the prior residuals
__global__ void residuals_kernel(const double *model, const double *data, const double *uncert,
                                 int n, const double *off, const double *plow,
                                 const double *pup, int np, double *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + np; i += gridDim.x * blockDim.x) {
        if (i < n) {
            out[i] = (model[i] - data[i]) / uncert[i];
        } else {
            const int j = i - n;
            out[i] = (off[j] > 0.0) ? off[j] / pup[j] : off[j] / plow[j];
        }
    }
}

// Sum of squares of v[0..n) with optional per-element weights given by scale
// (wavelet: weight = inv[floor(log2 j)] for j >= 2, 0 below): one workgroup of 1024
// threads, canonical order (strided partials over 1024 lanes, halving tree).
__global__ __launch_bounds__(1024) void sumsq_kernel(const double *v, int n, const double *inv,
                                                     double *out)
{
    __shared__ double s_part[1024];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += 1024) {
        if (inv == nullptr) {
            acc[0] = mc3_fma(v[i], v[i], acc[0]);
        } else if (i >= 2) {
            acc[0] = mc3_fma(v[i] * v[i], inv[31 - __clz(i)], acc[0]);
        }
    }
    const double t = lane_tree<16>(acc, s_part);
    if (threadIdx.x == 0)
        out[0] = t;
}

__global__ void diff_kernel(const double *a, const double *b, int n, int npad, double *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += gridDim.x * blockDim.x)
        out[i] = (i < n) ? a[i] - b[i] : 0.0;
}

// Per-chain mean and (population) variance of every free parameter over the rows a
// chain has written, skipping its first `burn` rows: the reductions the
// Gelman-Rubin test needs (mc3/stats/gelman.py:12-92) without copying Z to the host.
// One workgroup per chain; two passes (mean, then squared deviations).
__global__ __launch_bounds__(256) void chain_moments_kernel(const double *Z, long long M0,
                                                            int nchains, int nf, int zs,
                                                            long long nref, long long burn,
                                                            double *mean, double *var)
{
    __shared__ double s_red[256];
    const int c = blockIdx.x, t = threadIdx.x;
    const long long n = nref - burn;
    for (int k = 0; k < nf; ++k) {
        double acc = 0.0;
        for (long long r = burn + t; r < nref; r += 256)
            acc += Z[(size_t)(M0 + r * nchains + c) * zs + k];
        s_red[t] = acc;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2)
                s_red[t] += s_red[t + s2];
            __syncthreads();
        }
        const double mu = s_red[0] / (double)n;
        __syncthreads();
        acc = 0.0;
        for (long long r = burn + t; r < nref; r += 256) {
            const double d = Z[(size_t)(M0 + r * nchains + c) * zs + k] - mu;
            acc += d * d;
        }
        s_red[t] = acc;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2)
                s_red[t] += s_red[t + s2];
            __syncthreads();
        }
        if (t == 0) {
            mean[(size_t)c * nf + k] = mu;
            var[(size_t)c * nf + k] = s_red[0] / (double)n;
        }
        __syncthreads();
    }
}

// ---- posterior summary on the device (mc3/stats/stats.py:805-852,876-964) -----------------
// What update_output / calc_sample_statistics compute from the burned posterior with
// NumPy -- per-parameter mean, standard deviation, extrema, median and central percentiles
// -- without copying Z anywhere: two deterministic reduction passes and an exact
// multi-rank radix select over sortable 64-bit keys.
__device__ __forceinline__ uint64_t sortable_key(double v)
{
    const uint64_t b = mc3_bits(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);     // total order of finite doubles
}
__device__ __forceinline__ double key_to_double(uint64_t k)
{
    return mc3_from_bits((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k);
}

// pass 1: per block and parameter, sum / min / max of rows [row0, row0 + n); also the
// sortable keys, dense per parameter (keys[k][i]), for the selection passes
__global__ __launch_bounds__(256) void summary_pass1_kernel(const double *Z, long long row0,
                                                            long long n, int nf, int zs,
                                                            uint64_t *keys, double *part)
{
    __shared__ double s_sum[256], s_min[256], s_max[256];
    const int t = threadIdx.x;
    for (int k = 0; k < nf; ++k) {
        double sum = 0.0, lo = INFINITY, hi = -INFINITY;
        for (long long i = (long long)blockIdx.x * 256 + t; i < n; i += (long long)gridDim.x * 256) {
            const double v = Z[(size_t)(row0 + i) * zs + k];
            keys[(size_t)k * n + i] = sortable_key(v);
            sum += v;
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
        s_sum[t] = sum; s_min[t] = lo; s_max[t] = hi;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2) {
                s_sum[t] += s_sum[t + s2];
                s_min[t] = s_min[t + s2] < s_min[t] ? s_min[t + s2] : s_min[t];
                s_max[t] = s_max[t + s2] > s_max[t] ? s_max[t + s2] : s_max[t];
            }
            __syncthreads();
        }
        if (t == 0) {
            double *o = part + ((size_t)k * gridDim.x + blockIdx.x) * 4;
            o[0] = s_sum[0]; o[1] = s_min[0]; o[2] = s_max[0];
        }
        __syncthreads();
    }
}

// between the passes: block partials -> mean / min / max per parameter (one block)
__global__ __launch_bounds__(256) void summary_finish1_kernel(const double *part, int nblocks,
                                                              int nf, long long n, double *res)
{
    __shared__ double s_sum[256], s_min[256], s_max[256];
    const int t = threadIdx.x;
    for (int k = 0; k < nf; ++k) {
        double sum = 0.0, lo = INFINITY, hi = -INFINITY;
        for (int b = t; b < nblocks; b += 256) {
            const double *o = part + ((size_t)k * nblocks + b) * 4;
            sum += o[0];
            lo = o[1] < lo ? o[1] : lo;
            hi = o[2] > hi ? o[2] : hi;
        }
        s_sum[t] = sum; s_min[t] = lo; s_max[t] = hi;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2) {
                s_sum[t] += s_sum[t + s2];
                s_min[t] = s_min[t + s2] < s_min[t] ? s_min[t + s2] : s_min[t];
                s_max[t] = s_max[t + s2] > s_max[t] ? s_max[t + s2] : s_max[t];
            }
            __syncthreads();
        }
        if (t == 0) {
            res[4 * k] = s_sum[0] / (double)n;          // mean
            res[4 * k + 2] = s_min[0];
            res[4 * k + 3] = s_max[0];
        }
        __syncthreads();
    }
}

// pass 2: sum of squared deviations from the mean (np.var's second pass)
__global__ __launch_bounds__(256) void summary_pass2_kernel(const double *Z, long long row0,
                                                            long long n, int nf, int zs,
                                                            const double *res, double *part)
{
    __shared__ double s_sum[256];
    const int t = threadIdx.x;
    for (int k = 0; k < nf; ++k) {
        const double mu = res[4 * k];
        double sum = 0.0;
        for (long long i = (long long)blockIdx.x * 256 + t; i < n; i += (long long)gridDim.x * 256) {
            const double d = Z[(size_t)(row0 + i) * zs + k] - mu;
            sum += d * d;
        }
        s_sum[t] = sum;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2)
                s_sum[t] += s_sum[t + s2];
            __syncthreads();
        }
        if (t == 0)
            part[((size_t)k * gridDim.x + blockIdx.x) * 4 + 3] = s_sum[0];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void summary_finish2_kernel(const double *part, int nblocks,
                                                              int nf, long long n, double *res)
{
    __shared__ double s_sum[256];
    const int t = threadIdx.x;
    for (int k = 0; k < nf; ++k) {
        double sum = 0.0;
        for (int b = t; b < nblocks; b += 256)
            sum += part[((size_t)k * nblocks + b) * 4 + 3];
        s_sum[t] = sum;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2)
                s_sum[t] += s_sum[t + s2];
            __syncthreads();
        }
        if (t == 0)
            res[4 * k + 1] = s_sum[0] / (double)n;      // population variance
        __syncthreads();
    }
}

// Exact order statistics by radix selection, 8 bits per pass from the top: every
// (parameter, rank) pair carries the key prefix fixed so far and the rank left inside it;
// a pass histograms the next byte of the keys that share a pair's prefix (LDS, then one
// atomic flush per block), a one-block kernel picks each pair's byte.  MAXR ranks at most.
constexpr int SEL_MAXR = 8;
struct SelectState {
    uint64_t prefix[SEL_MAXR];      // key bits fixed so far (high part), rest zero
    long long rank[SEL_MAXR];       // rank inside the keys that share the prefix
};

__global__ __launch_bounds__(256) void select_hist_kernel(const uint64_t *keys, long long n, int nr,
                                                          int pass, const SelectState *st,
                                                          unsigned int *hist)
{
    __shared__ unsigned int s_h[SEL_MAXR * 256];
    const int k = blockIdx.y, t = threadIdx.x;
    for (int q = t; q < nr * 256; q += 256)
        s_h[q] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const uint64_t himask = pass ? ~0ull << (shift + 8) : 0ull;
    uint64_t pre[SEL_MAXR];
    bool first[SEL_MAXR];           // pairs that share a prefix share the count: the first adds
    for (int r = 0; r < nr; ++r) {
        pre[r] = st[k].prefix[r] & himask;
        first[r] = true;
        for (int r2 = 0; r2 < r; ++r2)
            first[r] = first[r] && pre[r2] != pre[r];
    }
    const uint64_t *col = keys + (size_t)k * n;
    for (long long i = (long long)blockIdx.x * 256 + t; i < n; i += (long long)gridDim.x * 256) {
        const uint64_t key = col[i];
        const unsigned int digit = (unsigned int)(key >> shift) & 255u;
        for (int r = 0; r < nr; ++r)
            if (first[r] && (key & himask) == pre[r])
                atomicAdd(&s_h[r * 256 + digit], 1u);
    }
    __syncthreads();
    for (int q = t; q < nr * 256; q += 256)
        if (s_h[q])
            atomicAdd(&hist[(size_t)k * SEL_MAXR * 256 + q], s_h[q]);
}

__global__ __launch_bounds__(256) void select_pick_kernel(int nr, int pass, SelectState *st,
                                                          unsigned int *hist)
{
    const int k = blockIdx.x, t = threadIdx.x;
    const int shift = 56 - 8 * pass;
    const uint64_t himask = pass ? ~0ull << (shift + 8) : 0ull;
    __shared__ unsigned int s_h[SEL_MAXR * 256];
    unsigned int *h = hist + (size_t)k * SEL_MAXR * 256;
    // a pair without counts of its own reads those of the first pair with its prefix
    for (int q = t; q < nr * 256; q += 256) {
        const int r = q >> 8;
        int owner = r;
        for (int r2 = r - 1; r2 >= 0; --r2)
            if ((st[k].prefix[r2] & himask) == (st[k].prefix[r] & himask))
                owner = r2;
        s_h[q] = h[owner * 256 + (q & 255)];
    }
    __syncthreads();
    if (t < nr) {
        long long left = st[k].rank[t];
        unsigned int d = 0;
        for (; d < 255; ++d) {
            const unsigned int cnt = s_h[t * 256 + d];
            if (left < (long long)cnt)
                break;
            left -= cnt;
        }
        st[k].rank[t] = left;
        st[k].prefix[t] = (st[k].prefix[t] & himask) | ((uint64_t)d << shift);
    }
    __syncthreads();
    for (int q = t; q < SEL_MAXR * 256; q += 256)
        h[q] = 0;
}

__global__ void select_result_kernel(const SelectState *st, int nf, int nr, double *out)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nf * nr)
        out[q] = key_to_double(st[q / nr].prefix[q % nr]);      // [nf][nr]
}

// chisq = -2 (log_post - log_prior) of rows [row0, row0 + n) (mc3/stats/stats.py:821 with
// log_prior of :287-392: Gaussian priors with two-sided widths where priorlow, priorup > 0;
// the reference's log-uniform branch where priorlow < 0), and the packed dense copy of the
// rows' free parameters -- everything update_output needs per row, in one pass
__global__ void rows_chisq_kernel(const DevProblem P, long long row0, long long n, double *chisq)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const double *z = P.Z + (size_t)(row0 + i) * P.zs;
        double s = 0.0;
        for (int k = 0; k < P.nfree; ++k) {
            const int j = P.ifree[k];
            double d = 0.0;
            if (P.priorlow[j] > 0.0 && P.priorup[j] > 0.0) {
                const double off = z[k] - P.prior[j];
                d = (off == 0.0) ? 0.0 : off / (off < 0.0 ? P.priorlow[j] : P.priorup[j]);
            } else if (P.priorlow[j] < 0.0) {
                d = 2.0 * mc3_det_log(z[k]);
            }
            s = s + d * d;
        }
        chisq[i] = -2.0 * (P.log_post[row0 + i] - (-0.5 * s));
    }
}

// Gaussian kernel density of a 1D sample at m points (scipy.stats.gaussian_kde.evaluate,
// called by cred_region, mc3/stats/stats.py:395-467): pdf[j] = norm * sum_i exp(-(x_j -
// s_i)^2 / (2 cov)).  One block per evaluation point.
__global__ __launch_bounds__(256) void kde_kernel(const double *sample, int n, const double *x,
                                                  double inv2cov, double norm, double *pdf)
{
    __shared__ double s_sum[256];
    const int t = threadIdx.x;
    const double xj = x[blockIdx.x];
    double sum = 0.0;
    for (int i = t; i < n; i += 256) {
        const double d = xj - sample[i];
        sum += mc3_det_exp(-(d * d) * inv2cov);
    }
    s_sum[t] = sum;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) {
        if (t < s2)
            s_sum[t] += s_sum[t + s2];
        __syncthreads();
    }
    if (t == 0)
        pdf[blockIdx.x] = s_sum[0] * norm;
}

// dense [rows][nf] (what the C-ABI exchanges) <-> padded [rows][zs] (device layout of Z)
__global__ void pack_rows_kernel(double *dense, const double *padded, long long nrows, int nf,
                                 int zs)
{
    const long long total = nrows * nf;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        dense[i] = padded[(i / nf) * zs + (i % nf)];
}

__global__ void unpack_rows_kernel(double *padded, const double *dense, long long nrows, int nf,
                                   int zs)
{
    const long long total = nrows * nf;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        padded[(i / nf) * zs + (i % nf)] = dense[i];
}

// ---- initial population on the device (mc3/mcmc_driver.py:229-270) -------------------
// Trial t of the kick-off: nfree draws around the initial guess ('normal': params +
// pstep * N(0,1), Box-Muller pairs; 'uniform': between the boundaries), keyed like every
// other draw by Philox(seed; counter = (t, chain = 0xffffffff, slot)) -- a chain id no walker
// has.  One thread per trial writes the full parameter vector (fixed values kept, shared
// parameters copied, chain.py:246-247) for the chisq kernel, the free parameters, and
// whether the FREE parameters are inside their boundaries.
constexpr uint32_t KICKOFF_STREAM = 0xffffffffu;

__global__ void init_trials_kernel(const DevProblem P, int kickoff_uniform, uint64_t seed,
                                   long long t0, int ntrial, double *full, double *freep,
                                   int32_t *inside)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntrial)
        return;
    const uint64_t t = (uint64_t)(t0 + i);
    double *row = full + (size_t)i * P.nparams;
    for (int j = 0; j < P.nparams; ++j)
        row[j] = P.params[j];
    int ok = 1;
    for (int k2 = 0; 2 * k2 < P.nfree; ++k2) {
        const Draw2 d = draw2(seed, t, KICKOFF_STREAM, (uint32_t)k2);
        double a, b;
        if (kickoff_uniform) {
            a = u01(d.q0);
            b = u01(d.q1);
        } else {
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(d.q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(d.q1), &sn, &cs);
            a = rad * cs;
            b = rad * sn;
        }
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * k2 + h;
            if (k >= P.nfree)
                break;
            const double r = h ? b : a;
            const int j = P.ifree[k];
            const double v = kickoff_uniform ? P.fmin[k] + (P.fmax[k] - P.fmin[k]) * r
                                             : P.params[j] + P.fstep[k] * r;
            row[j] = v;
            freep[(size_t)i * P.nfree + k] = v;
            ok &= !(v < P.fmin[k] || v > P.fmax[k]);
        }
    }
    for (int sidx = 0; sidx < P.nshare; ++sidx)
        row[P.ishare[sidx]] = row[P.share_src[sidx]];
    inside[i] = ok;
}

// The reference's acceptance rule, in trial order (one thread: the rule is sequential and
// the work is a few operations per trial).  A trial counts as a failure when a parameter
// lies outside its boundaries -- tested on the full vector BEFORE the shared parameters are
// refreshed, i.e. with the copies the last in-bounds trial left (mcmc_driver.py:249-255) --
// or when its chisq is not finite; the loop ends with M0 rows or after 100 M0 failures.
// st[0] = rows filled, st[1] = failures, st[2] = trials consumed; stale[] = the shared
// parameters' current copies.
__global__ void init_scan_kernel(const DevProblem P, int ntrial, const double *full,
                                 const double *freep, const int32_t *inside,
                                 const double *chisq, long long *st, double *stale,
                                 double *Z0, double *lp0)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    long long filled = st[0], failed = st[1], used = st[2];
    const long long M0 = P.M0, nmax = 100 * P.M0;
    for (int i = 0; i < ntrial; ++i) {
        if (filled == M0 || failed == nmax)
            break;
        used += 1;
        bool ok = inside[i] != 0;
        for (int sidx = 0; sidx < P.nshare && ok; ++sidx) {
            const int j = P.ishare[sidx];
            ok = !(stale[sidx] > P.pmax[j] || stale[sidx] < P.pmin[j]);
        }
        if (!ok) {
            failed += 1;
            continue;
        }
        for (int sidx = 0; sidx < P.nshare; ++sidx)
            stale[sidx] = full[(size_t)i * P.nparams + P.share_src[sidx]];
        const double lp = -0.5 * chisq[i];
        if ((mc3_bits(lp) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull) {
            failed += 1;                                  // inf or NaN
            continue;
        }
        for (int k = 0; k < P.nfree; ++k)
            Z0[(size_t)filled * P.nfree + k] = freep[(size_t)i * P.nfree + k];
        lp0[filled] = lp;
        filled += 1;
    }
    st[0] = filled;
    st[1] = failed;
    st[2] = used;
}

// per-chain state to / from the checkpoint area (8-byte words; the seven items back to back
// in `ckpt`): one launch instead of seven asynchronous device-to-device copies per run
struct CheckpointItems {
    unsigned long long *ptr[7];
    unsigned long long words[7];
};

__global__ void checkpoint_kernel(CheckpointItems it, unsigned long long *ckpt, int save)
{
    unsigned long long at = 0;
    for (int k = 0; k < 7; ++k) {
        for (unsigned long long i = blockIdx.x * blockDim.x + threadIdx.x; i < it.words[k];
             i += (unsigned long long)gridDim.x * blockDim.x) {
            if (save)
                ckpt[at + i] = it.ptr[k][i];
            else
                it.ptr[k][i] = ckpt[at + i];
        }
        at += it.words[k];
    }
}

__global__ void set_ctl_kernel(long long *ctl, long long base, long long run0)
{
    ctl[0] = base;
    if (run0 >= 0)
        ctl[1] = run0;      // ctl[CTL_ABORT] is only ever cleared by the host (settle)
}

__global__ void selftest_math_kernel(int op, const double *in, int n, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double x = in[i];
    double r, s, c;
    switch (op) {
    case 0: r = mc3_det_exp(x); break;
    case 1: r = mc3_det_log(x); break;
    case 2: r = mc3_det_exp2(x); break;
    case 3: mc3_det_sincos2pi(x, &s, &c); r = s; break;
    case 4: mc3_det_sincos2pi(x, &s, &c); r = c; break;
    case 5: r = __builtin_sqrt(x); break;
    default: r = 1.0 / x; break;
    }
    out[i] = r;
}

__global__ void selftest_philox_kernel(const uint32_t *ctr, int n, uint32_t k0, uint32_t k1,
                                       uint32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint32_t w[4];
    philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], k0, k1, w);
    for (int q = 0; q < 4; ++q)
        out[4 * i + q] = w[q];
}

}  // namespace mc3hip
