/*
 * oracle/oracle_lockstep.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of mc3's per-iteration chain update (mc3/chain.py:183-299)
 * in the LOCKSTEP form the HIP kernels implement: all chains advance together
 * from a snapshot of the population taken at the last Z refresh, and every
 * random number is a pure function Philox4x32-10(seed; iteration, chain, slot).
 * The proposal / bounds / Metropolis / Z-append rules are the reference's, line
 * by line (citations below); what differs from the reference is only (i) the
 * generator (the reference uses NumPy's MT19937 stream, chain.py:185-259) and
 * (ii) Jacobi instead of Gauss-Seidel visibility of the other chains
 * (SURVEY.md 3.3-12).
 *
 * ONE step function, two drivers.  chain_step() below is the only statement of the
 * step rules; it takes its random numbers from a draw source (ls_draws) and sees
 * the other chains through whatever the driver hands it:
 *   ls_run()       lockstep driver -- Philox draws, population snapshot of the last
 *                  refresh (what the HIP kernels do; the GPU's bit-exact checker);
 *   ls_run_tape()  the reference's own sequential loop (chain.py:183-299, one
 *                  worker) -- draws read from a TAPE recorded from the reference's
 *                  np.random calls, chains updated one after another so that each
 *                  sees the previous one's new state.  Replaying the reference runs
 *                  of tests/golden/seq_*.npz through it must reproduce them (Z rows,
 *                  accept and out-of-bounds counters, best fit: identical; chisq to
 *                  1e-10): tests/test_oracle_tape.py.  That pins the very function
 *                  that checks the GPU to the reference, not a sibling of it.
 *
 * Arithmetic contract (DESIGN.md "Arithmetic contract"): built with
 * -ffp-contract=off; a fused multiply-add happens only where written
 * (model Horner/dot, the squared-residual accumulation, the D4 filter taps,
 * and inside include/mc3hip_detmath.h).  Sums over data points follow the
 * canonical order "tiles in order; inside a tile `vlanes` strided partial sums,
 * then a halving tree".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's library.
 */
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "mc3hip_detmath.h"

enum { LS_MRW = 0, LS_DEMC = 1, LS_SNOOKER = 2 };
enum { LS_POLY = 0, LS_LINEAR = 1 };

typedef struct {
    int32_t nchains, chain0, nlocal;
    int32_t nparams, nfree, nshare, ndata;
    int32_t sampler, wlike, model_kind, nmodel, thinning;
    int32_t ngprior;
    int32_t tile, vlanes;
    int32_t wl_threads;     /* wavelet path: threads that share a chain (64 * waves) */
    int32_t tile_tree;      /* 1: 64 partial sums at a time, then the group totals */
    int32_t zlag;           /* declared cadence of the multi-GPU runs: the other chains are
                             * seen `zlag` refreshes later -- DEMC partners of refresh R are the
                             * rows of refresh R-1-zlag (the initial snapshot before there is
                             * one), snooker's history ends with refresh R-1-zlag; 0 = strict */
    int32_t wl_nreal;       /* wavelet likelihood of a series padded to ndata = 2^M points: how
                             * many of them are data (0: all) -- the residuals of the padding
                             * are exactly zero (include/mc3hip.h mc3hip_problem.wl_nreal) */
    int32_t reserved1;
    int64_t M0, zlen;
    uint64_t seed;
    double gamma, fepsilon;
    const int32_t *ifree, *ishare, *share_src, *igprior;
    const double *params, *pmin, *pmax, *pstep, *prior, *priorlow, *priorup;
    const double *indep, *data, *w;
} ls_problem;

typedef struct {
    double *Z, *log_post;
    int32_t *zchain;
    double *X, *chisq;
    int64_t *numaccept, *outbounds;     /* [nchains], [nfree] */
    double *best_chisq;
    int64_t *best_iter;
    double *bestx;
    double *trace_chisq;                /* optional [niter][nchains] */
    int32_t *trace_flag;                /* bit0 in-bounds, bit1 accepted, bit2 snooker jump */
    const double *snap0;                /* population seen by refresh 0 */
    int64_t refresh_done;
} ls_state;

/* ---- Philox4x32-10 (Salmon et al. 2011), written out independently of the
 * device implementation in mc3_amd/csrc/philox.h; both are pinned to the
 * Random123 known-answer vectors in tests/test_philox.py. */
void ls_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* two 64-bit draws for (iteration, chain, slot) */
static void draw2(uint64_t seed, uint64_t iter, uint32_t chain, uint32_t slot,
                  uint64_t *q0, uint64_t *q1)
{
    const uint32_t ctr[4] = { (uint32_t)iter, (uint32_t)(iter >> 32), chain, slot };
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t w[4];
    ls_philox4x32_10(ctr, key, w);
    *q0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    *q1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
}

static double u01(uint64_t q) { return (double)(q >> 11) * 0x1.0p-53; }          /* [0,1) */
static double u01_open(uint64_t q) { return (double)((q >> 11) + 1) * 0x1.0p-53; } /* (0,1] */
static uint64_t bounded(uint64_t q, uint64_t n)                                   /* [0,n) */
{
    return (uint64_t)(((unsigned __int128)q * n) >> 64);
}
static uint32_t bounded32(uint32_t w, uint32_t n)                                 /* [0,n) */
{
    return (uint32_t)(((uint64_t)w * n) >> 32);
}

/* ---- draw source ------------------------------------------------------------
 * What the step rules ask for, in the reference's call order:
 *   index(lo, hi)    np.random.randint(lo, hi)        integer in [lo, hi)
 *   uniform(lo, hi)  np.random.uniform(lo, hi)        lo + (hi-lo)*[0,1)
 * Philox mode computes each from (seed; iteration, chain, slot) -- a pure function,
 * so the order of the requests does not matter; tape mode hands out the values the
 * reference's generator produced, in order, and checks that the KIND of every
 * request is the kind that was recorded (a step rule that asked for a different
 * draw sequence than the reference did fails right there). */
enum { DR_IR1, DR_IR2, DR_SJUMP, DR_IZ, DR_GS, DR_R1, DR_R2, DR_METRO };
enum { TK_NORMAL = 0, TK_INDEX = 1, TK_UNIFORM = 2 };

typedef struct {
    int tape_mode;
    /* Philox */
    uint64_t seed, iter;
    uint32_t chain;
    int sampler;
    /* tape */
    const double *tape;
    const int8_t *kind;
    int64_t pos, len;
    int error;                 /* 1 = ran off the tape, 2 = kind mismatch */
    /* dot products of the snooker projection: the reference calls np.dot, i.e. the BLAS
     * ddot of the box the goldens were recorded on, whose scalar tail contracts
     * multiply-add; the lockstep contract rounds every product */
    int fma_dot;
} ls_draws;

static double tape_next(ls_draws *D, int kind)
{
    if (D->pos >= D->len) {
        D->error = D->error ? D->error : 1;
        return 0.0;
    }
    if (D->kind[D->pos] != kind)
        D->error = D->error ? D->error : 2;
    return D->tape[D->pos++];
}

static int64_t draw_index(ls_draws *D, int id, int64_t lo, int64_t hi)
{
    if (D->tape_mode)
        return (int64_t)tape_next(D, TK_INDEX);
    uint64_t q0, q1;
    draw2(D->seed, D->iter, D->chain, (id == DR_IZ) ? 1u : 0u, &q0, &q1);
    switch (id) {
    case DR_IR1: return lo + (int64_t)bounded(q0, (uint64_t)(hi - lo));
    case DR_IR2: return lo + (int64_t)bounded(q1, (uint64_t)(hi - lo));
    case DR_IZ:  return lo + (int64_t)bounded(q1, (uint64_t)(hi - lo));
    case DR_R1:  return lo + (int64_t)bounded32((uint32_t)(q0 >> 32), (uint32_t)(hi - lo));
    default:     return lo + (int64_t)bounded32((uint32_t)q0, (uint32_t)(hi - lo));   /* DR_R2 */
    }
}

static double draw_uniform(ls_draws *D, int id, double lo, double hi)
{
    if (D->tape_mode)
        return tape_next(D, TK_UNIFORM);
    uint64_t q0, q1;
    double u;
    switch (id) {
    case DR_SJUMP: draw2(D->seed, D->iter, D->chain, 1u, &q0, &q1); u = u01(q0); break;
    case DR_GS:    draw2(D->seed, D->iter, D->chain, 2u, &q0, &q1); u = u01(q0); break;
    default:       /* DR_METRO: second word of slot 2 (snooker) or of slot 0 */
        draw2(D->seed, D->iter, D->chain, (D->sampler == LS_SNOOKER) ? 2u : 0u, &q0, &q1);
        u = u01(q1);
        break;
    }
    if (lo == 0.0 && hi == 1.0)
        return u;
    return lo + (hi - lo) * u;
}

/* support-distribution draw N(0, pstep[ifree]), chain.py:185.  Philox: one vector per
 * chain and iteration (Box-Muller on slots 3+); tape: the next nfree recorded values
 * (the reference draws ONE vector per worker-iteration, shared by its chains). */
static void draw_normal(ls_draws *D, const ls_problem *P, double *normal)
{
    const int nf = P->nfree;
    if (D->tape_mode) {
        for (int k = 0; k < nf; k++)
            normal[k] = tape_next(D, TK_NORMAL);
        return;
    }
    uint64_t q0, q1;
    for (int j = 0; 2 * j < nf; j++) {
        draw2(D->seed, D->iter, D->chain, (uint32_t)(3 + j), &q0, &q1);
        const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(q0)));
        double sn, cs;
        mc3_det_sincos2pi(u01(q1), &sn, &cs);
        normal[2*j] = (rad * cs) * P->pstep[P->ifree[2*j]];
        if (2*j + 1 < nf)
            normal[2*j + 1] = (rad * sn) * P->pstep[P->ifree[2*j + 1]];
    }
}

static double dotn(const ls_draws *D, const double *a, const double *b, int n)
{
    double s = 0.0;                                     /* sum_k a[k] b[k], in index order */
    for (int k = 0; k < n; k++)
        s = D->fma_dot ? mc3_fma(a[k], b[k], s) : s + a[k] * b[k];
    return s;
}

/* ---- model ---------------------------------------------------------------- */
static inline double model_at(const ls_problem *P, const double *p, int i)
{
    if (P->model_kind == LS_POLY) {
        const double x = P->indep[i];
        double m = p[P->nmodel - 1];
        for (int j = P->nmodel - 2; j >= 0; j--)
            m = mc3_fma(m, x, p[j]);
        return m;
    }
    const double *A = P->indep;                 /* column k at A + k*ndata */
    double m = A[i] * p[0];
    for (int k = 1; k < P->nmodel; k++)
        m = mc3_fma(A[(size_t)k * P->ndata + i], p[k], m);
    return m;
}

void ls_eval_model(const ls_problem *P, const double *p, double *out)
{
    for (int i = 0; i < P->ndata; i++)
        out[i] = model_at(P, p, i);
}

/* prior term: mc3/stats/stats.py:211-216 + src_c/include/stats.h:96-106 */
static double prior_term(const ls_problem *P, const double *p)
{
    double pr = 0.0;
    for (int g = 0; g < P->ngprior; g++) {
        const int j = P->igprior[g];
        const double d = p[j] - P->prior[j];
        const double s = (d > 0.0) ? d / P->priorup[j] : d / P->priorlow[j];
        pr = pr + s * s;
    }
    return pr;
}

static double halving_tree(double *part, int n)
{
    for (int s = n / 2; s >= 1; s /= 2)
        for (int l = 0; l < s; l++)
            part[l] = part[l] + part[l + s];
    return part[0];
}

/* the `vlanes` strided partial sums of a tile: one halving tree over all of them, or
 * ("waves first", the matrix-core tile kernel) groups of 64, then the group totals */
static double tile_sum(double *part, int n, int waves_first)
{
    if (!waves_first || n <= 64)
        return halving_tree(part, n);
    const int ng = n / 64;
    for (int w = 0; w < ng; w++)
        part[w] = halving_tree(part + 64 * w, 64);     /* (w <= 64 w: no overlap) */
    return halving_tree(part, ng);
}

/* The tiled plan's per-tile sums of a chain: sixty-four at a time as a halving tree (a lane of
 * the owner wave per tile, lanes beyond the tiles 0.0), the groups of sixty-four added in
 * order (csrc/chisq.h finish_chisq, csrc/resident.h ResidentOwner::collect). */
static double tiles_sum(const double *tt, int nt)
{
    double total = 0.0;
    for (int t0 = 0; t0 < nt; t0 += 64) {
        double g[64];
        for (int l = 0; l < 64; l++)
            g[l] = (t0 + l < nt) ? tt[t0 + l] : 0.0;
        total = total + halving_tree(g, 64);
    }
    return total;
}

/* tile_tree = 2 (design matrices through the matrix cores, tile = 2048 points): the weights
 * are folded into the operands -- column k of the design matrix and the data are multiplied
 * by w = 1/uncert ONCE, each product rounded, when the problem is set up -- so a residual is
 *   r_i = fma(-1, (d w)_i, fma((A_{K-1} w)_i, p_{K-1}, ... (A_0 w)_i p_0))
 * (what v_mfma_f64_16x16x4_f64 computes with the weighted data riding as operand row K with
 * coefficient -1: its products are added in k order with one rounding each), squared and
 * accumulated per lane over the lane's sixteen points of the tile in step order; then the 16
 * lanes of a row (halving tree, strides 8..1), then the tile's eight rows (4..1).  Point of
 * lane j, row wv, step st (block vb = st / 2, half h = st % 2): 1024 h + 128 wv + 16 vb + j.
 * The +inf-model trap (chain.py:322-323) is resolved as the kernels do: only a non-finite
 * total has the (unweighted) model values scanned. */
static double gauss_chisq_weighted_rows(const ls_problem *P, const double *p)
{
    const int tile = P->tile, K = P->nmodel;
    const int npad = (P->ndata + tile - 1) / tile * tile;
    const double *A = P->indep;
    const int nt = npad / tile;
    double *tt = (double *)malloc((size_t)nt * sizeof(double));
    for (int t0 = 0; t0 < npad; t0 += tile) {
        double rows[8];
        for (int wv = 0; wv < 8; wv++) {
            double lanes[16];
            for (int j = 0; j < 16; j++) {
                double acc = 0.0;
                for (int st = 0; st < 16; st++) {
                    const int i = t0 + 1024 * (st & 1) + 128 * wv + 16 * (st >> 1) + j;
                    const int ii = (i < P->ndata) ? i : P->ndata - 1;
                    const double wi = (i < P->ndata) ? P->w[ii] : 0.0;
                    double m = (A[ii] * wi) * p[0];
                    for (int k = 1; k < K; k++)
                        m = mc3_fma(A[(size_t)k * P->ndata + ii] * wi, p[k], m);
                    const double r = mc3_fma(P->data[ii] * wi, -1.0, m);
                    acc = mc3_fma(r, r, acc);
                }
                lanes[j] = acc;
            }
            rows[wv] = halving_tree(lanes, 16);
        }
        tt[t0 / tile] = halving_tree(rows, 8);
    }
    const double total = tiles_sum(tt, nt);
    free(tt);
    if (!(total - total == 0.0)) {                        /* inf or NaN: was a model value +inf? */
        for (int i = 0; i < P->ndata; i++)
            if (model_at(P, p, i) == INFINITY)
                return INFINITY;
    }
    if (total == INFINITY)
        return INFINITY;
    return total + prior_term(P, p);
}

/* Gaussian chi-squared: src_c/_chisq.c:131-137 with w = 1/uncert, in the
 * canonical summation order.  model == +inf anywhere => +inf (chain.py:322). */
static double gauss_chisq(const ls_problem *P, const double *p, double *part)
{
    if (P->tile_tree == 2 && P->tile == 2048 && P->model_kind != LS_POLY)
        return gauss_chisq_weighted_rows(P, p);
    const int VL = P->vlanes;
    /* The kernels pad the data to a multiple of `vlanes` points that repeat the
     * last point with weight 0: a padded point adds fma(r, r, acc) with
     * r = (model_last - data_last) * 0.  Spelled out here so that even the
     * non-finite corner (model = -inf or NaN) is mirrored exactly. */
    const int unit = (P->tile > 0) ? P->tile : VL;
    const int npad = (P->ndata + unit - 1) / unit * unit;
    const int tile = (P->tile > 0) ? P->tile : npad;
    int sawinf = 0;
    double total = 0.0;
    /* (the tiled plan, P->tile > 0: the tiles' sums as tiles_sum; otherwise there is one) */
    const int nt = npad / tile;
    double *tt = (P->tile > 0) ? (double *)malloc((size_t)nt * sizeof(double)) : NULL;
    for (int t0 = 0; t0 < npad; t0 += tile) {
        const int t1 = (t0 + tile < npad) ? t0 + tile : npad;
        for (int l = 0; l < VL; l++) {
            double acc = 0.0;
            for (int i = t0 + l; i < t1; i += VL) {
                const int ii = (i < P->ndata) ? i : P->ndata - 1;
                const double m = model_at(P, p, ii);
                if (m == INFINITY)
                    sawinf = 1;
                const double wi = (i < P->ndata) ? P->w[ii] : 0.0;
                const double r = (m - P->data[ii]) * wi;
                acc = mc3_fma(r, r, acc);
            }
            part[l] = acc;
        }
        if (tt)
            tt[t0 / tile] = tile_sum(part, VL, P->tile_tree);
        else
            total = total + tile_sum(part, VL, P->tile_tree);
    }
    if (tt) {
        total = tiles_sum(tt, nt);
        free(tt);
    }
    if (sawinf || total == INFINITY)
        return INFINITY;
    return total + prior_term(P, p);
}

/* Wavelet pseudo chi-squared: src_c/_dwt.c:76-114, wavelet.h:27-36,115-117.
 * ndata = 2^M.  Detail coefficients are consumed as they are produced (they
 * are final once written, wavelet.h:33,36); only the smooth half goes on to the
 * next level.  The pyramid is the reference's; what is canonical here is only the
 * ORDER in which the squared details are added up, which follows the blocked
 * layout of the kernel (wavelet.h block_wavelet) for T = wl_threads threads:
 *   pre    while more than 32 T values are left: output i of a level belongs to
 *          thread i mod T (large ndata only);
 *   A      Ta = min(T, n/2) threads own contiguous blocks of B = n/Ta values and run
 *          log2 B levels inside their block (halo: the next block's first two
 *          values, wrapping), leaving one value per thread;
 *   B      more than 64 values left: the same again with 64 lanes of one wave;
 *   C      at most 64 values left, one per lane: output i of a level belongs to
 *          lane i.
 * An owner adds the squares of its details of one level in index order,
 * acc = fma(d, d, acc), and folds the level in with the weight of its scale,
 * w = fma(acc, 1/sigma_W^2, w); the owners' w are summed by a halving tree per
 * stage, and the stages as (A + B) + C (pre shares A's accumulators). */
static const double D4C0 = 0.4829629131445341, D4C1 = 0.83651630373780772,
                    D4C2 = 0.22414386804201339, D4C3 = -0.12940952255126034;

/* threads per chain of the wavelet geometry: as given, or the library's default rule
 * (mc3hip.hip choose_wpc: one wave up to 2048 points, four at 4096, eight above) */
static int wl_threads_of(const ls_problem *P)
{
    if (P->wl_threads >= 64)
        return P->wl_threads;
    return P->ndata > 4096 ? 512 : (P->ndata > 2048 ? 256 : 64);
}

static int ilog2i(int n)
{
    int m = 0;
    while ((1 << m) < n)
        m++;
    return m;
}

/* one level nn -> nn/2 of a[] into b[]; detail i is added to acc[owner(i)] */
static void d4_level(const double *a, double *b, int nn, double *acc, int block, int stride_T)
{
    const int nh = nn >> 1;
    for (int i = 0; i < nh; i++) {
        const double a0 = a[2*i], a1 = a[2*i + 1];
        const double a2 = a[(2*i + 2) & (nn - 1)], a3 = a[(2*i + 3) & (nn - 1)];
        double s = D4C0 * a0;
        s = mc3_fma(D4C1, a1, s);
        s = mc3_fma(D4C2, a2, s);
        s = mc3_fma(D4C3, a3, s);
        double d = D4C3 * a0;
        d = mc3_fma(-D4C2, a1, d);
        d = mc3_fma(D4C1, a2, d);
        d = mc3_fma(-D4C0, a3, d);
        b[i] = s;
        const int owner = stride_T ? i % stride_T : i / block;
        acc[owner] = mc3_fma(d, d, acc[owner]);
    }
}

static double wavelet_chisq(const ls_problem *P, const double *p,
                            double *buf0, double *buf1, double *part)
{
    const int N = P->ndata;
    const int T = wl_threads_of(P);
    const int M = ilog2i(N);
    const double gam = p[P->nparams - 3];
    const double sr = p[P->nparams - 2];
    const double sw = p[P->nparams - 1];
    const double sr2 = sr * sr, sw2 = sw * sw;
    int sawinf = 0;
    const int nreal = (P->wl_nreal > 0 && P->wl_nreal < N) ? P->wl_nreal : N;
    for (int i = 0; i < N; i++) {
        const double m = model_at(P, p, i);
        if (m == INFINITY)
            sawinf = 1;
        buf0[i] = (i < nreal) ? P->data[i] - m : 0.0;      /* src_c/_dwt.c:89-91 */
    }
    if (sawinf)
        return INFINITY;
    double inv[32];
    for (int m = 1; m < M; m++)
        inv[m] = 1.0 / (sr2 * mc3_det_exp2(-gam * (double)m) + sw2);
    /* part[] holds 2 T doubles here: w[T] and acc[T] (scratch_alloc sizes it) */
    double *w = part, *acc = part + T;
    double *a = buf0, *b = buf1, *t;
    int nn = N;
    for (int l = 0; l < T; l++)
        w[l] = acc[l] = 0.0;
    /* pre: strided owners while the blocks would exceed 32 values per thread */
    while (nn > 32 * T) {
        d4_level(a, b, nn, acc, 0, T);
        const double iv = inv[ilog2i(nn >> 1)];
        for (int l = 0; l < T; l++) {
            w[l] = mc3_fma(acc[l], iv, w[l]);
            acc[l] = 0.0;
        }
        t = a; a = b; b = t;
        nn >>= 1;
    }
    /* A: contiguous blocks, levels inside the block */
    double sumA = 0.0, sumB = 0.0, sumC = 0.0;
    {
        const int Ta = (nn / 2 < T) ? (nn / 2 > 0 ? nn / 2 : 1) : T;
        for (int n = nn / Ta; n >= 2 && nn >= 4; n >>= 1) {
            d4_level(a, b, nn, acc, n / 2, 0);
            const double iv = inv[ilog2i(nn >> 1)];
            for (int l = 0; l < Ta; l++) {
                w[l] = mc3_fma(acc[l], iv, w[l]);
                acc[l] = 0.0;
            }
            t = a; a = b; b = t;
            nn >>= 1;
        }
        sumA = halving_tree(w, T);
    }
    /* B: one wave, blocks of nn/64 */
    if (nn > 64) {
        for (int l = 0; l < 64; l++)
            w[l] = acc[l] = 0.0;
        for (int n = nn / 64; n >= 2 && nn >= 4; n >>= 1) {
            d4_level(a, b, nn, acc, n / 2, 0);
            const double iv = inv[ilog2i(nn >> 1)];
            for (int l = 0; l < 64; l++) {
                w[l] = mc3_fma(acc[l], iv, w[l]);
                acc[l] = 0.0;
            }
            t = a; a = b; b = t;
            nn >>= 1;
        }
        sumB = halving_tree(w, 64);
    }
    /* C: one value per lane */
    for (int l = 0; l < 64; l++)
        w[l] = acc[l] = 0.0;
    while (nn >= 4) {
        d4_level(a, b, nn, acc, 1, 0);
        const double iv = inv[ilog2i(nn >> 1)];
        for (int l = 0; l < 64; l++) {
            w[l] = mc3_fma(acc[l], iv, w[l]);
            acc[l] = 0.0;
        }
        t = a; a = b; b = t;
        nn >>= 1;
    }
    sumC = halving_tree(w, 64);
    const double Tsum = (sumA + sumB) + sumC;
    const double sS2 = sr2 * mc3_det_exp2(-gam) * 0.72134752 + sw2;
    const double S = (a[0] * a[0]) / sS2 + (a[1] * a[1]) / sS2;
    double Lg = 2.0 * mc3_det_log(6.283185307179586 * sS2);
    for (int m = 1; m < M; m++) {
        const double sW2 = sr2 * mc3_det_exp2(-gam * (double)m) + sw2;
        Lg = Lg + (double)(1 << m) * mc3_det_log(6.283185307179586 * sW2);
    }
    const double c = (S + Lg) + Tsum;
    if (c == INFINITY)
        return c;
    return c + prior_term(P, p);
}

typedef struct {
    double *part, *buf0, *buf1, *nextp, *jump, *normal, *dz;
} ls_scratch;

static size_t part_doubles(const ls_problem *P)
{
    const size_t T = (size_t)wl_threads_of(P);
    const size_t n = 2 * T > 128 ? 2 * T : 128;
    return n > (size_t)P->vlanes ? n : (size_t)P->vlanes;
}

static double eval_chisq(const ls_problem *P, const double *p, ls_scratch *S)
{
    if (P->wlike)
        return wavelet_chisq(P, p, S->buf0, S->buf1, S->part);
    return gauss_chisq(P, p, S->part);
}

/* chisq of an arbitrary parameter vector (initial population, best-fit statistics,
 * Chain.eval_model): the library evaluates those with its one-workgroup-per-vector kernel
 * whatever plan the chain update runs under, i.e. in the untiled order ("vlanes strided
 * partial sums over all the data, one halving tree").  Under a tiled plan this differs from
 * the chain's own sums by rounding only; the checker mirrors it so that whole sample() runs
 * can be compared bit for bit (tests/test_gpu_api.py). */
static double eval_chisq_untiled(const ls_problem *P, const double *p, ls_scratch *S)
{
    if (P->wlike || P->tile == 0)
        return eval_chisq(P, p, S);
    ls_problem Q = *P;
    Q.tile = 0;
    Q.tile_tree = 0;
    return eval_chisq(&Q, p, S);
}

/* chisq for n parameter vectors (rows of `params`, nparams each) */
int ls_eval_chisq(const ls_problem *P, const double *params, int n, double *out)
{
    ls_scratch S;
    memset(&S, 0, sizeof S);
    S.part = malloc(sizeof(double) * part_doubles(P));
    S.buf0 = malloc(sizeof(double) * (size_t)P->ndata);
    S.buf1 = malloc(sizeof(double) * (size_t)P->ndata);
    for (int r = 0; r < n; r++)
        out[r] = eval_chisq_untiled(P, params + (size_t)r * P->nparams, &S);
    free(S.part); free(S.buf0); free(S.buf1);
    return 0;
}

/* ---- one chain, one iteration: mc3/chain.py:190-274 -----------------------
 * x        this chain's current free parameters (updated in place on acceptance)
 * normal   the support-distribution draw of this iteration (chain.py:185)
 * pop      what DEMC sees of the other chains, [nchains][nfree]: the snapshot of the
 *          last refresh (lockstep) or the live states (sequential, chain.py:230-232)
 * zsize    rows of Z visible to snooker (chain.py:197-203) */
static void chain_step(const ls_problem *P, ls_state *St, ls_scratch *S, ls_draws *D,
                       int c, int64_t iter, const double *normal, const double *pop,
                       int64_t zsize, double *tr_chisq, int32_t *tr_flag)
{
    const int nf = P->nfree;
    double *x = St->X + (size_t)c * nf;
    double *nextp = S->nextp, *jump = S->jump;
    double mrfactor = 1.0;
    int sjump = 0;
    const double *z = NULL;
    int32_t flag = 0;

    if (P->sampler == LS_SNOOKER) {                       /* chain.py:195-217 */
        const int64_t iR1 = draw_index(D, DR_IR1, 0, zsize);
        int64_t iR2 = draw_index(D, DR_IR2, 1, zsize);
        if (iR2 == iR1)
            iR2 = 0;
        sjump = draw_uniform(D, DR_SJUMP, 0.0, 1.0) < 0.1;
        const double *Z1 = St->Z + (size_t)iR1 * nf, *Z2 = St->Z + (size_t)iR2 * nf;
        if (sjump) {
            flag |= 4;
            const int64_t iz = draw_index(D, DR_IZ, 0, zsize);
            const double gs = draw_uniform(D, DR_GS, 1.2, 2.2);
            z = St->Z + (size_t)iz * nf;
            int same = 1;
            for (int k = 0; k < nf; k++)
                if (z[k] != x[k])
                    same = 0;
            if (same) {
                for (int k = 0; k < nf; k++)
                    jump[k] = gs * (Z2[k] - Z1[k]);
            } else {
                double *dz = S->dz;
                for (int k = 0; k < nf; k++)
                    dz[k] = x[k] - z[k];
                const double zp1 = dotn(D, Z1, dz, nf);
                const double zp2 = dotn(D, Z2, dz, nf);
                const double dd = dotn(D, dz, dz, nf);
                for (int k = 0; k < nf; k++)
                    jump[k] = ((gs * (zp1 - zp2)) * dz[k]) / dd;
            }
        } else {
            for (int k = 0; k < nf; k++)
                jump[k] = P->gamma * (Z1[k] - Z2[k]) + P->fepsilon * normal[k];
        }
    } else if (P->sampler == LS_MRW) {                    /* chain.py:219-220 */
        for (int k = 0; k < nf; k++)
            jump[k] = normal[k];
    } else {                                              /* chain.py:221-232 */
        const int n = P->nchains;
        int r1 = (int)draw_index(D, DR_R1, 1, n);
        if (r1 == c)
            r1 = 0;
        int r2 = (r1 + (int)draw_index(D, DR_R2, 2, n)) % n;
        if (r2 == c)
            r2 = (r1 + 1) % n;
        const double *x1 = pop + (size_t)r1 * nf, *x2 = pop + (size_t)r2 * nf;
        for (int k = 0; k < nf; k++)
            jump[k] = P->gamma * (x1[k] - x2[k]) + P->fepsilon * normal[k];
    }

    /* proposal and boundaries, chain.py:235-243.  nextp keeps whatever the previous
     * step of this worker left in its fixed and shared entries (chain.py:175: one
     * array for the whole run); they are rewritten before use, so starting from
     * params every time is the same thing. */
    memcpy(nextp, P->params, sizeof(double) * (size_t)P->nparams);
    int anyout = 0;
    for (int k = 0; k < nf; k++) {
        const int j = P->ifree[k];
        nextp[j] = x[k] + jump[k];
        if (nextp[j] < P->pmin[j] || nextp[j] > P->pmax[j])
            anyout = 1;
    }
    double nextchisq = NAN;
    if (anyout) {
        for (int k = 0; k < nf; k++) {
            const int j = P->ifree[k];
            if (nextp[j] < P->pmin[j] || nextp[j] > P->pmax[j]) {
                /* (shared by the chains' threads -- and by worker PROCESSES, ls_run_worker) */
                __atomic_fetch_add(&St->outbounds[k], 1, __ATOMIC_RELAXED);
            }
        }
    } else {
        flag |= 1;
        for (int s = 0; s < P->nshare; s++)               /* chain.py:246-247 */
            nextp[P->ishare[s]] = nextp[P->share_src[s]];
        nextchisq = eval_chisq(P, nextp, S);              /* chain.py:249 */
        if (sjump) {                                      /* chain.py:251-255 */
            double *np_free = S->dz;                      /* (dz is dead by now) */
            for (int k = 0; k < nf; k++)
                np_free[k] = nextp[P->ifree[k]];
            double cnorm = 0.0, nnorm = 0.0;
            for (int k = 0; k < nf; k++) {
                const double dc = x[k] - z[k];
                const double dn = np_free[k] - z[k];
                cnorm = D->fma_dot ? mc3_fma(dc, dc, cnorm) : cnorm + dc * dc;
                nnorm = D->fma_dot ? mc3_fma(dn, dn, nnorm) : nnorm + dn * dn;
            }
            mrfactor = mc3_det_pow_half(nnorm / cnorm, nf - 1);
        }
        /* Metropolis draw -- only for proposals inside the boundaries (chain.py:259) */
        const double u = draw_uniform(D, DR_METRO, 0.0, 1.0);
        const double ratio = mc3_det_exp(0.5 * (St->chisq[c] - nextchisq)) * mrfactor;
        if (ratio > u) {                                  /* chain.py:257-274 */
            flag |= 2;
            for (int k = 0; k < nf; k++)
                x[k] = nextp[P->ifree[k]];
            St->chisq[c] = nextchisq;
            St->numaccept[c] += 1;
            if (nextchisq < St->best_chisq[c]) {
                St->best_chisq[c] = nextchisq;
                St->best_iter[c] = iter;
                memcpy(St->bestx + (size_t)c * nf, x, sizeof(double) * (size_t)nf);
            }
        }
    }
    if (tr_chisq)
        *tr_chisq = nextchisq;
    if (tr_flag)
        *tr_flag = flag;
}

static void scratch_alloc(const ls_problem *P, ls_scratch *S)
{
    S->part = malloc(sizeof(double) * part_doubles(P));
    S->buf0 = malloc(sizeof(double) * (size_t)P->ndata);
    S->buf1 = malloc(sizeof(double) * (size_t)P->ndata);
    S->nextp = malloc(sizeof(double) * (size_t)P->nparams);
    S->jump = malloc(sizeof(double) * (size_t)P->nfree);
    S->normal = malloc(sizeof(double) * (size_t)P->nfree);
    S->dz = malloc(sizeof(double) * (size_t)P->nfree);
}

static void scratch_free(ls_scratch *S)
{
    free(S->part); free(S->buf0); free(S->buf1);
    free(S->nextp); free(S->jump); free(S->normal); free(S->dz);
}

/* One chain's part of refresh R (`thinning` iterations and the Z append): shared by the two
 * lockstep drivers below. */
static void refresh_chain(const ls_problem *P, ls_state *St, ls_scratch *S, ls_draws *D, int c,
                          int64_t R, int64_t R0, const double *snap, int64_t zsize,
                          int64_t row0, int need_normal)
{
    const int nf = P->nfree;
    for (int j = 0; j < P->thinning; j++) {
        const int64_t iter = R * P->thinning + j;
        const int64_t trow = (R - R0) * P->thinning + j;
        D->iter = (uint64_t)iter;
        D->chain = (uint32_t)c;
        if (need_normal)
            draw_normal(D, P, S->normal);
        chain_step(P, St, S, D, c, iter, S->normal, snap, zsize,
                   St->trace_chisq ? St->trace_chisq + trow * P->nchains + c : NULL,
                   St->trace_flag ? St->trace_flag + trow * P->nchains + c : NULL);
    }
    /* Z append, chain.py:276-289 (strided slot M0 + ID + k*nchains) */
    const int64_t row = row0 + c;
    memcpy(St->Z + (size_t)row * nf, St->X + (size_t)c * nf, sizeof(double) * (size_t)nf);
    St->log_post[row] = -0.5 * St->chisq[c];
    St->zchain[row] = c;
}

/* LOCKSTEP driver.  Advance chains [chain0, chain0+nlocal) by nrefresh Z refreshes of
 * `thinning` iterations each: Philox draws, every chain sees the others through the
 * snapshot of the previous refresh.  Rows of the other chains must already be in Z
 * when a later refresh needs them (the multi-rank exchange is the caller's business). */
int ls_run(const ls_problem *P, ls_state *St, int64_t nrefresh, int nthreads)
{
    const int nf = P->nfree;
    const int64_t R0 = St->refresh_done;
    if (P->M0 + (R0 + nrefresh) * P->nchains > P->zlen)
        return -1;
    /* one thread team for the whole call; chains of a refresh are independent,
     * the implicit barrier of the omp-for separates refreshes */
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        ls_scratch S;
        scratch_alloc(P, &S);
        ls_draws D;
        memset(&D, 0, sizeof D);
        D.seed = P->seed;
        D.sampler = P->sampler;
        const int need_normal = (P->sampler == LS_MRW) || (P->fepsilon != 0.0);
        for (int k = 0; k < nf; k++)
            S.normal[k] = 0.0;
        for (int64_t R = R0; R < R0 + nrefresh; R++) {
            /* what this refresh sees of the other chains: the rows of refresh R-1-zlag
             * (zlag = 0: of the refresh before -- per-iteration Jacobi; the declared
             * cadence of a run sharded over GPUs looks zlag refreshes further back, so
             * that a row has the time of zlag iterations to cross the fabric) */
            const int64_t Rs = R - 1 - (int64_t)P->zlag;
            const double *snap = (Rs < 0) ? St->snap0
                               : St->Z + (size_t)(P->M0 + Rs * P->nchains) * nf;
            const int64_t zsize = P->M0 + (Rs + 1 > 0 ? Rs + 1 : 0) * P->nchains;
            const int64_t row0 = P->M0 + R * P->nchains;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
            for (int c = P->chain0; c < P->chain0 + P->nlocal; c++)
                refresh_chain(P, St, &S, &D, c, R, R0, snap, zsize, row0, need_normal);
        }
        scratch_free(&S);
    }
    St->refresh_done = R0 + nrefresh;
    return 0;
}

/* The same lockstep run as the reference's fan-out lays it out (mc3/chain.py:163,190-191;
 * hub: mcmc_driver.py:204-221,300-307): `ncpu` WORKER PROCESSES, worker `wid` advancing the
 * chains wid, wid + ncpu, wid + 2 ncpu, ... one after another, all workers meeting once per
 * refresh (the reference: once per iteration, through a pipe each to the hub; here a
 * sense-reversing barrier in memory the workers share -- the cheapest meeting there is).
 * Every array of `St` and the barrier must live in memory shared by the workers (MAP_SHARED,
 * mapped before they were forked: oracle/lockstep.py run_processes); the out-of-bounds
 * counters are added to atomically (chain_step), everything else has one writer.  Results
 * equal ls_run's bit for bit (tests/test_lockstep_oracle.py).  bench.py's cpu_baseline
 * "port-processes" (SURVEY.md 8d(ii)). */
typedef struct {
    int32_t count, sense, nworkers, pad;
} ls_barrier;

/* 0 = everybody met; -1 = the meeting was called off (ls_barrier_abandon: a worker died or
 * was never started, and the others must not spin for ever -- the `pad` word says so) */
static int barrier_meet(ls_barrier *b, int *local_sense)
{
    *local_sense = !*local_sense;
    if (__atomic_add_fetch(&b->count, 1, __ATOMIC_ACQ_REL) == b->nworkers) {
        __atomic_store_n(&b->count, 0, __ATOMIC_RELAXED);
        __atomic_store_n(&b->sense, *local_sense, __ATOMIC_RELEASE);
        return 0;
    }
    int spins = 0;
    while (__atomic_load_n(&b->sense, __ATOMIC_ACQUIRE) != *local_sense) {
        if (__atomic_load_n(&b->pad, __ATOMIC_RELAXED) != 0)
            return -1;
        if (++spins > 4096) {
            sched_yield();          /* (more workers than cores must not starve each other) */
            spins = 0;
        }
    }
    return 0;
}

/* (for the parent of the workers, which only keeps the time: start and end meeting) */
int ls_barrier_attend(ls_barrier *b, int *local_sense) { return barrier_meet(b, local_sense); }
/* ... in two halves, so that the parent can look after its children while it waits: count
 * oneself in (1 = that completed the meeting), then ask whether it is over (1), still on (0)
 * or called off (-1) */
int ls_barrier_arrive(ls_barrier *b, int *local_sense)
{
    *local_sense = !*local_sense;
    if (__atomic_add_fetch(&b->count, 1, __ATOMIC_ACQ_REL) == b->nworkers) {
        __atomic_store_n(&b->count, 0, __ATOMIC_RELAXED);
        __atomic_store_n(&b->sense, *local_sense, __ATOMIC_RELEASE);
        return 1;
    }
    return 0;
}
int ls_barrier_passed(ls_barrier *b, int local_sense)
{
    if (__atomic_load_n(&b->sense, __ATOMIC_ACQUIRE) == local_sense)
        return 1;
    return __atomic_load_n(&b->pad, __ATOMIC_RELAXED) != 0 ? -1 : 0;
}
void ls_barrier_abandon(ls_barrier *b) { __atomic_store_n(&b->pad, 1, __ATOMIC_RELEASE); }

/* bar[0]: the workers' meeting per refresh (nworkers = ncpu); bar[1]: start and end of the
 * run, attended by the parent too (nworkers = ncpu + 1) */
int ls_run_worker(const ls_problem *P, ls_state *St, int64_t nrefresh, int wid, int ncpu,
                  ls_barrier *bar)
{
    const int nf = P->nfree;
    const int64_t R0 = St->refresh_done;
    if (P->M0 + (R0 + nrefresh) * P->nchains > P->zlen || ncpu < 1 || wid < 0 || wid >= ncpu)
        return -1;
    ls_scratch S;
    scratch_alloc(P, &S);
    ls_draws D;
    memset(&D, 0, sizeof D);
    D.seed = P->seed;
    D.sampler = P->sampler;
    const int need_normal = (P->sampler == LS_MRW) || (P->fepsilon != 0.0);
    for (int k = 0; k < nf; k++)
        S.normal[k] = 0.0;
    int sense = 0, sense_run = 0;
    int called_off = barrier_meet(&bar[1], &sense_run);  /* everybody stands ready */
    for (int64_t R = R0; R < R0 + nrefresh && !called_off; R++) {
        const int64_t Rs = R - 1 - (int64_t)P->zlag;
        const double *snap = (Rs < 0) ? St->snap0
                           : St->Z + (size_t)(P->M0 + Rs * P->nchains) * nf;
        const int64_t zsize = P->M0 + (Rs + 1 > 0 ? Rs + 1 : 0) * P->nchains;
        const int64_t row0 = P->M0 + R * P->nchains;
        for (int c = P->chain0 + wid; c < P->chain0 + P->nlocal; c += ncpu)   /* chain.py:190-191 */
            refresh_chain(P, St, &S, &D, c, R, R0, snap, zsize, row0, need_normal);
        called_off = barrier_meet(&bar[0], &sense);      /* mcmc_driver.py:300-307 */
    }
    if (!called_off)
        called_off = barrier_meet(&bar[1], &sense_run);  /* everybody is done */
    scratch_free(&S);
    if (called_off)
        return -2;                  /* (another worker is gone: the parent called the run off) */
    St->refresh_done = R0 + nrefresh;
    return 0;
}

/* SEQUENTIAL driver: the reference's worker loop with ncpu = 1 (chain.py:183-299),
 * replaying a tape of the reference's np.random draws (kind[] = TK_*, value[]).
 * Chains are visited one after another inside an iteration and see each other's
 * already-updated states (freepars for DEMC, chain.py:230-232; Z and zsize for
 * snooker, :197-217); rows are appended every `thinning` iterations at the strided
 * slot (mrw/demc, :288) or at zsize (snooker, :281-282 -- the same slot with one
 * worker); the run ends when Z is full (:279-280, so snooker makes one more partial
 * pass whose steps still count) or when chain 0 is complete (mrw/demc, :297-299).
 * The global best fit follows chain.py:268-274: *gbest_chisq/gbestx on entry = the
 * initial population's best (mcmc_driver.py:273-278).
 * Returns 0, or 1 (tape too short) / 2 (a draw of another kind was recorded: the
 * step rules diverged from the reference) / 3 (tape not used up). */
int ls_run_tape(const ls_problem *P, ls_state *St, const double *tape, const int8_t *kind,
                int64_t ntape, int fma_dot, int64_t *numaccept_total, double *gbest_chisq,
                double *gbestx, int64_t *steps_done)
{
    const int nf = P->nfree, n = P->nchains;
    ls_scratch S;
    scratch_alloc(P, &S);
    ls_draws D;
    memset(&D, 0, sizeof D);
    D.tape_mode = 1;
    D.tape = tape;
    D.kind = kind;
    D.len = ntape;
    D.sampler = P->sampler;
    D.fma_dot = fma_dot;
    int64_t zsize = P->M0, nsteps = 0, iter = 0;
    int64_t *index = malloc(sizeof(int64_t) * (size_t)n);
    int64_t *chainsize = malloc(sizeof(int64_t) * (size_t)n);
    for (int c = 0; c < n; c++) {
        index[c] = P->M0 + c;                             /* chain.py:163-164 */
        chainsize[c] = P->M0 / n;
    }
    const int64_t chainlen = P->zlen / n;                 /* chain.py:155 */
    int njump = 0, done = 0;
    while (!done && !D.error) {
        njump += 1;
        draw_normal(&D, P, S.normal);                     /* chain.py:185: one per iteration */
        for (int c = 0; c < n && !D.error; c++) {
            const int acc0 = (int)St->numaccept[c];
            chain_step(P, St, &S, &D, c, iter, S.normal, St->X, zsize,
                       St->trace_chisq ? St->trace_chisq + nsteps : NULL,
                       St->trace_flag ? St->trace_flag + nsteps : NULL);
            nsteps += 1;
            if ((int)St->numaccept[c] != acc0 && St->chisq[c] < *gbest_chisq) {
                *gbest_chisq = St->chisq[c];              /* chain.py:268-274 */
                memcpy(gbestx, St->X + (size_t)c * nf, sizeof(double) * (size_t)nf);
            }
            if (njump == P->thinning) {                   /* chain.py:276-289 */
                if (zsize == P->zlen) {
                    done = 1;
                    break;
                }
                if (P->sampler == LS_SNOOKER)
                    index[c] = zsize;
                zsize += 1;
                St->zchain[index[c]] = c;
                memcpy(St->Z + (size_t)index[c] * nf, St->X + (size_t)c * nf,
                       sizeof(double) * (size_t)nf);
                St->log_post[index[c]] = -0.5 * St->chisq[c];
                index[c] += n;
                chainsize[c] += 1;
            }
        }
        if (njump == P->thinning)
            njump = 0;
        iter += 1;
        if (P->sampler != LS_SNOOKER && chainsize[0] == chainlen)
            done = 1;                                     /* chain.py:297-299 */
    }
    int64_t total = 0;
    for (int c = 0; c < n; c++)
        total += St->numaccept[c];
    *numaccept_total = total;
    *steps_done = nsteps;
    St->refresh_done = (zsize - P->M0) / n;
    const int rc = D.error ? D.error : (D.pos != D.len ? 3 : 0);
    free(index); free(chainsize);
    scratch_free(&S);
    return rc;
}

/* ---- initial population: mc3/mcmc_driver.py:229-270 in the Philox form --------------
 * Trial t: nfree draws around the initial guess (kickoff 0: params + pstep * N(0,1),
 * Box-Muller pairs; 1: uniform between the boundaries) from Philox(seed; counter = (t,
 * chain = 0xffffffff, slot)); accepted in trial order when every parameter is inside its
 * boundaries -- tested on the full vector BEFORE the shared parameters are refreshed
 * (:249-255) -- and chisq is finite; ends with M0 rows or after 100 M0 failures.
 * Z0 [M0][nfree], lp0 [M0] (zero beyond *filled). */
/* Where the trial vectors of the kick-off come from: Philox (what the GPU draws) or a tape of
 * the reference's own np.random trial vectors (tests/golden/make_golden.py kickoff_case).
 * ONE statement of the acceptance rule, init_population_core, serves both -- the kick-off's
 * counterpart of chain_step() and its draw source. */
typedef struct {
    const double *tape;     /* [ntape][nfree] recorded trial vectors, or NULL: Philox */
    int64_t ntape;
    int kickoff;            /* Philox: 0 normal, 1 uniform */
    uint64_t seed;
} kick_source;

/* trial t into trial[nfree]; 0 when the tape has run out */
static int kick_trial(const ls_problem *P, const kick_source *src, uint64_t t, double *trial)
{
    const int nf = P->nfree;
    if (src->tape) {
        if ((int64_t)t >= src->ntape)
            return 0;
        memcpy(trial, src->tape + (size_t)t * nf, sizeof(double) * (size_t)nf);
        return 1;
    }
    for (int k2 = 0; 2 * k2 < nf; k2++) {
        uint64_t q0, q1;
        draw2(src->seed, t, 0xffffffffu, (uint32_t)k2, &q0, &q1);
        double a, b;
        if (src->kickoff) {
            a = u01(q0);
            b = u01(q1);
        } else {
            const double rad = __builtin_sqrt(-2.0 * mc3_det_log(u01_open(q0)));
            double sn, cs;
            mc3_det_sincos2pi(u01(q1), &sn, &cs);
            a = rad * cs;
            b = rad * sn;
        }
        for (int h = 0; h < 2 && 2 * k2 + h < nf; h++) {
            const int k = 2 * k2 + h, j = P->ifree[k];
            const double r = h ? b : a;
            trial[k] = src->kickoff ? P->pmin[j] + (P->pmax[j] - P->pmin[j]) * r
                                    : P->params[j] + P->pstep[j] * r;
        }
    }
    return 1;
}

/* mcmc_driver.py:240-262: trials in order; a trial fails when any parameter of the full vector
 * lies outside its boundaries -- tested BEFORE the shared parameters are refreshed, i.e. on
 * the copies the last in-bounds trial left (:249-255) -- or when its chisq is not finite; ends
 * with M0 rows or after 100 M0 failures.  *used_out: trials consumed. */
static void init_population_core(const ls_problem *P, const kick_source *src, double *Z0,
                                 double *lp0, int64_t *filled_out, int64_t *failed_out,
                                 int64_t *used_out)
{
    const int nf = P->nfree;
    ls_scratch S;
    scratch_alloc(P, &S);
    double *values = malloc(sizeof(double) * (size_t)P->nparams);
    double *trial = malloc(sizeof(double) * (size_t)nf);
    memcpy(values, P->params, sizeof(double) * (size_t)P->nparams);
    memset(Z0, 0, sizeof(double) * (size_t)P->M0 * (size_t)nf);
    memset(lp0, 0, sizeof(double) * (size_t)P->M0);
    int64_t filled = 0, failed = 0;
    uint64_t t = 0;
    const int64_t nmax = 100 * P->M0;
    for (; filled < P->M0 && failed < nmax; t++) {
        if (!kick_trial(P, src, t, trial))
            break;
        for (int k = 0; k < nf; k++)
            values[P->ifree[k]] = trial[k];
        int out = 0;
        for (int j = 0; j < P->nparams; j++)
            if (values[j] > P->pmax[j] || values[j] < P->pmin[j])
                out = 1;
        if (out) {
            failed++;
            continue;
        }
        for (int s = 0; s < P->nshare; s++)
            values[P->ishare[s]] = values[P->share_src[s]];
        const double lp = -0.5 * eval_chisq_untiled(P, values, &S);
        if (!(lp - lp == 0.0)) {                          /* inf or NaN */
            failed++;
            continue;
        }
        memcpy(Z0 + (size_t)filled * nf, trial, sizeof(double) * (size_t)nf);
        lp0[filled] = lp;
        filled++;
    }
    *filled_out = filled;
    *failed_out = failed;
    *used_out = (int64_t)t;
    free(values); free(trial);
    scratch_free(&S);
}

int ls_init_population(const ls_problem *P, int kickoff, uint64_t seed, double *Z0,
                       double *lp0, int64_t *filled_out, int64_t *failed_out)
{
    const kick_source src = {NULL, 0, kickoff, seed};
    int64_t used;
    init_population_core(P, &src, Z0, lp0, filled_out, failed_out, &used);
    return 0;
}

/* the same rule fed the reference's own trial vectors, tape[ntape][nfree] in the order its hub
 * drew them */
int ls_init_population_tape(const ls_problem *P, const double *tape, int64_t ntape, double *Z0,
                            double *lp0, int64_t *filled_out, int64_t *failed_out,
                            int64_t *used_out)
{
    const kick_source src = {tape, ntape, 0, 0};
    init_population_core(P, &src, Z0, lp0, filled_out, failed_out, used_out);
    return 0;
}

/* exported so tests can pin the shared deterministic functions to libm */
double ls_det_exp(double x) { return mc3_det_exp(x); }
double ls_det_exp2(double x) { return mc3_det_exp2(x); }
double ls_det_log(double x) { return mc3_det_log(x); }
void ls_det_sincos2pi(double u, double *s, double *c) { mc3_det_sincos2pi(u, s, c); }
double ls_det_pow_half(double b, int e2) { return mc3_det_pow_half(b, e2); }
