/*
 * mc3hip.h -- C-ABI of libmc3hip.so: mc3's per-iteration chain update on MI355X.
 *
 * Drop-in boundary.  The reference has no C-ABI for this
 * path: its native code is two CPython modules (_chisq, _dwt) called from a
 * Python loop that runs in forked worker processes.  The entry points below are
 * what a binding for the same path binds instead -- plain pointers and sizes, no
 * Python, NumPy or torch types.  Each one cites the reference code it replaces
 * (paths relative to the mc3 v3.2.1 tree).  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns MC3HIP_OK (0) or a negative error code and never
 *     aborts the process; mc3hip_last_error() gives the message (thread-local).
 *   - host arrays are contiguous float64 / int32 / int64, C order.  (The
 *     reference reads arbitrary strides through INDd, src_c/include/ind.h:7;
 *     the Python edge makes arrays contiguous before calling.)
 *   - the caller owns every host buffer; the library owns device memory for the
 *     lifetime of the context and never keeps a host pointer after a call.
 *   - one context drives one GPU on one HIP stream; contexts are independent.
 */
#ifndef MC3HIP_H
#define MC3HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC3HIP_OK            0
#define MC3HIP_ERR_ARG      -1   /* invalid argument / unsupported configuration */
#define MC3HIP_ERR_HIP      -2   /* HIP runtime error (message has the HIP string) */
#define MC3HIP_ERR_STATE    -3   /* call sequence error (e.g. run before set_population) */
#define MC3HIP_ERR_FULL     -4   /* Z buffer exhausted (chain.py:279 "stop when we fill Z") */
#define MC3HIP_ERR_COMM     -5   /* RCCL error */
#define MC3HIP_ERR_ABORTED  -6   /* peer-to-peer mode: a persistent launch gave up waiting for
                                    another rank's rows; see mc3hip_p2p_rollback             */

/* sampler: mc3/chain.py:195,219,221 */
#define MC3HIP_MRW      0
#define MC3HIP_DEMC     1
#define MC3HIP_SNOOKER  2

/* declared model family (the reference calls an arbitrary Python callable,
 * chain.py:317-319; the device path evaluates these two families in-kernel) */
#define MC3HIP_MODEL_POLY    0   /* model_i = sum_k p[k] * x_i^k, k < nmodel      */
#define MC3HIP_MODEL_LINEAR  1   /* model_i = sum_k A[k][i] * p[k], k < nmodel    */

#define MC3HIP_FLAG_NO_GRAPH    1u /* launch eagerly instead of through a hipGraph  */
#define MC3HIP_FLAG_NO_PERSIST  2u /* never use the persistent kernel               */
#define MC3HIP_FLAG_P2P         4u /* allocate Z / log_post fine-grained and IPC-shareable
                                      so that mc3hip_p2p_attach can be used               */
#define MC3HIP_FLAG_NO_COURIER  8u /* peer-to-peer mode: the chain wave stores its rows into
                                      the other ranks' replicas itself instead of handing
                                      them to a second wave (for more than two ranks on ONE
                                      device, a test set-up: the two-wave workgroups of all
                                      of them would not be co-resident)                    */

#define MC3HIP_FLAG_NO_JIT     16u /* (rounds 4-5: "the caller will not specialise"; since round 6
                                      that is the DEFAULT and this bit is ignored)          */
#define MC3HIP_FLAG_JIT        32u /* the caller WILL specialise at run time: it asks
                                      mc3hip_jit_request for the loop to compile and hands the
                                      code object to mc3hip_jit_install before the first
                                      mc3hip_run.  Only with this bit is a shape planned FOR such
                                      a loop where that changes the data layout (2 049 ... 16 384
                                      points with fixed / shared coefficients, priors, the random
                                      walk, design matrices: tiles in the registers of four or
                                      eight waves instead of the tiled plan) -- a caller with
                                      flags = 0, or one whose compiler failed, gets the plan the
                                      library serves with its own instances                  */

#define MC3HIP_MAX_FREE     64
#define MC3HIP_MAX_PARAMS  128

typedef struct mc3hip_ctx mc3hip_ctx;

/*
 * Everything Chain.__init__ receives (mc3/chain.py:22-155) plus the sizes
 * mcmc() derives (mc3/mcmc_driver.py:116-134).
 */
typedef struct mc3hip_problem {
    int32_t struct_bytes;   /* sizeof(mc3hip_problem), checked                          */
    int32_t device;         /* HIP device ordinal                                       */
    int32_t nchains;        /* total number of chains (all ranks)                       */
    int32_t chain0;         /* first chain owned by this context                        */
    int32_t nlocal;         /* chains owned by this context (contiguous block)          */
    int32_t nparams;        /* len(params), incl. fixed and shared                      */
    int32_t ndata;
    int32_t sampler;        /* MC3HIP_MRW / _DEMC / _SNOOKER                            */
    int32_t wlike;          /* 0: Gaussian chisq (_chisq.c:111); 1: wavelet (_dwt.c:56),
                               power-of-two ndata <= 2^22                               */
    int32_t model_kind;     /* MC3HIP_MODEL_*                                           */
    int32_t nmodel;         /* model coefficients = params[0:nmodel]                    */
    int32_t thinning;       /* iterations per Z refresh (chain.py:276)                  */
    uint32_t flags;
    int32_t trace_iters;    /* >0: keep per-step chisq/flags of the last run (tests)    */
    int64_t M0;             /* rows of the initial population (mcmc_driver.py:127)      */
    int64_t zlen;           /* total Z rows = M0 + nzchain*nchains (mcmc_driver.py:134) */
    uint64_t seed;          /* Philox key                                               */
    double gamma;           /* fgamma*2.38/sqrt(2 nfree) (chain.py:175)                 */
    double fepsilon;        /* chain.py:217,232                                         */
    const double *params;   /* [nparams] initial / fixed values                         */
    const double *pstep;    /* [nparams] >0 free, 0 fixed, <0 shared (chain.py:141-142) */
    const double *pmin;     /* [nparams]                                                */
    const double *pmax;     /* [nparams]                                                */
    const double *prior;    /* [nparams]                                                */
    const double *priorlow; /* [nparams] Gaussian prior iff priorlow>0 and priorup>0    */
    const double *priorup;  /* [nparams]                               (stats.py:211)   */
    const double *indep;    /* POLY: x[ndata]; LINEAR: A[nmodel][ndata]                 */
    const double *data;     /* [ndata]                                                  */
    const double *uncert;   /* [ndata] (unused when wlike, chain.py:327-329)            */
    int32_t zlag;           /* declared cadence of a run sharded over GPUs: the other chains are
                               seen `zlag` refreshes later than in the strict lockstep -- DEMC
                               partners of refresh R (chain.py:221-232) are the rows of refresh
                               R-1-zlag (the first nchains rows of the initial population until
                               there is one), snooker's history (chain.py:197-205) ends with
                               refresh R-1-zlag -- so that a row has the time of zlag refreshes
                               to cross the fabric and an iteration costs max(arithmetic, hop /
                               zlag), not their sum.  0 = strict (the one-GPU default).  Results
                               depend on zlag and on nothing else: not on how the chains are
                               sharded, not on the launch structure              */
    int32_t wl_nreal;       /* wavelet likelihood of a series whose length is no power of two:
                               the caller pads data / uncert / indep to ndata = 2^M points and
                               says here how many of them are data (0 or ndata: all).  The
                               residuals of the padding points are exactly zero whatever the
                               model says there -- the zero padding of src_c/_dwt.c:82-91 made
                               well defined (the reference then transforms the UNPADDED length,
                               _dwt.c:94, which only 2^M survives); ignored without wlike      */
} mc3hip_problem;

/* how the library sums over data points (the checker follows the same order) */
typedef struct mc3hip_plan {
    int32_t vlanes;         /* strided partial sums per tile                  */
    int32_t tile;           /* points per tile, 0 = one tile                  */
    int32_t waves_per_chain;
    int32_t graph_nodes;    /* refreshes per hipGraph replay, 0 = eager       */
    int32_t persistent;     /* >0: refreshes per launch of the persistent kernel
                               (in-launch hand-off of Z rows), 0 = one launch per refresh */
    int32_t fast;           /* 1: the lean persistent instance (one wave per chain, fully
                               free polynomial, Gaussian chisq, no priors/shared/epsilon);
                               2: one of the loops laid out by hand for that shape -- DEMC
                               (lean_demc_kernel; thinning > 1 or zlag > 0: lean_demc_lag_kernel), snooker
                               (lean_snooker_kernel), random walk (lean_mrw_kernel), each also
                               with two to four chains per wave (lean_*_mc_kernel);
                               3: wavelet likelihood, nine-wave kernel with the hand-structured
                               chain wave (thinning 1, no priors / shared parameters / trace);
                               4: resident tiled plan with the hand-structured chain owner
                               (DEMC, thinning 1, no shared parameters / trace)             */
    int32_t comm_graph;     /* >0: the RCCL exchange runs inside a hipGraph of that many
                               refreshes (step kernel + all-gather + scatter per refresh)   */
    int32_t recoveries;     /* persistent launches that gave up and were redone so far        */
    int32_t tile_tree;      /* how a tile's 1024 strided partial sums are added up: 0 = one
                               halving tree over all of them (strides 512..1); 2 = the
                               matrix-core tile kernels: the weights folded into the operand
                               rows (design columns and data times w, rounded once), a lane's
                               sixteen points in step order, the 16 lanes of a row (8..1), the
                               tile's eight rows (4..1) -- oracle_lockstep.c
                               gauss_chisq_weighted_rows states it                              */
    int32_t resident;       /* 1 = tiled plan in ONE persistent launch: every workgroup keeps
                               its tile of the data (and of the design matrix) on chip,
                               proposals and per-tile sums are handed over inside the
                               launch (design matrices of 5..11 columns, polynomials of
                               2..4 coefficients)                                          */
    int32_t chains_per_workgroup; /* persistent launch: chains one workgroup advances per
                               refresh, one after another (> 1 when the ensemble has more
                               chains than workgroups can be co-resident; the reference's
                               workers loop over their chains the same way, chain.py:163,
                               190-191); results do not depend on it                          */
    int32_t jit;            /* 1: the hand-laid-out loop that runs was specialised at run time
                               for this model (mc3hip_jit_request / mc3hip_jit_install)         */
    int32_t capacity;       /* persistent launch: workgroups of the instance that runs which the
                               device holds at once (occupancy query x CUs); the launch's grid
                               -- nlocal, or nlocal / chains_per_workgroup -- must fit, and so
                               must the SUM over contexts that share a device (ranks of a test
                               set-up on one GPU: mc3_amd/parallel.py checks it)                */
    int32_t zlag;           /* the cadence this context runs (mc3hip_problem.zlag)              */
    int32_t remote_delay_ns;/* test hook MC3HIP_TEST_REMOTE_DELAY_NS (read at creation): every
                               row bound for ANOTHER rank's replica is held back that long
                               before it is stored -- several ranks on one GPU then behave like
                               ranks an xGMI hop apart; 0 in production                        */
    int32_t owner_waves;    /* resident tiled plan: > 0 = the chains' owners are that many waves
                               of their own behind a workgroup's eight tile waves, and a tile
                               wave makes up its blocks of sixteen chains at run time from the
                               proposals that are there (resident_async_kernel: nobody waits
                               for the ensemble's slowest chain); 0 = the phases in step      */
} mc3hip_plan;

/* Replaces Chain.__init__ + the shared-memory allocation of mcmc()
 * (mc3/chain.py:22-155, mc3/mcmc_driver.py:143-221): uploads the problem and
 * allocates Z / log_post / zchain / freepars / counters on the device. */
int mc3hip_create(const mc3hip_problem *problem, mc3hip_ctx **out);
void mc3hip_destroy(mc3hip_ctx *ctx);
const char *mc3hip_last_error(void);
int mc3hip_get_plan(const mc3hip_ctx *ctx, mc3hip_plan *plan);

/* Run-time specialisation ("the model ... JIT-specialised into the kernel").  The library is
 * built with the hand-laid-out one-wave loops for polynomials of 2..4 coefficients; for the
 * other members of the declared family that fit one register tile -- polynomials of 5..16
 * coefficients, design matrices of up to 16 columns, ndata <= 1024, one GPU -- the same loop
 * source is compiled for the very model (coefficient count a compile-time constant, the design
 * matrix's tile resident in registers) by the caller, who owns a compiler; the library itself
 * never needs one.  So are (round 5) polynomials of 2..4 coefficients on 1 025 ... 8 192 points
 * with what the library's own loops for those sizes do not carry -- a fixed or shared
 * coefficient, a Gaussian prior, fepsilon != 0, the random walk: the tiles in the registers of
 * four waves ("lean_mrw_body<3, true, false, false, false, false, 4, 8> /" "*threads=256*" "/").
 *   mc3hip_jit_request: the expression to instantiate, e.g.
 *     "lean_demc_body<203, false, false, false, false, false, false>" (C string into buf;
 *     empty: nothing to specialise -- the library has the kernel, or the shape has no loop).
 *     A request that ends in a comment "/" "*threads=N*" "/" is for a workgroup of N threads
 *     (helper waves share the data sum of a design matrix of five columns and more: N = 256);
 *     without it N = 64.  The caller compiles, for gfx950 and with the library's flags
 *     (-O3 -ffp-contract=off),
 *       #include "kernels_lean.h"
 *       extern "C" __global__ __launch_bounds__(N)
 *       void mc3_jit_kernel(const mc3hip::DevProblem P, const int nref) { mc3hip::<expr>(P, nref); }
 *     into a code object (mc3_amd/jit.py: hipcc --genco, cached on disk by content hash);
 *   mc3hip_jit_install: hands the code object over (before the first mc3hip_run).  The context
 *     then runs that kernel wherever it would have run the general chain for want of an
 *     instance; results are those of the general chain, bit for bit. */
int mc3hip_jit_request(const mc3hip_ctx *ctx, char *buf, int32_t capacity);
int mc3hip_jit_install(mc3hip_ctx *ctx, const void *code_object, int64_t bytes,
                       const char *kernel_name);

/* Replaces the hub's initial-population loop (mc3/mcmc_driver.py:229-270): M0 trial
 * vectors around the initial guess (kickoff 0: params + pstep * N(0,1); 1: uniform between
 * the boundaries), the reference's acceptance rule applied in trial order (inside the
 * boundaries -- shared parameters tested with the copies the last accepted trial left --
 * and finite chisq; at most 100 M0 failures), chisq on the device.  Trials are drawn from
 * Philox(seed) on a stream of their own, so the population does not depend on how the
 * chains are sharded.  Z0 [M0][nfree] and log_post0 [M0] receive the rows (zeros beyond
 * *filled); the caller applies the reference's "filled >= M0 - 1" test. */
int mc3hip_init_population(mc3hip_ctx *ctx, int32_t kickoff, uint64_t seed, double *Z0,
                           double *log_post0, int64_t *filled, int64_t *failed);

/* Replaces the hub's Z[0:M0] / log_post[0:M0] fill and the workers' start-up
 * (mcmc_driver.py:260-261, chain.py:163-170): Z0 is [M0][nfree], the chains start
 * from its first nchains rows unless X0/chisq0 ([nchains][nfree], [nchains]) are
 * given (resume: chain.py:166-169). */
int mc3hip_set_population(mc3hip_ctx *ctx, const double *Z0,
                          const double *log_post0, const double *X0,
                          const double *chisq0);

/* Replaces Chain.run's loop (mc3/chain.py:183-299) and the hub's per-iteration
 * barrier (mcmc_driver.py:300-307): advance every local chain by nrefresh Z
 * refreshes (nrefresh*thinning iterations).  Asynchronous; mc3hip_sync waits. */
int mc3hip_run(mc3hip_ctx *ctx, int64_t nrefresh);
int mc3hip_sync(mc3hip_ctx *ctx);
/* Streaming runs: the reference sizes Z for the whole run and stops when it is full
 * (mc3/mcmc_driver.py:134, chain.py:279-280); a caller that has taken the rows it wants
 * (mc3hip_get_rows, the streaming checkpoint) recycles the buffer instead.  The last `keep`
 * refreshes (rows, log_post, zchain) become refreshes [0, keep), everything behind them is
 * unwritten again, the chains carry on from where they are and draw from Philox(seed) from
 * then on (iteration numbers start over, so the key must change).  keep >= 1 + zlag, at
 * most half of the refreshes done; mrw / demc only -- snooker draws from the WHOLE history
 * (chain.py:197-205), which recycling would cut.  Ranks of a peer-to-peer run all call it
 * at the same refresh, between two barriers -- and, like the next mc3hip_run, only AFTER they
 * have agreed that nobody's launch gave up (mc3hip_sync on every rank, then a vote:
 * Communicator.quiesce): the checkpoint a roll-back would need is released here.  A rank
 * whose own launch gave up gets MC3HIP_ERR_ABORTED and keeps its checkpoint. */
int mc3hip_recycle(mc3hip_ctx *ctx, int64_t keep, uint64_t seed);
/* GPU time in ms between the start and end of the last mc3hip_run (HIP events
 * recorded on the context's stream). */
int mc3hip_last_run_ms(mc3hip_ctx *ctx, float *ms);

/* What update_output reads from the Chain object (mc3/stats/stats.py:813-826):
 * rows [row0, row0+nrows) of Z / log_post / zchain.  Any pointer may be NULL. */
int mc3hip_get_rows(mc3hip_ctx *ctx, int64_t row0, int64_t nrows, double *Z,
                    double *log_post, int32_t *zchain);

/* The same for rows that can no longer change -- the initial population and the refreshes an
 * mc3hip_sync has confirmed -- on a stream of the call's own: neither waits for nor disturbs a
 * run in flight, so a long run's host copy (what mcmc() hands to update_output,
 * mc3/mcmc_driver.py:350-352, mc3/stats/stats.py:805-852) is made while the next segment
 * computes.  One-GPU runs; MC3HIP_ERR_STATE for rows beyond the confirmed refreshes. */
int mc3hip_get_rows_settled(mc3hip_ctx *ctx, int64_t row0, int64_t nrows, double *Z,
                            double *log_post, int32_t *zchain);
/* Overwrite rows of Z / log_post that ANOTHER rank owns (host-staged exchange for
 * callers that cannot use the RCCL path below; same visibility rule: rows of
 * refresh R must be in place before refresh R+1 runs). */
int mc3hip_put_rows(mc3hip_ctx *ctx, int64_t row0, int64_t nrows, const double *Z,
                    const double *log_post);
/* Current chain states and counters (freepars, chisq, numaccept, outbounds and
 * the best-fit tracking of chain.py:261-274), indexed by LOCAL chain.  best_iter
 * is the iteration at which a chain found its best state (-1: never accepted).
 * Any pointer may be NULL. */
int mc3hip_get_state(mc3hip_ctx *ctx, double *X, double *chisq,
                     int64_t *numaccept, int64_t *outbounds,
                     double *best_chisq, int64_t *best_iter, double *bestx,
                     int64_t *refresh_done);
/* Per-step trace of the last run (problem.trace_iters > 0): chisq of the
 * proposal and flags (bit0 in bounds, bit1 accepted, bit2 snooker jump),
 * [iters][nlocal]. */
int mc3hip_get_trace(mc3hip_ctx *ctx, int64_t iters, double *chisq,
                     int32_t *flags);

/* The reductions of the Gelman-Rubin test (mc3/stats/gelman.py:12-92, called from
 * the hub at mcmc_driver.py:327-346) on the device: mean and population variance of
 * every free parameter per chain over the refreshes done so far, skipping each
 * chain's first `burn` rows; [nchains][nfree] each.  The caller combines them into
 * the potential scale reduction factor; Z is not copied to the host. */
int mc3hip_chain_moments(mc3hip_ctx *ctx, int64_t burn, double *mean, double *var);
/* The same over the first nrefresh refreshes' rows as they stand in Z, whoever wrote them
 * (rows installed with mc3hip_put_rows, e.g. those of a run that is being resumed). */
int mc3hip_chain_moments_rows(mc3hip_ctx *ctx, int64_t nrefresh, int64_t burn,
                              double *mean, double *var);

/* Replaces the NumPy passes of update_output / calc_sample_statistics over the burned
 * posterior (mc3/stats/stats.py:805-852,876-964; np.mean, np.std, np.percentile) with
 * reductions over the rows where they are: per free parameter the mean, population
 * variance, minimum and maximum of the rows of refreshes [burn, nrefresh) (nrefresh <= 0:
 * all that have run), and exact order statistics (0-based ranks, nranks <= 8, radix
 * selection) from which the caller interpolates medians and percentiles exactly as NumPy
 * does.  order_stats is [nranks][nfree]. */
int mc3hip_posterior_summary(mc3hip_ctx *ctx, int64_t nrefresh, int64_t burn, int32_t nranks,
                             const int64_t *ranks, double *order_stats, double *mean,
                             double *var, double *vmin, double *vmax);
/* output['chisq'] of update_output (stats.py:821): -2 (log_post - log_prior) of rows
 * [row0, row0+nrows), log_prior as mc3/stats/stats.py:287-392. */
int mc3hip_get_chisq(mc3hip_ctx *ctx, int64_t row0, int64_t nrows, double *chisq);
/* The Gaussian kernel density cred_region evaluates through scipy (stats.py:395-467,
 * gaussian_kde.evaluate): pdf[j] = sum_i exp(-(x_j - s_i)^2 / (2 cov)) / (n sqrt(2 pi cov)). */
int mc3hip_kde_pdf(int32_t device, const double *sample, int32_t n, const double *x,
                   int32_t m, double cov, double *pdf);

/* Replaces Chain.eval_model(params, ret='chisq') -> ms.chisq / ms.dwt_chisq
 * (chain.py:302-340, stats.py:150-284, _chisq.c:111-140, _dwt.c:56-119) for n
 * parameter vectors at once ([n][nparams], shared parameters already copied). */
int mc3hip_eval_chisq(mc3hip_ctx *ctx, const double *params, int32_t n,
                      double *chisq_out);
/* Replaces Chain.eval_model(params, ret='model') for the declared family. */
int mc3hip_eval_model(mc3hip_ctx *ctx, const double *params, double *model_out);

/* The public functions of the reference's two C extension modules on arbitrary
 * model arrays (what mc3.stats.chisq / residuals / dwt_chisq call,
 * mc3/stats/stats.py:97-284): _chisq.residuals (src_c/_chisq.c:37-79),
 * _chisq.chisq (:111-140) and _dwt.chisq (src_c/_dwt.c:56-119).  off/plow/pup are the
 * np Gaussian-prior entries (param - prior, lower and upper widths) the Python
 * wrapper selects (stats.py:211-216); np may be 0.  residuals_out holds n + np
 * doubles.  dwt_chisq needs power-of-two n <= 2^24 (see mc3hip_create). */
int mc3hip_residuals(int32_t device, const double *model, const double *data,
                     const double *uncert, int32_t n, const double *off,
                     const double *plow, const double *pup, int32_t np,
                     double *residuals_out);
int mc3hip_chisq(int32_t device, const double *model, const double *data,
                 const double *uncert, int32_t n, const double *off, const double *plow,
                 const double *pup, int32_t np, double *chisq_out);
int mc3hip_dwt_chisq(int32_t device, const double *params, int32_t nparams,
                     const double *model, const double *data, int32_t n,
                     const double *off, const double *plow, const double *pup,
                     int32_t np, double *chisq_out);

/* Replaces ms.dwt_daub4 (_dwt.c:154-186, wavelet.h:16-128): n is padded to the
 * next power of two with zeros, out must hold that many doubles. */
int mc3hip_dwt_daub4(int32_t device, const double *in, int32_t n,
                     int32_t inverse, double *out);

/* Multi-GPU (no reference counterpart; replaces the multiprocessing fan-out,
 * mcmc_driver.py:204-221): one context per rank owns a contiguous block of
 * chains; after every refresh the new Z rows are exchanged with ONE in-place
 * ncclAllGather (RCCL) so that the next proposals see the whole population.
 * unique_id: 128 bytes from mc3hip_comm_unique_id on rank 0, sent to the other
 * ranks by the caller. */
int mc3hip_comm_unique_id(void *unique_id_128);
int mc3hip_comm_init(mc3hip_ctx *ctx, const void *unique_id_128, int32_t rank,
                     int32_t nranks);
/* Exchange log_post rows [row0, row0+nrows) that other ranks own (needed only
 * when the posterior is read back). */
int mc3hip_comm_gather_log_post(mc3hip_ctx *ctx);
/* What RCCL itself reports for the communicator of mc3hip_comm_init (ncclCommCount,
 * ncclCommUserRank): the evidence a multi-GPU measurement carries that the collective
 * really spanned that many ranks.  MC3HIP_ERR_STATE without a communicator. */
int mc3hip_comm_info(mc3hip_ctx *ctx, int32_t *nranks, int32_t *rank);
/* Free and total device memory in bytes (hipMemGetInfo): callers size Z, which grows
 * with the run length (128-byte rows: core.h ZROW), against it.  No reference counterpart
 * (the reference's Z lives in host shared memory, mcmc_driver.py:158-171). */
int mc3hip_device_memory(int32_t device, int64_t *free_bytes, int64_t *total_bytes);

/* Peer-to-peer variant of the multi-GPU exchange (contexts created with
 * MC3HIP_FLAG_P2P, one per rank/process, equal contiguous chain blocks): every rank
 * exports IPC handles of its Z / log_post replicas (128 bytes), the caller hands
 * all ranks' handles to every rank, and from then on the persistent kernel publishes
 * each new row straight into every replica (system-scope stores over xGMI) and polls
 * its local one -- no collective and no kernel boundary between iterations.  The
 * caller must barrier across ranks after attach (before the first run) and after
 * mc3hip_sync (before reading rows back: peers may still be storing into this
 * replica). */
int mc3hip_p2p_export(mc3hip_ctx *ctx, void *handles_128);
int mc3hip_p2p_attach(mc3hip_ctx *ctx, int32_t rank, int32_t nranks,
                      const void *all_handles /* [nranks][128] */);
/* Undo mc3hip_p2p_attach (e.g. when it succeeded on some ranks only): unmaps the peers'
 * replicas; the context is a single-rank / RCCL / host-staged context again. */
int mc3hip_p2p_detach(mc3hip_ctx *ctx);
/* Recovery of the peer-to-peer mode, the counterpart of what mc3hip_sync does by itself on
 * one GPU.  When mc3hip_sync (or any read-back call) returned MC3HIP_ERR_ABORTED on ANY
 * rank, every rank -- after all of them have synchronised and met at a barrier, so that
 * nobody stores into a replica any more -- calls this: the per-chain state goes back to
 * the checkpoint taken before the first unconfirmed run, the rows written since are
 * discarded, the peer-to-peer hand-off is dropped, and [*from_refresh, *to_refresh) is
 * what has to be redone through mc3hip_comm_init or the host-staged exchange
 * (mc3hip_run + mc3hip_put_rows).  Contract that makes the checkpoint sufficient: in
 * peer-to-peer mode a caller starts the next mc3hip_run only after all ranks agreed that
 * the previous ones completed. */
int mc3hip_p2p_rollback(mc3hip_ctx *ctx, int64_t *from_refresh, int64_t *to_refresh);
/* Drop the RCCL communicator of mc3hip_comm_init (fall back to the host-staged exchange). */
int mc3hip_comm_destroy(mc3hip_ctx *ctx);

/* Self tests used by tests/ on the GPU: evaluate the shared deterministic
 * functions on the device (op: 0 exp, 1 log, 2 exp2, 3 sin2pi, 4 cos2pi,
 * 5 sqrt, 6 1/x) and Philox4x32-10 ([n][4] counters, one key). */
int mc3hip_selftest_math(int32_t device, int32_t op, const double *in,
                         int32_t n, double *out);
int mc3hip_selftest_philox(int32_t device, const uint32_t *ctr, int32_t n,
                           const uint32_t key[2], uint32_t *out);

/* Diagnostic builds (-DMC3HIP_STAMPS) record the shader clock at five phases of
 * the last step-kernel launch (workgroup 0); the product build returns zeros. */
int mc3hip_debug_stamps(mc3hip_ctx *ctx, uint64_t *out16);

/* Launch ledger (test infrastructure; no counterpart in the reference, whose one code path is
 * mc3/chain.py:183-299): which of the library's kernel instances this process has launched, and
 * how often -- one line each, "<launches> <of which with no MC3HIP_* switch in the environment>
 * <offset of the instance's handle in libmc3hip.so> <device kernel name> [<switch> ...]"
 * (a loop specialised at run time: offset 0x0, name "jit:<kernel>").  The
 * test suite adds the ledgers of its processes up and fails if an instance the library carries
 * was never launched beside the oracle (tests/test_zz_instance_coverage.py).  At most cap - 1
 * bytes and a terminating 0 are written; *needed = the size of the whole text. */
int mc3hip_debug_launch_counts(char *buf, int64_t cap, int64_t *needed);
/* ... and the chain kernel the context launched last, labelled the same way ("" before the
 * first launch): the soak tests assert the kernel they mean to check is the one that ran. */
int mc3hip_debug_last_kernel(mc3hip_ctx *ctx, char *buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MC3HIP_H */
