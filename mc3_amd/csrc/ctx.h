// ctx.h -- the host-side context behind the C-ABI (include/mc3hip.h) and the seams between
// the translation units of libmc3hip.so.
//
// The library is built from several translation units compiled in parallel: mc3hip.hip (the
// C-ABI, the launch structure, the small kernels around the chain update) and one or more
// k_*.hip per kernel family, each of which instantiates its kernels and hands out launchers / kernel
// pointers through the functions declared at the end of this file.  No device code crosses a
// translation unit, so no relocatable device code is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <string>
#include <vector>

#include "core.h"

typedef struct ncclComm *ncclComm_t;

struct mc3hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    mc3hip::DevProblem P{};
    int model_kind = 0, mt = 0, wpc = 1;
    size_t smem = 0;
    int64_t zlen = 0;
    int64_t refresh_done = 0;
    bool populated = false;
    uint32_t flags = 0;
    std::vector<void *> allocs;
    char *arena = nullptr;
    size_t arena_used = 0;
    long long *ctl = nullptr;
    double *eval_params = nullptr, *eval_out = nullptr;
    int eval_cap = 0;
    double *model_out = nullptr;
    // graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_nodes = 0;
    int persist_refs = 0;           // refreshes per persistent launch, 0 = off
    long long capacity = 0;         // co-resident workgroups of the persistent instance that runs
    bool persist_checked = false;
    bool fast = false;              // lean persistent instance selected
    bool chainwave = false;         // wavelet likelihood, eight waves: a ninth runs the chain
                                    // (kernels_wavelet.h wavelet_cw_kernel)
    // 2 049 ... 16 384 points, the plain strict DEMC shape, an ensemble that is co-resident at
    // four / eight waves per chain: tiles the waves of a chain keep in registers between them
    // (kernels_lean.h lean_demc_wide_kernel NT; 0 = not this context).  Set at creation
    // (mc3hip.hip wide_tiles_shape), taken back by decide_persist if the grid does not fit.
    int wide_tiles = 0;
    // ... and of the same sizes with what those loops do not carry -- fixed / shared
    // coefficients, priors, the support distribution, the random walk: 2, 4 or 8 tiles in the
    // registers of FOUR waves (16 of EIGHT) of a loop specialised at run time (lean_*_body<.., NW = 4, NT>;
    // mc3hip_jit_request); 0 = not this context
    int wide_jit_tiles = 0;
    int wide_jit_waves = 0;         // (4 or 8)
    // wavelet likelihood, one wave per chain (<= 2048 points), the shapes of cw_lean_eligible,
    // the one-wave wavelet chains written out by hand: kernels_wavelet.h wavelet_one_kernel
    bool wl_one = false;
    int mc_chains = 0;              // chains a workgroup of the persistent kernel advances per
                                    // refresh (0: one workgroup per chain), its grid and the
    int mc_grid = 0;                // dynamic LDS with the chains' waiting state appended
    size_t mc_smem = 0;
    double *stage = nullptr;        // dense staging for Z row transfers (grow-only)
    size_t stage_doubles = 0;
    // rows of refreshes an mc3hip_sync has confirmed are read back on a stream of their own,
    // beside whatever runs on `stream` (mc3hip_get_rows_settled)
    hipStream_t copy_stream = nullptr;
    double *stage2 = nullptr;
    size_t stage2_doubles = 0;
    int64_t confirmed_done = 0;     // refreshes complete and not subject to a redo
    // checkpoint of the per-chain state taken before the first persistent launch that
    // has not been confirmed yet (recovery if a launch gives up)
    char *ckpt = nullptr;
    int64_t ckpt_R = 0;
    bool ckpt_valid = false;
    long long last_run0 = 0;
    int recoveries = 0;
    int tile_kmax = 0;              // tiled plan: coefficient capacity of the tile kernel
    int tile_chain_block = 0;       // chains per workgroup column of the tile kernel
    bool resident = false;          // tiled plan as one persistent launch (resident_tile_kernel)
    int resident_chain_block = 0;   // its column height
    // ... with the owners as waves of their own and the blocks of chains made up at run time
    // (resident_async_kernel): that many owner waves behind the eight tile waves, 0 = the
    // kernel in step
    int resident_owner_waves = 0;
    size_t rs_sched_words = 0;      // (its scheduler's rows, DevProblem::rs_sched: words allocated)
    bool ckpt_settled = false;      // peer-to-peer: the launches since the checkpoint completed
                                    // HERE; confirmed by the caller's next mc3hip_run
    double *moments = nullptr;      // scratch of mc3hip_chain_moments (2 x nchains x nfree)
    // comm
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    int64_t logpost_gathered = 0;   // refreshes whose log_post rows are already exchanged
    // RCCL exchange captured in a hipGraph: per refresh (step kernel, all-gather of ring
    // slot r, ring_in kernel) x comm_graph_nodes
    double *xring = nullptr;
    int comm_graph_nodes = 0;
    hipGraph_t cgraph = nullptr;
    hipGraphExec_t cgraph_exec = nullptr;
    bool comm_graph_failed = false;
    bool comm_warm = false;         // one eager collective has run (RCCL set up its channels)
    // run-time specialisation (mc3hip_jit_install): the hand-laid-out loop compiled for this
    // context's model
    hipModule_t jit_module = nullptr;
    hipFunction_t jit_fn = nullptr;
    // peer-to-peer hand-off
    int p2p_ranks = 0;
    std::vector<void *> p2p_opened;
    // the chain kernel this context launched last (launch ledger below; nullptr: none yet, or
    // the run-time specialisation, whose name is jit_name)
    const void *last_kernel = nullptr;
    std::string jit_name;
};

namespace mc3hip {
namespace host {

// every chain kernel that covers many refreshes has the signature (DevProblem, nref); the
// resident tiled kernels take their column height as well
typedef void (*chain_kernel_fn)(const DevProblem, const int);
typedef void (*resident_kernel_fn)(const DevProblem, const int, const int);

constexpr int WL_EVAL_BLOCKS = 256;

// ---- the environment, read in ONE place ------------------------------------------------------
// Every variable the library looks at goes through env_text().  The RELEASE library knows the
// variables DESIGN.md section 8a documents: sizes (MC3HIP_PERSIST_REFS, _PERSIST_CAPACITY,
// _RESIDENT_MAXCOL, _GRAPH_NODES), the test hooks (MC3HIP_TEST_*), and switches that send a
// shape down a path the library ALSO takes by itself for other shapes (the general chain
// instead of a hand-laid-out loop, the three-launch tiled plan instead of the resident one, the
// kernel in step instead of resident_async_kernel, ...) -- what the tests use to reach those
// paths with small problems.  The A/B switches of experiments (priorities, superseded kernel
// flavours, forced wave counts) exist only in a development build (-DMC3HIP_DEV, python -m
// mc3_amd.build --dev): in the release library dev_env_text() is a constant and the code behind
// such a switch is not compiled.
inline const char *env_text(const char *name) { return getenv(name); }
#ifdef MC3HIP_DEV
inline const char *dev_env_text(const char *name) { return env_text(name); }
#else
constexpr const char *dev_env_text(const char *) { return nullptr; }
#endif

// ---- launch ledger (mc3hip.hip): how often every kernel instance of the library was launched
// by this process, by the address of the kernel's handle -- mc3hip_debug_launch_counts reads it, the test suite's
// instance-coverage report (tests/test_zz_instance_coverage.py) is made from it.  A launch is
// one call per many thousand refreshes (or one graph capture): a mutex and a short table.
const void *ledger_note(const void *kernel);
// every kernel of the library is launched through this, so that none escapes the ledger
#define MC3_LAUNCH(kernel, ...)                                                            \
    do {                                                                                   \
        auto mc3_k_ = (kernel);                                                            \
        ::mc3hip::host::ledger_note(reinterpret_cast<const void *>(mc3_k_));               \
        hipLaunchKernelGGL(mc3_k_, __VA_ARGS__);                                           \
    } while (0)
// ... a chain kernel also leaves its name with the context (mc3hip_debug_last_kernel)
#define MC3_LAUNCH_CHAIN(c, kernel, ...)                                                   \
    do {                                                                                   \
        auto mc3_k_ = (kernel);                                                            \
        (c)->last_kernel = ::mc3hip::host::ledger_note(reinterpret_cast<const void *>(mc3_k_)); \
        hipLaunchKernelGGL(mc3_k_, __VA_ARGS__);                                           \
    } while (0)

// the wavelet kernel with a wave of its own for the chain (kernels_wavelet.h): nine waves, and
// behind the compute waves' LDS areas the chain wave's own area, the command word, the result
constexpr int CW_THREADS = 576;
constexpr int CW_EXTRA_DOUBLES = 64 + MC3HIP_MAX_PARAMS + 2 + 48 + 4 + 4;
// the shapes the hand-structured chain wave (lean_chain_wave) is written for; everything else
// runs the general chain behind the same kernel
inline bool cw_lean_eligible(const mc3hip_ctx *c)
{
    // (the declared multi-GPU cadence, zlag > 0: the general chain behind the same kernel)
    return c->P.thinning == 1 && c->P.trace_iters == 0 && c->P.nshare == 0 && c->P.ngprior == 0 &&
           c->P.zlag == 0;
}
// ... and those the hand-structured owner of the resident tiled plans is written for
inline bool rs_lean_eligible(const mc3hip_ctx *c)
{
    return c->P.sampler == MC3HIP_DEMC && c->P.thinning == 1 && c->P.trace_iters == 0 &&
           c->P.nshare == 0;
}
inline bool resident_owner_lean(const mc3hip_ctx *c)
{
    return rs_lean_eligible(c) && !env_text("MC3HIP_NO_RS_LEAN");
}
// ... and the shapes resident_async_kernel takes (the grid permitting: decide_persist): design
// matrices through the matrix cores -- any sampler, thinning, shared parameters (the owners
// hand-structured where resident_owner_lean says so, the general chain otherwise)
inline bool resident_async_shape(const mc3hip_ctx *c)
{
    // (four blocks of operand rows beside the general chain: an operand spilled and reloaded at
    // the head of every block -- the kernel in step serves those)
    return c->mt == MT_LINEAR && c->P.tile == 2048 && !env_text("MC3HIP_NO_RS_ASYNC") &&
           (resident_owner_lean(c) || (c->P.nmodel + 4) / 4 <= 3);
}
inline bool chainwave_runs(const mc3hip_ctx *c, bool p2p)
{
    // (several chains per workgroup: only the hand-structured chain wave takes turns)
    return c->chainwave && !p2p && (c->mc_chains == 0 || c->P.cw_lean != 0);
}
// several chains per workgroup, snooker or mrw: the pipelined kernel with a tenth wave for the
// tail (kernels_wavelet.h wavelet_pipe_kernel); same LDS as the nine-wave kernel
inline bool chainwave_pipe(const mc3hip_ctx *c, bool p2p)
{
    return chainwave_runs(c, p2p) && c->mc_chains > 0 && c->P.sampler != MC3HIP_DEMC;
}
inline int chainwave_threads(const mc3hip_ctx *c, bool p2p)
{
    return chainwave_pipe(c, p2p) ? CW_THREADS + 128 : CW_THREADS;
}

// Waves per chain are dispatched as {1, 4, 16}; the wavelet geometries have 1, 4 or 8:
// the widest one rides on the dispatch slot of the widest Gaussian one.
template <int WPC>
constexpr int wavelet_waves() { return WPC == 16 ? 8 : WPC; }

// ---- which instance serves a context: pure functions of the context ---------------------
// polynomials of 2..4 coefficients: the tile kernel with the degree compiled in
inline int tile_exact_degree(const mc3hip_ctx *c)
{
    return (c->mt != MT_LINEAR && c->tile_kmax == 4 && c->P.nmodel >= 2) ? c->P.nmodel : 0;
}

// design-matrix models with two points per virtual lane (5..16 coefficients): the data
// sums go through the FP64 matrix cores (tile_chisq_mfma_kernel)
inline bool tile_uses_mfma(const mc3hip_ctx *c)
{
    // (2..15 coefficients: with the data as one more operand row, at most four blocks of four)
    return c->mt == MT_LINEAR && c->P.tile == 2048 && c->P.nmodel >= 2 && c->P.nmodel <= 15 &&
           c->P.aw != nullptr;
}

// resident tiled plan: design-matrix models through the matrix cores, or polynomials of
// 2..4 coefficients
inline bool resident_poly(const mc3hip_ctx *c)
{
    return c->mt != MT_LINEAR && c->P.tile == 4096 && tile_exact_degree(c) >= 2;
}

// DEMC whose partner rows are (mostly) published long before they are needed -- the declared
// multi-GPU cadence (zlag >= 1: rows of refresh R-1-zlag), thinning > 1 (the iterations of a
// refresh look at the same rows) -- has a hand-laid-out loop of its own, lean_demc_lag_kernel:
// one tile, one chain per wave, any thinning, fixed / shared coefficients and priors, one GPU
// or peer to peer with a courier wave
inline bool lag_demc_shape(const mc3hip_ctx *c)
{
    const DevProblem &P = c->P;
    // (... and of the strict lockstep with thinning > 1, zlag = 0: only the first iteration of
    // a refresh then asks for rows that are just being published and waits for them)
    return (P.zlag > 0 || P.thinning > 1) && P.sampler == MC3HIP_DEMC && P.npad <= VLANES &&
           P.zs == ZROW && !env_text("MC3HIP_NO_LEAN");
}

// The model tag a run-time specialisation of the one-wave loops would be compiled for, or 0:
// members of the declared family the library has no instance for -- polynomials of 5..8
// coefficients, design matrices of up to 10 columns -- that fit ONE register tile (ndata <=
// 1024), every parameter a coefficient, Gaussian chisq, one GPU, a sampler flavour that exists
// as a hand-laid-out loop (mc3hip.hip mc3hip_jit_request names it).
inline int jit_model_tag(const mc3hip_ctx *c)
{
    const DevProblem &P = c->P;
    if (P.wlike || P.trace_iters != 0 || P.tile > 0 ||
        (c->flags & MC3HIP_FLAG_P2P) || c->p2p_ranks > 1 || c->comm != nullptr ||
        P.zs != ZROW || P.nmodel != P.nparams || P.nfree < 1 ||
        env_text("MC3HIP_NO_LEAN") || dev_env_text("MC3HIP_NO_FAST"))
        return 0;
    // more than 1024 points: polynomials of 2..4 coefficients, the tiles in the registers of
    // four waves (the shapes lean_demc_wide_kernel / lean_snooker_wide_kernel do not carry)
    // (... and of the models that are specialised anyway: polynomials of 5..8 coefficients,
    // design matrices of up to ten columns as far as a thread's share of the tiles -- K + 2
    // values per point -- fits its registers: mc3hip.hip wide_jit_geometry)
    if (c->wide_jit_tiles > 0) {
        // (thinning > 1: the snooker and random-walk loops have a flavour for it -- not with the
        // support distribution; DEMC's is the one-tile loop of the declared cadence)
        const bool thin_ok = P.thinning == 1 ||
                             (P.sampler == MC3HIP_MRW) ||
                             (P.sampler == MC3HIP_SNOOKER && P.fepsilon == 0.0) ||
                             (P.sampler == MC3HIP_DEMC && P.fepsilon == 0.0);
        if (!thin_ok || P.zlag != 0 || c->mc_chains != 0)
            return 0;
        if (c->model_kind == MC3HIP_MODEL_POLY && P.nmodel >= 2 && P.nmodel <= 16)
            return P.nmodel;
        if (c->model_kind == MC3HIP_MODEL_LINEAR && P.nmodel >= 1 && P.nmodel <= 16)
            return MT_LINK0 + P.nmodel;
        return 0;
    }
    // What is left of the polynomials of 2..4 coefficients (the library's own instances: the
    // plain shapes, and one tile with fixed coefficients / priors at thinning 1): ONE wave,
    //  * one tile, fixed coefficients / priors AND thinning > 1 (snooker, random walk);
    //  * 1 025 ... 4 096 points (two tiles in registers, the others streamed: MULTI) with fixed
    //    coefficients / priors or thinning > 1, where the ensemble is too large for the tiles to
    //    sit in the registers of four waves per chain (snooker and the random walk; DEMC with
    //    thinning 1 and without the support distribution).
    if (c->model_kind == MC3HIP_MODEL_POLY && P.nmodel >= 2 && P.nmodel <= 4 && c->wpc == 1 &&
        P.zlag == 0 && c->mc_chains == 0 && P.npad <= 4 * VLANES) {
        const bool plain = P.nfree == P.nparams && P.ngprior == 0 && P.nshare == 0;
        const bool other = P.sampler == MC3HIP_MRW ||
                           (P.sampler == MC3HIP_SNOOKER && P.fepsilon == 0.0);
        if (P.npad <= VLANES)
            return (!plain && P.thinning > 1 && other) ? P.nmodel : 0;
        if ((!plain || P.thinning > 1) &&
            (other || (P.sampler == MC3HIP_DEMC && P.thinning == 1 && P.fepsilon == 0.0)))
            return P.nmodel;
        return 0;
    }
    if (c->wpc != 1 || P.npad > VLANES)
        return 0;
    int tag = 0;
    // (as many coefficients as a row of Z has lanes: 16 -- round 5; before: 8 / 10)
    if (c->model_kind == MC3HIP_MODEL_POLY && P.nmodel >= 5 && P.nmodel <= 16)
        tag = P.nmodel;
    else if (c->model_kind == MC3HIP_MODEL_LINEAR && P.nmodel >= 1 && P.nmodel <= 16)
        tag = MT_LINK0 + P.nmodel;
    else
        return 0;
    const bool plain = P.nfree == P.nparams && P.ngprior == 0 && P.nshare == 0;
    switch (P.sampler) {
    // (the support distribution's deviates and the random walk's: a Philox slot per two
    // coefficients, as many as the model has -- round 5; before: four at most)
    case MC3HIP_DEMC:          // (the support distribution: the strict headline loop only)
        return (P.fepsilon == 0.0 || (P.thinning == 1 && P.zlag == 0)) ? tag : 0;
    case MC3HIP_SNOOKER:
        return ((P.fepsilon == 0.0 && (P.thinning == 1 || plain)) ||
                (P.fepsilon != 0.0 && P.thinning == 1 && P.zlag == 0)) ? tag : 0;
    default:
        return (P.thinning == 1 || plain) ? tag : 0;
    }
}
// waves of the specialised loop's workgroup: helper waves share the data sum (gauss.h
// SumHelpers) where one wave's registers hardly hold the tile -- design matrices of five
// columns and more (K + 2 values per point, 16 points per lane) on the strict DEMC and the
// snooker loops.  MC3HIP_JIT_WAVES=1|2|4 overrides (an experiment; results do not change).
inline int jit_waves(const mc3hip_ctx *c)
{
    const int tag = jit_model_tag(c);
    const DevProblem &P = c->P;
    if (tag != 0 && c->wide_jit_tiles > 0)
        return c->wide_jit_waves;
    if (tag != 0 && tag <= 4)                   // (the leftovers of the small polynomials)
        return 1;
    // (the random walk's loop has had helper waves since round 5: from eleven columns on -- up
    // to ten its one wave is as fast)
    if (tag < MT_LINK0 || (P.sampler == MC3HIP_MRW && tag - MT_LINK0 <= 10) ||
        (P.sampler == MC3HIP_DEMC && (P.thinning > 1 || P.zlag > 0)))
        return 1;
    if (const char *e = dev_env_text("MC3HIP_JIT_WAVES")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4)
            return v;
    }
    const int K = tag - MT_LINK0;
    // (measured, 256 chains x 1000 points with three priors: five columns 157 / 174 / 181 M
    // chain-steps/s with 1 / 2 / 4 waves, ten columns 112 / 136 / 157)
    return K >= 5 ? 4 : 1;
}
inline bool jit_serves(const mc3hip_ctx *c)
{
    return c->jit_fn != nullptr && jit_model_tag(c) != 0;
}

inline int wide_tiles(const mc3hip_ctx *c);

// the lean persistent instance applies when nothing of the general machinery is
// needed (run_chain.h run_chain FAST)
inline bool fast_eligible(const mc3hip_ctx *c)
{
    const DevProblem &P = c->P;
    // (fixed coefficients and Gaussian priors are fine: the FX kernels map coefficient j to
    // its free parameter or to its fixed value, handoff.h coefficient_map;
    // instantiated for one GPU and thinning = 1 only: FX of persist_fast_kernel)
    const bool plain = P.nparams == P.nfree && P.ngprior == 0 && P.nshare == 0;
    // (peer to peer: the hand-laid-out snooker / random-walk loops with their courier wave,
    // k_lean4.hip, and the loop of the declared cadence, k_lean3.hip -- the lean run_chain
    // instances have no FX x P2P flavour)
    const bool p2p_flag = (c->flags & MC3HIP_FLAG_P2P) != 0;
    const bool lean_p2p_other = (P.sampler == MC3HIP_SNOOKER || P.sampler == MC3HIP_MRW) &&
                                P.npad <= VLANES && P.zs == ZROW && !env_text("MC3HIP_NO_LEAN");
    const bool fx_ok = P.nfree >= 1 &&
                       ((P.thinning == 1 && (!p2p_flag || lean_p2p_other)) || lag_demc_shape(c));
    // more than one 1024-point tile (streamed): plain problems, one GPU, thinning = 1
    const bool multi = P.npad > VLANES;
    const bool size_ok = !multi || (plain && fx_ok && P.tile == 0);
    // (the support distribution, fepsilon != 0: the headline loop has an EPS flavour --
    // lean_demc_kernel: DEMC, strict lockstep, thinning 1, one tile, one GPU)
    // ... and so has the snooker loop, for up to four free parameters (two Philox slots)
    const bool eps_ok = P.fepsilon == 0.0 || P.sampler == MC3HIP_MRW ||   // (chain.py:219-220)
                        (P.sampler == MC3HIP_DEMC && P.thinning == 1 && P.zlag == 0 && !p2p_flag &&
                         P.nfree <= 4 && !multi && P.zs == ZROW && !env_text("MC3HIP_NO_LEAN")) ||
                        (P.sampler == MC3HIP_SNOOKER && P.thinning == 1 && P.nfree <= 4 &&
                         !p2p_flag && !multi && P.zs == ZROW && !env_text("MC3HIP_NO_LEAN"));
    return (c->wpc == 1 || wide_tiles(c) > 0) && !P.wlike && size_ok && c->mt >= 2 && c->mt <= 4 &&
           (plain || fx_ok) && P.nmodel == P.nparams &&
           eps_ok && P.trace_iters == 0 &&
           (P.sampler == MC3HIP_DEMC || P.sampler == MC3HIP_SNOOKER ||
            P.sampler == MC3HIP_MRW) &&
           !env_text("MC3HIP_NO_LEAN");         // (the switch: the general chain for everything)
}

inline bool lean_rows_shape(const mc3hip_ctx *c);

// shapes that ONLY a one-chain-per-wave hand-laid-out loop serves leanly (the lean run_chain
// instances have no such flavour): an ensemble of them beyond the co-resident waves runs the
// general chain (mc3hip.hip decide_persist)
inline bool lean_one_chain_only(const mc3hip_ctx *c)
{
    const DevProblem &P = c->P;
    const bool plain = P.nfree == P.nparams && P.ngprior == 0 && P.nshare == 0;
    // (peer to peer with thinning 1: kernels_rows.h has the FX flavour)
    return jit_serves(c) || (P.fepsilon != 0.0 && P.sampler != MC3HIP_MRW) ||
           (lag_demc_shape(c) && !plain &&
            (P.thinning != 1 || ((c->flags & MC3HIP_FLAG_P2P) && !lean_rows_shape(c))));
}

// the headline shape (DEMC, thinning 1) has a loop of its own: lean_demc_kernel
inline bool lean_demc_eligible(const mc3hip_ctx *c)
{
    return c->fast && c->mc_chains == 0 && c->P.sampler == MC3HIP_DEMC && c->P.thinning == 1 &&
           c->P.zlag == 0 && c->P.nparams == c->mt && c->P.zs == ZROW && !env_text("MC3HIP_NO_LEAN");
}
// ... and so has the declared multi-GPU cadence: lean_demc_lag_kernel
inline bool lean_demc_lag_eligible(const mc3hip_ctx *c)
{
    return c->fast && c->mc_chains == 0 && lag_demc_shape(c) && c->P.nparams == c->mt;
}

// ... in the peer-to-peer mode with a second wave that carries the rows to the other ranks'
// replicas (lean_demc_kernel COURIER; MC3HIP_NO_COURIER=1: the chain wave stores them itself)
inline bool lean_plain_shape(const mc3hip_ctx *c);
inline bool lean_demc_lag_eligible(const mc3hip_ctx *c);
inline bool lean_snooker_eligible(const mc3hip_ctx *c);
inline bool lean_mrw_eligible(const mc3hip_ctx *c);
inline bool lean_courier(const mc3hip_ctx *c, bool p2p)
{
    if (!p2p)       // (one GPU: a courier wave was measured slower -- docs/HISTORY.md, round 3)
        return false;
    return (lean_demc_eligible(c) || lean_demc_lag_eligible(c) || lean_snooker_eligible(c) ||
            lean_mrw_eligible(c)) &&
           !(c->flags & MC3HIP_FLAG_NO_COURIER) && !env_text("MC3HIP_NO_COURIER");
}

// every coefficient free, no Gaussian priors, no shared parameters (otherwise: the FX instances)
inline bool lean_plain_shape(const mc3hip_ctx *c)
{
    return c->P.nfree == c->P.nparams && c->P.ngprior == 0 && c->P.nshare == 0;
}

// ... and the snooker and the random-walk samplers (one tile, one chain per wave, one GPU):
// lean_snooker_kernel, lean_mrw_kernel
inline bool lean_other_sampler_eligible(const mc3hip_ctx *c, int sampler)
{
    const DevProblem &P = c->P;
    // (thinning > 1: the THIN instances, plain shapes only)
    const bool plain = P.nfree == P.nparams && P.ngprior == 0 && P.nshare == 0;
    // (more than one tile: the MULTI instances, plain shapes with thinning 1 only)
    // (peer to peer, with a courier wave: the one-tile instances)
    const bool p2p = (c->flags & MC3HIP_FLAG_P2P) || c->p2p_ranks > 1;
    return c->fast && c->mc_chains == 0 && P.sampler == sampler &&
           (P.thinning == 1 || plain) &&
           (P.npad <= VLANES || (plain && P.thinning == 1 && !p2p)) &&
           P.nparams == c->mt && P.zs == ZROW && !env_text("MC3HIP_NO_LEAN");
}
inline bool lean_snooker_eligible(const mc3hip_ctx *c)
{
    return lean_other_sampler_eligible(c, MC3HIP_SNOOKER);
}
inline bool lean_mrw_eligible(const mc3hip_ctx *c)
{
    return lean_other_sampler_eligible(c, MC3HIP_MRW) && c->mt <= 4;
}

// ... and the same shape with two tiles and two or three chains per wave, one after another:
// lean_demc_mc_kernel (the one-tile shapes of every sampler: kernels_rows.h, below)
inline bool lean_demc_mc_eligible(const mc3hip_ctx *c)
{
    // (four chains per wave AND two register-resident tiles do not fit the registers)
    const DevProblem &P = c->P;
    return c->fast && c->mc_chains >= 2 && c->mc_chains <= 3 && P.npad > VLANES &&
           P.nfree == P.nparams && P.ngprior == 0 && P.nshare == 0 &&
           P.sampler == MC3HIP_DEMC && P.thinning == 1 && P.zlag == 0 &&
           P.nparams == c->mt && P.zs == ZROW && !(c->flags & MC3HIP_FLAG_P2P) &&
           c->p2p_ranks <= 1 && !env_text("MC3HIP_NO_LEAN");
}

// ... and, for the one-tile shapes of those, with the
// chains of a wave side by side in its rows of 16 lanes: kernels_rows.h lean_*_rows_kernel
// (MC3HIP_NO_ROWS: the general lean instance takes turns, run_chain MC)
// (the shapes those loops serve -- on one GPU or sharded over several, peer to peer -- whatever
// the number of chains per wave decide_persist settles on)
inline bool lean_rows_shape(const mc3hip_ctx *c)
{
    const DevProblem &P = c->P;
    return P.thinning == 1 && P.nparams == c->mt && c->mt >= 2 && c->mt <= 4 && P.zs == ZROW &&
           P.npad <= VLANES && P.nfree >= 1 && (P.fepsilon == 0.0 || P.sampler == MC3HIP_MRW) &&
           !env_text("MC3HIP_NO_LEAN");
}
inline bool lean_rows_eligible(const mc3hip_ctx *c)
{
    // (every sampler, strict or in the declared cadence)
    return c->fast && c->mc_chains >= 2 && c->mc_chains <= 4 && lean_rows_shape(c);
}

// waves that share the data sum of a chain in the hand-laid-out loops: four (eight) where the
// tiles of more than 1024 points sit in their registers (kernels_lean.h lean_demc_wide_kernel,
// lean_snooker_wide_kernel), one otherwise
inline int wide_tiles(const mc3hip_ctx *c)
{
    return (c->wide_tiles > 0 && !(c->flags & MC3HIP_FLAG_P2P) && c->p2p_ranks <= 1 &&
            c->comm == nullptr && c->mc_chains == 0)
               ? c->wide_tiles : 0;
}
inline int lean_waves(const mc3hip_ctx *c)
{
    if (const int nt = wide_tiles(c))
        if (c->fast)
            return nt > 8 ? 8 : 4;
    // (helper waves for the ONE-tile headline loop -- two / four waves sharing a chain's data sum
    // -- were measured slower than one at every ensemble size, profiles/r03_lean_waves.jsonl,
    // and went with round 6)
    return 1;
}

// ---- k_general_w{1,4,16}.hip: one workgroup per chain, Gaussian chisq, by waves per chain ---
void launch_step_gauss_w1(mc3hip_ctx *c, int roff);
void launch_step_gauss_w4(mc3hip_ctx *c, int roff);
chain_kernel_fn persist_instance_gauss_w1(const mc3hip_ctx *c, bool p2p);
chain_kernel_fn persist_instance_gauss_w4(const mc3hip_ctx *c, bool p2p);
void launch_eval_gauss_w1(mc3hip_ctx *c, int n);
void launch_eval_gauss_w4(mc3hip_ctx *c, int n);

// ---- k_wavelet_{w1,w4,w8,large}.hip: the same with the wavelet likelihood, by geometry ----
#define MC3_WAVELET_UNIT(tag)                                                       \
    void launch_step_wavelet_##tag(mc3hip_ctx *c, int roff);                        \
    chain_kernel_fn persist_instance_wavelet_##tag(const mc3hip_ctx *c, bool p2p);  \
    void launch_eval_wavelet_##tag(mc3hip_ctx *c, int n);                           \
    hipError_t set_smem_attr_wavelet_##tag(mc3hip_ctx *c);
MC3_WAVELET_UNIT(w1)
MC3_WAVELET_UNIT(w1b)
MC3_WAVELET_UNIT(w4)
MC3_WAVELET_UNIT(w8)
MC3_WAVELET_UNIT(large)
#undef MC3_WAVELET_UNIT
// (k_wavelet_w8mc.hip: the nine-wave kernel with several chains per workgroup)
chain_kernel_fn persist_instance_wavelet_w8mc(const mc3hip_ctx *c);
chain_kernel_fn persist_instance_wavelet_w8pp(const mc3hip_ctx *c);   // (pipelined: ten waves)
chain_kernel_fn persist_instance_wavelet_w8cw(const mc3hip_ctx *c);
chain_kernel_fn persist_instance_wavelet_w8cg(const mc3hip_ctx *c);

// ---- k_lean.hip: the lean one-wave persistent kernels ------------------------------------
// ONE function picks the instance a context runs, so that the launch and the occupancy query
// that sizes the persistent grid can never disagree (the MULTI / FX instances use more than
// twice the registers of the plain ones: 4 single-wave workgroups per CU instead of 8).
chain_kernel_fn fast_instance(const mc3hip_ctx *c, bool p2p);
chain_kernel_fn lean_other_instance(const mc3hip_ctx *c);   // k_lean2.hip
chain_kernel_fn lean_lag_instance(const mc3hip_ctx *c, bool p2p);   // k_lean3.hip
chain_kernel_fn lean_other_p2p_instance(const mc3hip_ctx *c);       // k_lean4.hip
chain_kernel_fn lean_demc_mc_instance(const mc3hip_ctx *c, bool p2p);   // k_lean5.hip
chain_kernel_fn lean_rows_instance(const mc3hip_ctx *c, bool p2p);      // k_lean6.hip
chain_kernel_fn lean_rows_instance_p2p(const mc3hip_ctx *c);            // k_lean7.hip

// ---- k_tiled.hip: the tiled plan as three launches per iteration -------------------------
void launch_propose(mc3hip_ctx *c, int roff, int j);
void launch_finish(mc3hip_ctx *c, int roff, int j);
void launch_tile(mc3hip_ctx *c);
hipError_t tile_set_attr(mc3hip_ctx *c);

// ---- k_resident.hip: the tiled plan as one persistent launch -----------------------------
resident_kernel_fn resident_kernel(const mc3hip_ctx *c, bool p2p);
size_t resident_lds_bytes(const mc3hip_ctx *c);
// (the kernel that runs, its workgroup: the one in step, or resident_async_kernel when
// c->resident_owner_waves > 0)
resident_kernel_fn resident_async_kernel_of(const mc3hip_ctx *c, bool p2p);
size_t resident_async_lds_bytes(const mc3hip_ctx *c);

}  // namespace host
}  // namespace mc3hip
