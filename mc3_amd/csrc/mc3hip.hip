// mc3hip.hip -- C-ABI (include/mc3hip.h) over the chain-update kernels.
//
// Host side of the drop-in boundary: owns the device copies of what the reference
// keeps in multiprocessing shared memory (mc3/mcmc_driver.py:143-202) and drives the
// chain update on a private HIP stream in one of three launch structures
// (decide_persist, mc3hip_run):
//   - persistent: ONE launch for up to 16384 Z refreshes, the new rows handed between
//     workgroups inside the launch (single GPU, or several GPUs through
//     mc3hip_p2p_attach); checkpointed, recoverable (settle());
//   - per refresh: one fused step kernel per refresh, batched into a hipGraph, or
//     eagerly with one in-place RCCL all-gather after each (librccl is dlopen'ed on
//     first use, so single-GPU users do not need it);
//   - tiled: proposal / per-tile data sums for all chains / accept as three launches
//     per iteration for large data sets.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ctx.h"
#include "post.h"

using namespace mc3hip;
using namespace mc3hip::host;

// ---- launch ledger (ctx.h) -------------------------------------------------------------------
extern char **environ;
namespace {
struct LedgerRow {
    const void *kernel;          // handle of an instance of this library, or nullptr:
    std::string name;            // ... a kernel specialised at run time, by name
    unsigned long long launches; // all launches, and those with no MC3HIP_* switch in the
    unsigned long long plain;    // environment: what the dispatch picks when left alone
    std::string switches;        // the switches seen beside the others (" NAME NAME ...")
};
std::mutex g_ledger_lock;
std::vector<LedgerRow> g_ledger;

// the MC3HIP_* variables of the environment that steer the dispatch (everything but the test
// suite's own plumbing), " NAME" each
std::string ledger_switches()
{
    std::string out;
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "MC3HIP_", 7) != 0)
            continue;
        const char *eq = strchr(*e, '=');
        const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
        // (not switches of the dispatch: the tests' plumbing, the rendezvous of the ranks, the
        // hooks that let several ranks share ONE device or make a launch give up, the cadence)
        static const char *const plumbing[] = {
            "MC3HIP_LEDGER_DIR", "MC3HIP_LIBRARY", "MC3HIP_JIT_CACHE", "MC3HIP_RDV_",
            "MC3HIP_BENCH_", "MC3HIP_COMM_TIMEOUT", "MC3HIP_VERBOSE", "MC3HIP_ZLAG",
            "MC3HIP_RESIDENT_MAXCOL", "MC3HIP_PERSIST_CAPACITY", "MC3HIP_TEST_"};
        bool skip = false;
        for (const char *p : plumbing)
            skip = skip || name.compare(0, strlen(p), p) == 0;
        if (skip)
            continue;
        out += " " + name;
    }
    return out;
}

void ledger_count(LedgerRow &r)
{
    const std::string sw = ledger_switches();
    ++r.launches;
    if (sw.empty()) {
        ++r.plain;
        return;
    }
    size_t at = 0;
    while (at < sw.size()) {                         // (each " NAME" once)
        const size_t end = sw.find(' ', at + 1);
        const std::string one = sw.substr(at, end == std::string::npos ? end : end - at);
        if ((r.switches + " ").find(one + " ") == std::string::npos)
            r.switches += one;
        at = end;
    }
}

void ledger_note_named(const std::string &name)
{
    std::lock_guard<std::mutex> hold(g_ledger_lock);
    for (LedgerRow &r : g_ledger)
        if (!r.kernel && r.name == name)
            return ledger_count(r);
    g_ledger.push_back(LedgerRow{nullptr, name, 0, 0, std::string()});
    ledger_count(g_ledger.back());
}
}  // namespace

const void *mc3hip::host::ledger_note(const void *kernel)
{
    std::lock_guard<std::mutex> hold(g_ledger_lock);
    for (LedgerRow &r : g_ledger)
        if (r.kernel == kernel) {
            ledger_count(r);
            return kernel;
        }
    g_ledger.push_back(LedgerRow{kernel, std::string(), 0, 0, std::string()});
    ledger_count(g_ledger.back());
    return kernel;
}

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                               \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess)                                                       \
            return fail(MC3HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,             \
                        hipGetErrorString(e_), __FILE__, __LINE__);                 \
    } while (0)

// ---- RCCL through dlopen -----------------------------------------------------
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
constexpr int NCCL_FLOAT64 = 8;   // ncclDataType_t: ncclDouble

int load_rccl()
{
    if (g_rccl.handle)
        return MC3HIP_OK;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h)
            break;
    }
    if (!h)
        return fail(MC3HIP_ERR_COMM, "cannot dlopen librccl.so: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(h, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather ||
        !g_rccl.CommDestroy)
        return fail(MC3HIP_ERR_COMM, "librccl.so lacks an expected symbol");
    g_rccl.handle = h;
    return MC3HIP_OK;
}

// a scratch allocation that grows with the problem must fit what the device has left: say so
// in the problem's terms instead of failing inside hipMalloc
int check_device_room(size_t bytes, const char *what)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
        return MC3HIP_OK;                   // (cannot tell: let the allocation speak)
    if (bytes > free_b)
        return fail(MC3HIP_ERR_ARG, "%s needs %.2f GiB of device memory, %.2f GiB are free",
                    what, (double)bytes / (1u << 30), (double)free_b / (1u << 30));
    return MC3HIP_OK;
}

int choose_wpc(int nlocal, int ndata, int wlike)
{
    // One wave per chain when there are enough chains to fill the chip or the
    // data sum is short; otherwise split a chain's data over 4 waves.
    const char *env = env_text("MC3HIP_WPC");
    if (wlike) {
        // blocked D4 pyramid (wavelet.h block_wavelet): at most 32 points per thread;
        // one wave per chain up to 2048 points (no barrier at all), four at 4096, eight
        // above (two per SIMD: a wave alone issues FP64 at half rate).  The geometry is a
        // function of ndata alone; the summation order follows it, mc3hip_get_plan reports
        // it and the checker follows.
        (void)env;
        return (ndata > 4096) ? 8 : (ndata > 2048 ? 4 : 1);
    }
    if (env) {
        // (four waves only for more than one tile: their one-tile instances -- never chosen
        // without this switch -- went with round 6)
        const int v = atoi(env);
        if (v == 1 || (v == 4 && ndata > VLANES))
            return v;
    }
    if (ndata <= 4096)
        return 1;
    if (nlocal >= 1024)
        return 1;
    // (sixteen waves per chain -- 1024 threads, 128 registers each: every instance spilled --
    // went with round 4; few chains over very many points belong to the tiled plan)
    return 4;
}

}  // namespace


namespace {

// Small arrays are carved out of one arena so that everything a step kernel
// touches besides Z sits in a few adjacent pages (each separate hipMalloc is its
// own page-table walk at the start of every launch).
constexpr size_t ARENA_BYTES = 4u << 20, ARENA_MAX_ITEM = 256u << 10;

template <typename T>
int dev_alloc(mc3hip_ctx *c, T **ptr, size_t count)
{
    const size_t bytes = (count ? count : 1) * sizeof(T);
    if (bytes <= ARENA_MAX_ITEM) {
        if (!c->arena) {
            void *a = nullptr;
            HIP_TRY(hipMalloc(&a, ARENA_BYTES));
            c->allocs.push_back(a);
            c->arena = static_cast<char *>(a);
            c->arena_used = 0;
        }
        const size_t off = (c->arena_used + 255) & ~size_t(255);
        if (off + bytes <= ARENA_BYTES) {
            *ptr = reinterpret_cast<T *>(c->arena + off);
            c->arena_used = off + bytes;
            return MC3HIP_OK;
        }
    }
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    c->allocs.push_back(p);
    *ptr = static_cast<T *>(p);
    return MC3HIP_OK;
}

template <typename T>
int dev_upload(mc3hip_ctx *c, const T **dst, const T *src, size_t count)
{
    T *p = nullptr;
    int rc = dev_alloc(c, &p, count);
    if (rc)
        return rc;
    if (count)
        HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = p;
    return MC3HIP_OK;
}


// one refresh of the tiled plan: thinning x (propose, tile sums, finish)
void launch_refresh_tiled(mc3hip_ctx *c, int roff)
{
    for (int j = 0; j < c->P.thinning; ++j) {
        launch_propose(c, roff, j);
        launch_tile(c);
        launch_finish(c, roff, j);
    }
}

// ---- resident tiled plan (resident.h ResidentTile / ResidentPoly) ------------------------
// the problem has the shape a resident kernel is written for (whether its grid fits the
// device is decide_persist's question)
bool resident_shape(const mc3hip_ctx *c)
{
    const bool mfma = tile_uses_mfma(c) && c->P.nmodel >= 2 && c->P.nmodel <= 15;
    return (mfma || resident_poly(c)) && c->P.rs_prop && c->comm == nullptr &&
           !env_text("MC3HIP_NO_RESIDENT");
}

void launch_resident(mc3hip_ctx *c, int nref)
{
    const int cb = c->resident_chain_block;
    const int ncol = (c->P.nlocal + cb - 1) / cb;
    if (c->resident_owner_waves > 0) {
        // (the scheduler's blocks of this launch are told from any other's by its number)
        c->P.rs_gen = (c->P.rs_gen == 0x7fffffff) ? 1 : c->P.rs_gen + 1;
        MC3_LAUNCH_CHAIN(c, resident_async_kernel_of(c, c->p2p_ranks > 1), dim3(c->P.ntiles, ncol),
                           dim3(512 + 64 * (c->resident_owner_waves + 1)),
                           resident_async_lds_bytes(c), c->stream, c->P, nref, cb);
        return;
    }
    MC3_LAUNCH_CHAIN(c, resident_kernel(c, c->p2p_ranks > 1), dim3(c->P.ntiles, ncol), dim3(512),
                       resident_lds_bytes(c), c->stream, c->P, nref, cb);
}

// dynamic LDS of the lean kernels: the hand-laid-out loop uses none
size_t fast_smem(const mc3hip_ctx *c)
{
    return ((c->P.sampler != MC3HIP_MRW && lean_demc_eligible(c)) || lean_demc_lag_eligible(c))
               ? 0 : c->smem;
}

// (several chains per workgroup, run_chain MC: a smaller grid, the chains' waiting state in
// dynamic LDS behind the kernel's own areas; decide_persist sets the three numbers)
// threads of a workgroup of the lean instance that runs: one wave, except where helper waves
// share a chain's data sum (ctx.h lean_waves)
int fast_threads(const mc3hip_ctx *c, bool p2p)
{
    if (lean_courier(c, p2p))
        return 128;                              // (the chain wave and its courier)
    return (p2p || c->mc_chains > 0) ? 64 : 64 * lean_waves(c);
}

void launch_fast(mc3hip_ctx *c, int nref)
{
    if (jit_serves(c)) {
        // the loop compiled for this context's model (mc3hip_jit_install): one wave per chain
        DevProblem P = c->P;
        int n = nref;
        void *args[] = {&P, &n};
        ledger_note_named(c->jit_name);
        c->last_kernel = nullptr;
        (void)hipModuleLaunchKernel(c->jit_fn, (unsigned)c->P.nlocal, 1, 1,
                                    (unsigned)(64 * jit_waves(c)), 1, 1, 0, c->stream,
                                    args, nullptr);
        return;
    }
    const bool mc = c->mc_chains > 0, p2p = c->p2p_ranks > 1;
    MC3_LAUNCH_CHAIN(c, fast_instance(c, p2p), dim3(mc ? c->mc_grid : c->P.nlocal),
                       dim3(fast_threads(c, p2p)), mc ? c->mc_smem : fast_smem(c), c->stream,
                       c->P, nref);
}

int blocks_per_cu(const void *kernel, int threads, size_t smem)
{
    if (!kernel)                     // (no persistent instance serves this context)
        return 0;
    int nb = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, smem) ==
                   hipSuccess ? nb : 0;
}

// co-resident single-wave workgroups per CU of the instance that will run; a context
// created for the peer-to-peer mode is sized for both its instances (it runs the plain
// one until the peers are attached, and again after a roll-back)
int fast_blocks_per_cu(mc3hip_ctx *c)
{
    if (jit_serves(c)) {
        int nb = 0;
        return hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, c->jit_fn,
                                                                  64 * jit_waves(c), 0) ==
                       hipSuccess ? nb : 0;
    }
    const bool p2p = c->p2p_ranks > 1;
    int nb = blocks_per_cu(reinterpret_cast<const void *>(fast_instance(c, p2p)),
                           fast_threads(c, p2p), fast_smem(c));
    if ((c->flags & MC3HIP_FLAG_P2P) && !p2p) {
        const int np = blocks_per_cu(reinterpret_cast<const void *>(fast_instance(c, true)),
                                     fast_threads(c, true), fast_smem(c));
        nb = np < nb ? np : nb;
    }
    return nb;
}

template <int MT>
void launch_model_t(mc3hip_ctx *c)
{
    const int nb = (c->P.ndata + 255) / 256;
    MC3_LAUNCH((eval_model_kernel<MT>), dim3(nb > 1024 ? 1024 : nb), dim3(256), 0,
                       c->stream, c->P, (const double *)c->eval_params, c->model_out);
}

// The general chain kernels are instantiated per likelihood and workgroup width in
// translation units of their own (ctx.h); a context's c->wpc is 1, 4 or 16 with the Gaussian
// likelihood, 1, 4 or 8 with the wavelet one (choose_wpc), whose large-data instances (more
// than 2^14 points, P.wl_scratch) always have eight waves.
#define MC3_BY_WIDTH(call_gauss, call_wavelet)                                      \
    if (c->P.wlike) {                                                               \
        if (c->P.wl_scratch) return call_wavelet(large);                            \
        switch (c->wpc) {                                                           \
        case 8: return call_wavelet(w8);                                            \
        case 4: return call_wavelet(w4);                                            \
        default: return call_wavelet(w1);                                           \
        }                                                                           \
    }                                                                               \
    switch (c->wpc) {                                                               \
    case 4: return call_gauss(w4);                                                  \
    default: return call_gauss(w1);                                                 \
    }

void launch_step_fused(mc3hip_ctx *c, int roff)
{
#define G(tag) launch_step_gauss_##tag(c, roff)
#define W(tag) launch_step_wavelet_##tag(c, roff)
    MC3_BY_WIDTH(G, W)
#undef G
#undef W
}

// the exact instance (likelihood, tile count, peer-to-peer stores) of the general persistent
// kernel that will be launched
chain_kernel_fn persist_instance(const mc3hip_ctx *c, bool p2p)
{
#define G(tag) persist_instance_gauss_##tag(c, p2p)
#define W(tag) persist_instance_wavelet_##tag(c, p2p)
    MC3_BY_WIDTH(G, W)
#undef G
#undef W
}

void launch_eval_one(mc3hip_ctx *c, int n)
{
#define G(tag) launch_eval_gauss_##tag(c, n)
#define W(tag) launch_eval_wavelet_##tag(c, n)
    MC3_BY_WIDTH(G, W)
#undef G
#undef W
}

hipError_t set_smem_attr_wavelet(mc3hip_ctx *c)
{
#define G(tag) hipSuccess
#define W(tag) set_smem_attr_wavelet_##tag(c)
    MC3_BY_WIDTH(G, W)
#undef G
#undef W
}

// threads and dynamic LDS of the general persistent instance that runs
int persist_threads(const mc3hip_ctx *c, bool p2p)
{
    // (the one-wave wavelet chains written out by hand: a helper wave beside the chain's)
    // (four waves per chain -- 4 096 points --: the chain is wave 0's)
    if (c->wl_one && c->mc_chains == 0 && !p2p)
        return c->wpc == 1 ? 128 : 64 * c->wpc;
    return chainwave_runs(c, p2p) ? chainwave_threads(c, p2p) : 64 * c->wpc;
}
size_t persist_smem(const mc3hip_ctx *c, bool p2p)
{
    return chainwave_runs(c, p2p) ? c->smem + CW_EXTRA_DOUBLES * sizeof(double) : c->smem;
}

void launch_persist(mc3hip_ctx *c, int nref)
{
    const bool mc = c->mc_chains > 0, p2p = c->p2p_ranks > 1;
    MC3_LAUNCH_CHAIN(c, persist_instance(c, p2p), dim3(mc ? c->mc_grid : c->P.nlocal),
                       dim3(persist_threads(c, p2p)), mc ? c->mc_smem : persist_smem(c, p2p),
                       c->stream, c->P, nref);
}

// workgroups of the general persistent kernel that fit on one CU (0 on error)
void persist_blocks_per_cu(mc3hip_ctx *c, int *out)
{
    const bool p2p = c->p2p_ranks > 1;
    const void *f = reinterpret_cast<const void *>(persist_instance(c, p2p));
    if (f && persist_smem(c, p2p) > 64 * 1024)
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)persist_smem(c, p2p));
    int nb = blocks_per_cu(f, persist_threads(c, p2p), persist_smem(c, p2p));
    if ((c->flags & MC3HIP_FLAG_P2P) && !p2p) {
        const void *g = reinterpret_cast<const void *>(persist_instance(c, true));
        if (g && c->smem > 64 * 1024)
            (void)hipFuncSetAttribute(g, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)c->smem);
        const int np = blocks_per_cu(g, 64 * c->wpc, c->smem);
        nb = np < nb ? np : nb;
    }
    *out = nb;
}

void launch_step(mc3hip_ctx *c, int roff)
{
    if (c->P.tile > 0)
        launch_refresh_tiled(c, roff);
    else
        launch_step_fused(c, roff);
}

// The persistent kernel spins on rows other workgroups publish, so EVERY workgroup
// of its grid must be resident at once.  Residency comes from the grid size alone:
// use it only when nlocal fits the occupancy the runtime reports.
constexpr int MC_MAX_TURNS = 64;

int decide_persist(mc3hip_ctx *c)
{
    if (c->persist_checked)
        return c->persist_refs;
    c->persist_checked = true;
    c->persist_refs = 0;
    c->mc_chains = 0;
    c->P.mc_chains = 0;
    c->P.cw_lean = 0;
    c->P.rs_lean = 0;
    if ((c->flags & MC3HIP_FLAG_NO_PERSIST) || env_text("MC3HIP_NO_PERSIST"))
        return 0;
    c->resident = false;
    if (c->P.tile > 0) {
        // tiled plan: one launch only where the whole grid of the resident kernel -- tiles x
        // columns of chains -- is on the chip at once and every chain finds an owner wave
        if (!resident_shape(c))
            return 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) != hipSuccess)
            return 0;
        // (a context made for the peer-to-peer hand-off runs the plain instance until the
        // peers are attached: both must fit)
        int per_cu = 1 << 30;
        for (int p2p = 0; p2p < 2; ++p2p) {
            if (p2p != (c->p2p_ranks > 1 ? 1 : 0) && !(c->flags & MC3HIP_FLAG_P2P))
                continue;
            const void *f = reinterpret_cast<const void *>(resident_kernel(c, p2p != 0));
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)resident_lds_bytes(c));
            const int nb = blocks_per_cu(f, 512, resident_lds_bytes(c));
            per_cu = nb < per_cu ? nb : per_cu;
        }
        const long long capacity = (long long)per_cu * prop.multiProcessorCount;
        long long maxcol = capacity / c->P.ntiles;
        // (ranks that share one device -- the multi-process tests on a one-GPU box -- must
        // share its CUs too: MC3HIP_RESIDENT_MAXCOL caps the columns of chains per rank)
        if (const char *e = env_text("MC3HIP_RESIDENT_MAXCOL"))
            if (atoi(e) > 0 && atoi(e) < maxcol)
                maxcol = atoi(e);
        if (per_cu < 1 || maxcol < 1)
            return 0;
        // the fewest columns that fit, each a multiple of the 16-chain block
        long long cb = ((long long)c->P.nlocal + maxcol - 1) / maxcol;
        cb = (cb + 15) / 16 * 16;
        const long long ncol = ((long long)c->P.nlocal + cb - 1) / cb;
        if (ncol * c->P.ntiles > capacity || (long long)c->P.nlocal > 8 * ncol * c->P.ntiles)
            return 0;
        c->resident = true;
        c->capacity = capacity;
        c->resident_chain_block = (int)cb;
        c->P.rs_lean = resident_owner_lean(c) ? 1 : 0;
        // the owners as waves of their own, the blocks of chains made up at run time
        // (resident.h resident_async_kernel): one workgroup per CU, a column of at most
        // RA_CMAX chains, at most RA_OWNERS chains to own per workgroup
        c->resident_owner_waves = 0;
        const long long nwg = ncol * c->P.ntiles;
        const long long owners = ((long long)c->P.nlocal + nwg - 1) / nwg;
        // ... where it pays: an iteration in step costs ~6 us of hand-offs + 3.4 us per block
        // of sixteen chains of a column; with nobody in step the hand-offs overlap with the
        // blocks, but a chain's own round -- sum, decision, row, the partners' proposals, the
        // scheduler, the rows of the block -- is ~8 round trips of 1.2 us, and below ~20 us of
        // matrix work per iteration THAT is what an iteration takes (tools/
        // sweep_resident_async.py: 256 chains x 16 384 points x 10 columns 8.2 us in step, 11.3
        // not; 512 x 100 000 29.9 / 29.3, with snooker 31.9 / 30.1).  DEMC, whose proposals wait
        // for two partners' decisions, from six blocks per column; the samplers that wait for
        // nobody from four.  (MC3HIP_RS_ASYNC=1: whatever the size -- the tests.)
        const bool async_pays = env_text("MC3HIP_RS_ASYNC") != nullptr ||
                                cb / 16 >= (resident_owner_lean(c) ? 6 : 4);
        if (resident_async_shape(c) && async_pays && cb <= RA_CMAX && owners <= RA_OWNERS &&
            nwg <= prop.multiProcessorCount) {
            bool fits = true;
            for (int p2p = 0; p2p < 2; ++p2p) {
                if (p2p != (c->p2p_ranks > 1 ? 1 : 0) && !(c->flags & MC3HIP_FLAG_P2P))
                    continue;
                const void *f = reinterpret_cast<const void *>(
                    resident_async_kernel_of(c, p2p != 0));
                (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)resident_async_lds_bytes(c));
                fits = fits && blocks_per_cu(f, 512 + 64 * ((int)owners + 1),
                                             resident_async_lds_bytes(c)) >= 1;
            }
            // the scheduler's blocks: a row of sixteen words per block of a launch's iterations
            const long long refs = env_text("MC3HIP_PERSIST_REFS") &&
                                   atoi(env_text("MC3HIP_PERSIST_REFS")) > 0
                                       ? atoi(env_text("MC3HIP_PERSIST_REFS")) : 16384;
            // (a row per block of every round of proposals of a launch: `thinning` rounds to a
            // refresh)
            const long long rows = refs * c->P.thinning * ((cb + 15) / 16);
            const size_t words = (size_t)ncol * rows * 16;
            // (a large thinning makes a launch's rounds many: not beyond 1 GiB of rows)
            fits = fits && words * sizeof(unsigned long long) <= ((size_t)1 << 30);
            if (fits && words > c->rs_sched_words) {
                c->rs_sched_words = words;
                fits = dev_alloc(c, &c->P.rs_sched, words) == MC3HIP_OK &&
                       hipMemset(c->P.rs_sched, 0, words * sizeof(unsigned long long)) ==
                           hipSuccess;
                c->P.rs_gen = 0;
            }
            if (fits) {
                c->resident_owner_waves = (int)owners;
                c->P.rs_sched_rows = (int)rows;
            }
        }
        c->fast = false;
        c->persist_refs = 16384;
        if (const char *e = env_text("MC3HIP_PERSIST_REFS"))
            c->persist_refs = atoi(e) > 0 ? atoi(e) : 0;
        return c->persist_refs;
    }
    int nb = 0;
    c->fast = fast_eligible(c) || jit_serves(c);
    // (should a shape fast_eligible lets through have no lean instance after all: the general
    // chain -- never a launch of nothing)
    if (c->fast && !jit_serves(c)) {
        const bool p2p_now = c->p2p_ranks > 1;
        if (!fast_instance(c, p2p_now) ||
            ((c->flags & MC3HIP_FLAG_P2P) && !p2p_now && !fast_instance(c, true)))
            c->fast = false;
    }
    // wavelet likelihood with eight waves per chain: a ninth wave runs the chain
    c->chainwave = c->P.wlike && c->wpc == 8 && !c->P.wl_scratch &&
                   !(c->flags & MC3HIP_FLAG_P2P) &&
                   c->smem + CW_EXTRA_DOUBLES * sizeof(double) <= 160 * 1024;
    c->wl_one = !c->fast && c->P.wlike && (c->wpc == 1 || c->wpc == 4) && !c->P.wl_scratch &&
                cw_lean_eligible(c) &&
                !(c->flags & MC3HIP_FLAG_P2P) && c->p2p_ranks <= 1 && !env_text("MC3HIP_NO_WL_ONE") &&
                !env_text("MC3HIP_NO_CW_LEAN");
    // (bit 1: an out-of-bounds proposal rides with the turn behind it, kernels_wavelet.h
    // pipe_chain_wave_one; MC3HIP_NO_CW_RIDE gives it a slot of its own as before)
    c->P.cw_lean = (c->chainwave && cw_lean_eligible(c) && !env_text("MC3HIP_NO_CW_LEAN"))
                       ? (dev_env_text("MC3HIP_NO_CW_RIDE") ? 1 : 3)
                       : 0;
    if (c->fast)
        nb = fast_blocks_per_cu(c);
    else
        persist_blocks_per_cu(c, &nb);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess)
        return 0;
    // The full figure of the occupancy query is used: measured on MI355X, 2048
    // single-wave workgroups (8 per CU, the reported maximum) are all resident and run
    // 3.7x faster than one launch per refresh.  Should the query ever over-report
    // (MI355X_MICROARCH.md "Residency and cooperative launch"), the launch gives up after
    // a bounded wait and the run is redone with per-refresh launches (settle()).
    // MC3HIP_PERSIST_MARGIN=1 keeps one workgroup per CU of slack instead.
    const int per_cu = (nb > 1 && dev_env_text("MC3HIP_PERSIST_MARGIN")) ? nb - 1 : nb;
    long long capacity = (long long)per_cu * prop.multiProcessorCount;
    // (tests: pretend that fewer workgroups fit, so that small ensembles take the
    // several-chains-per-workgroup path)
    long long cap_limit = 1LL << 40;
    if (const char *e = env_text("MC3HIP_PERSIST_CAPACITY"))
        if (atoll(e) > 0)
            cap_limit = atoll(e);
    capacity = capacity < cap_limit ? capacity : cap_limit;
    if (nb < 1)
        return 0;
    if (c->fast && c->wide_tiles > 0 && c->P.nlocal > capacity) {
        // the wide loop (tiles in the registers of four / eight waves per chain) holds one chain
        // per workgroup: an ensemble that is not co-resident that way runs what it ran before
        c->wide_tiles = 0;
        c->fast = fast_eligible(c) || jit_serves(c);
        if (c->fast)
            nb = fast_blocks_per_cu(c);
        else
            persist_blocks_per_cu(c, &nb);
        const int pc = (nb > 1 && dev_env_text("MC3HIP_PERSIST_MARGIN")) ? nb - 1 : nb;
        capacity = (long long)pc * prop.multiProcessorCount;
        capacity = capacity < cap_limit ? capacity : cap_limit;
        if (nb < 1)
            return 0;
    }
    if (c->fast && jit_serves(c) && fast_eligible(c) && c->P.nlocal > capacity) {
        // a loop specialised at run time holds one chain per workgroup; where the library has an
        // instance of its own for the shape (the random walk beyond one tile: streamed tiles),
        // an ensemble that is not co-resident with the specialised loop runs that one, several
        // chains per workgroup
        c->jit_fn = nullptr;
        nb = fast_blocks_per_cu(c);
        const int pc = (nb > 1 && dev_env_text("MC3HIP_PERSIST_MARGIN")) ? nb - 1 : nb;
        capacity = (long long)pc * prop.multiProcessorCount;
        capacity = capacity < cap_limit ? capacity : cap_limit;
        if (nb < 1)
            return 0;
    }
    if (c->fast && c->P.nlocal > capacity && lean_one_chain_only(c)) {
        // fixed coefficients / priors with thinning > 1 or peer to peer, the support
        // distribution: only a one-chain-per-wave loop is lean there; more chains than it can
        // hold run the general chain
        c->fast = false;
        persist_blocks_per_cu(c, &nb);
        const int pc = (nb > 1 && dev_env_text("MC3HIP_PERSIST_MARGIN")) ? nb - 1 : nb;
        capacity = (long long)pc * prop.multiProcessorCount;
        capacity = capacity < cap_limit ? capacity : cap_limit;
        if (nb < 1)
            return 0;
    }
    // The wavelet kernel pipelined over the turns of a workgroup (kernels_wavelet.h
    // wavelet_pipe_kernel) also serves ONE chain per workgroup: the next proposal is issued as if
    // the one in flight were rejected (most are), and issued again if it was not.
    const bool spec_pipe = !c->fast && c->chainwave && c->P.cw_lean != 0 &&
                           c->P.sampler != MC3HIP_DEMC && c->P.nlocal <= capacity &&
                           !(c->flags & MC3HIP_FLAG_P2P) && c->p2p_ranks <= 1 &&
                           !env_text("MC3HIP_NO_CW_SPEC");
    if (c->P.nlocal > capacity || spec_pipe) {
        // More chains than workgroups can be co-resident: a workgroup advances several per
        // refresh (run_chain MC), as the reference's workers do with their ncpp chains
        // (mc3/chain.py:163,190-191).  The fewest turns whose grid fits: the waiting state of
        // a chain costs zs + 4 doubles of LDS, which can lower the occupancy, so the query is
        // repeated with the instance and the LDS size that would run.  One GPU only.
        // One GPU -- or, sharded over several, the loops with the chains of a wave side by side
        // (kernels_rows.h: up to four chains per wave, peer to peer as well).
        const bool p2p_ctx = (c->flags & MC3HIP_FLAG_P2P) || c->p2p_ranks > 1;
        if (!spec_pipe &&
            (env_text("MC3HIP_NO_MULTICHAIN") || (p2p_ctx && !(c->fast && lean_rows_shape(c)))))
            return 0;
        // (the nine-wave wavelet kernel takes turns only with its hand-structured chain wave;
        // should its LDS not leave room for the waiting states, the fused kernel is tried)
        bool found = false;
        for (int attempt = 0; attempt < (spec_pipe ? 1 : 2) && !found; ++attempt) {
            const bool cw = !c->fast && c->chainwave && c->P.cw_lean != 0;
            if (attempt == 1) {
                if (!cw)
                    break;
                c->P.cw_lean = 0;
            }
            const bool cw_now = attempt == 0 && cw;
            const size_t base = c->smem + (cw_now ? CW_EXTRA_DOUBLES * sizeof(double) : 0);
            const int cw_wpc = 64 * c->wpc;
            for (int turns = spec_pipe ? 1 : 2; turns <= (p2p_ctx ? 4 : (spec_pipe ? 1 : MC_MAX_TURNS)) && !found;
                 ++turns) {
                c->mc_chains = turns;                       // (selects the MC instance)
                const size_t smem = base + (size_t)turns * (c->P.zs + 4) * sizeof(double);
                if (smem > 160 * 1024)
                    break;
                const void *f = reinterpret_cast<const void *>(
                    c->fast ? fast_instance(c, false) : persist_instance(c, false));
                if (f && smem > 64 * 1024)
                    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)smem);
                const int threads = cw_now ? chainwave_threads(c, false) : cw_wpc;
                int nbm = blocks_per_cu(f, threads, smem);
                if (p2p_ctx) {
                    // (a context made for the peer-to-peer hand-off runs the plain instance
                    // until the peers are attached: both must fit)
                    const int nbp = blocks_per_cu(
                        reinterpret_cast<const void *>(fast_instance(c, true)), threads, smem);
                    nbm = nbp < nbm ? nbp : nbm;
                }
                if (nbm > 1 && dev_env_text("MC3HIP_PERSIST_MARGIN"))
                    nbm -= 1;
                const long long grid = ((long long)c->P.nlocal + turns - 1) / turns;
                long long room = (long long)nbm * prop.multiProcessorCount;
                room = room < cap_limit ? room : cap_limit;
                if (nbm >= 1 && grid <= room) {
                    capacity = room;                        // (of the instance that runs)
                    c->mc_grid = (int)grid;
                    c->mc_smem = smem;
                    c->P.mc_chains = turns;
                    c->P.mc_off = (int32_t)(base / sizeof(double));
                    found = true;
                }
            }
        }
        if (!found) {
            c->mc_chains = 0;
            if (!spec_pipe)
                return 0;                                   // (else: the nine-wave kernel, as ever)
        }
    }
    c->capacity = capacity;
    c->persist_refs = 16384;
    if (const char *e = env_text("MC3HIP_PERSIST_REFS"))
        c->persist_refs = atoi(e) > 0 ? atoi(e) : 0;
    return c->persist_refs;
}
void launch_eval(mc3hip_ctx *c, int n)
{
    if (c->P.wl_scratch && n > WL_EVAL_BLOCKS) {
        // large wavelet data sets: every workgroup needs its scratch, so the vectors go in
        // rounds (the kernel reads its inputs / writes its outputs at blockIdx offsets)
        double *const p0 = c->eval_params, *const o0 = c->eval_out;
        for (int at = 0; at < n; at += WL_EVAL_BLOCKS) {
            c->eval_params = p0 + (size_t)at * c->P.nparams;
            c->eval_out = o0 + at;
            const int m = n - at < WL_EVAL_BLOCKS ? n - at : WL_EVAL_BLOCKS;
            launch_eval_one(c, m);
        }
        c->eval_params = p0;
        c->eval_out = o0;
        return;
    }
    launch_eval_one(c, n);
}

void launch_model(mc3hip_ctx *c)
{
    switch (c->mt) {
    case MT_POLY2: launch_model_t<MT_POLY2>(c); break;
    case MT_POLY3: launch_model_t<MT_POLY3>(c); break;
    case MT_POLY4: launch_model_t<MT_POLY4>(c); break;
    case MT_LINEAR: launch_model_t<MT_LINEAR>(c); break;
    default: launch_model_t<MT_POLY_GENERIC>(c); break;
    }
}


int set_smem_attr(mc3hip_ctx *c)
{
    // kernels that need more than the default 64 KiB of dynamic LDS must opt in (only the
    // wavelet kernels do: the data or the abscissae of a chain live in LDS)
    if (c->smem <= 64 * 1024 || !c->P.wlike)
        return MC3HIP_OK;
    HIP_TRY(set_smem_attr_wavelet(c));
    return MC3HIP_OK;
}

// ---- dense host rows <-> padded device rows (core.h ZROW) ----------------------
constexpr size_t STAGE_MAX_DOUBLES = size_t(8) << 20;      // 64 MiB of staging at most
#define RC_TRY(expr)                                                                \
    do {                                                                            \
        const int rc_try_ = (expr);                                                 \
        if (rc_try_)                                                                \
            return rc_try_;                                                         \
    } while (0)

static int stage_reserve(mc3hip_ctx *c, size_t doubles)
{
    if (doubles <= c->stage_doubles)
        return MC3HIP_OK;
    if (c->stage)
        HIP_TRY(hipFree(c->stage));
    c->stage = nullptr;
    c->stage_doubles = 0;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, doubles * sizeof(double)));
    c->stage = static_cast<double *>(p);
    c->stage_doubles = doubles;
    return MC3HIP_OK;
}

static int rows_to_device(mc3hip_ctx *c, double *padded, const double *host, size_t nrows)
{
    const size_t nf = (size_t)c->P.nfree, zs = (size_t)c->P.zs;
    const size_t chunk = STAGE_MAX_DOUBLES / nf ? STAGE_MAX_DOUBLES / nf : 1;
    for (size_t r = 0; r < nrows; r += chunk) {
        const size_t m = nrows - r < chunk ? nrows - r : chunk;
        RC_TRY(stage_reserve(c, m * nf));
        HIP_TRY(hipMemcpyAsync(c->stage, host + r * nf, m * nf * sizeof(double),
                               hipMemcpyHostToDevice, c->stream));
        const int blocks = (int)((m * nf + 255) / 256 < 4096 ? (m * nf + 255) / 256 : 4096);
        MC3_LAUNCH(unpack_rows_kernel, dim3(blocks), dim3(256), 0, c->stream,
                           padded + r * zs, (const double *)c->stage, (long long)m, (int)nf,
                           (int)zs);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return MC3HIP_OK;
}

static int rows_to_host(mc3hip_ctx *c, double *host, const double *padded, size_t nrows)
{
    const size_t nf = (size_t)c->P.nfree, zs = (size_t)c->P.zs;
    const size_t chunk = STAGE_MAX_DOUBLES / nf ? STAGE_MAX_DOUBLES / nf : 1;
    for (size_t r = 0; r < nrows; r += chunk) {
        const size_t m = nrows - r < chunk ? nrows - r : chunk;
        RC_TRY(stage_reserve(c, m * nf));
        const int blocks = (int)((m * nf + 255) / 256 < 4096 ? (m * nf + 255) / 256 : 4096);
        MC3_LAUNCH(pack_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, c->stage,
                           padded + r * zs, (long long)m, (int)nf, (int)zs);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host + r * nf, c->stage, m * nf * sizeof(double),
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return MC3HIP_OK;
}

int exchange_rows(mc3hip_ctx *c, int64_t R)
{
    // one in-place all-gather of the refresh's Z rows: rank r's chains are the
    // contiguous block [chain0, chain0+nlocal) of the nchains rows of refresh R
    // (whole padded rows: zs doubles each)
    const size_t zs = (size_t)c->P.zs;
    double *base = c->P.Z + (size_t)(c->P.M0 + R * c->P.nchains) * zs;
    const double *send = base + (size_t)c->P.chain0 * zs;
    int rc = g_rccl.AllGather(send, base, (size_t)c->P.nlocal * zs, NCCL_FLOAT64, c->comm,
                              c->stream);
    if (rc != 0)
        return fail(MC3HIP_ERR_COMM, "ncclAllGather failed: %s",
                    g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return MC3HIP_OK;
}

}  // namespace

// (the library is compiled with -fvisibility=hidden: only the C-ABI is exported)
#pragma GCC visibility push(default)
extern "C" {

const char *mc3hip_last_error(void) { return g_err.c_str(); }

int mc3hip_create(const mc3hip_problem *pr, mc3hip_ctx **out)
{
    if (!pr || !out)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (pr->struct_bytes != (int32_t)sizeof(mc3hip_problem))
        return fail(MC3HIP_ERR_ARG, "mc3hip_problem size mismatch (%d vs %zu)",
                    pr->struct_bytes, sizeof(mc3hip_problem));
    if (pr->nparams < 1 || pr->nparams > MC3HIP_MAX_PARAMS)
        return fail(MC3HIP_ERR_ARG, "nparams=%d outside [1,%d]", pr->nparams,
                    MC3HIP_MAX_PARAMS);
    if (pr->ndata < 1)
        return fail(MC3HIP_ERR_ARG, "ndata must be positive");
    if (pr->nchains < 1 || pr->nlocal < 1 || pr->chain0 < 0 ||
        pr->chain0 + pr->nlocal > pr->nchains)
        return fail(MC3HIP_ERR_ARG, "bad chain partition (nchains=%d chain0=%d nlocal=%d)",
                    pr->nchains, pr->chain0, pr->nlocal);
    if (pr->sampler < MC3HIP_MRW || pr->sampler > MC3HIP_SNOOKER)
        return fail(MC3HIP_ERR_ARG, "unknown sampler %d", pr->sampler);
    if (pr->sampler == MC3HIP_DEMC && pr->nchains < 3)
        return fail(MC3HIP_ERR_ARG, "demc needs at least 3 chains (chain.py:223-229)");
    if (pr->thinning < 1)
        return fail(MC3HIP_ERR_ARG, "thinning must be >= 1");
    if (pr->zlag < 0 || pr->zlag > 64)
        return fail(MC3HIP_ERR_ARG, "zlag=%d outside [0,64]", pr->zlag);
    if (pr->model_kind != MC3HIP_MODEL_POLY && pr->model_kind != MC3HIP_MODEL_LINEAR)
        return fail(MC3HIP_ERR_ARG, "unknown model family %d", pr->model_kind);
    const int noise = pr->wlike ? 3 : 0;
    if (pr->nmodel < 1 || pr->nmodel + noise > pr->nparams)
        return fail(MC3HIP_ERR_ARG, "nmodel=%d does not fit nparams=%d%s", pr->nmodel,
                    pr->nparams, pr->wlike ? " (wavelet: last three are noise parameters)" : "");
    if (pr->wlike) {
        // src_c/_dwt.c:94 transforms the unpadded length: only 2^M is well defined
        if (pr->ndata < 2 || (pr->ndata & (pr->ndata - 1)))
            return fail(MC3HIP_ERR_ARG,
                        "wavelet likelihood needs power-of-two ndata (got %d)", pr->ndata);
        if (pr->ndata > (1 << 22))
            return fail(MC3HIP_ERR_ARG, "wavelet likelihood supports ndata <= 2^22");
        if (pr->wl_nreal < 0 || pr->wl_nreal > pr->ndata)
            return fail(MC3HIP_ERR_ARG, "wl_nreal=%d outside [0,%d]", pr->wl_nreal, pr->ndata);
    }
    if (pr->M0 < 2 || pr->zlen < pr->M0)
        return fail(MC3HIP_ERR_ARG, "bad Z sizes (M0=%lld zlen=%lld)", (long long)pr->M0,
                    (long long)pr->zlen);
    if (pr->M0 < pr->nchains)
        return fail(MC3HIP_ERR_ARG, "M0 must be >= nchains (hsize >= 1)");
    if (!pr->params || !pr->pstep || !pr->pmin || !pr->pmax || !pr->prior || !pr->priorlow ||
        !pr->priorup || !pr->indep || !pr->data || (!pr->wlike && !pr->uncert))
        return fail(MC3HIP_ERR_ARG, "null input array");

    // parameter bookkeeping: chain.py:141-150
    std::vector<int32_t> ifree, ishare, share_src, igprior;
    for (int j = 0; j < pr->nparams; ++j) {
        if (pr->pstep[j] > 0)
            ifree.push_back(j);
        else if (pr->pstep[j] < 0) {
            const int src = -(int)pr->pstep[j] - 1;                  // chain.py:247
            if (src < 0 || src >= pr->nparams)
                return fail(MC3HIP_ERR_ARG, "shared parameter %d points outside params", j);
            ishare.push_back(j);
            share_src.push_back(src);
        }
        if (pr->priorlow[j] > 0 && pr->priorup[j] > 0)
            igprior.push_back(j);                                    // stats.py:211
    }
    if (ifree.empty() || (int)ifree.size() > MC3HIP_MAX_FREE)
        return fail(MC3HIP_ERR_ARG, "nfree=%zu outside [1,%d]", ifree.size(), MC3HIP_MAX_FREE);

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (pr->device < 0 || pr->device >= ndev)
        return fail(MC3HIP_ERR_HIP, "device %d not present (%d devices)", pr->device, ndev);
    HIP_TRY(hipSetDevice(pr->device));

    mc3hip_ctx *c = new mc3hip_ctx();
    c->device = pr->device;
    c->flags = pr->flags;
    c->zlen = pr->zlen;
    c->model_kind = pr->model_kind;
    int rc = MC3HIP_OK;
#define TRY(expr)                                                                   \
    do {                                                                            \
        rc = (expr);                                                                \
        if (rc) {                                                                   \
            mc3hip_destroy(c);                                                      \
            return rc;                                                              \
        }                                                                           \
    } while (0)
#define HIP_TRY_C(expr)                                                             \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            mc3hip_destroy(c);                                                      \
            return fail(MC3HIP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                                           \
    } while (0)
    HIP_TRY_C(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY_C(hipEventCreate(&c->ev0));
    HIP_TRY_C(hipEventCreate(&c->ev1));

    DevProblem &P = c->P;
    P.nchains = pr->nchains;
    P.chain0 = pr->chain0;
    P.nlocal = pr->nlocal;
    P.nparams = pr->nparams;
    P.nfree = (int)ifree.size();
    P.nshare = (int)ishare.size();
    P.ndata = pr->ndata;
    P.sampler = pr->sampler;
    P.wlike = pr->wlike ? 1 : 0;
    P.nmodel = pr->nmodel;
    P.thinning = pr->thinning;
    P.ngprior = (int)igprior.size();
    P.trace_iters = pr->trace_iters > 0 ? pr->trace_iters : 0;
    P.M0 = pr->M0;
    P.seed = pr->seed;
    P.gamma = pr->gamma;
    P.fepsilon = pr->fepsilon;
    const size_t np = (size_t)pr->nparams, nd = (size_t)pr->ndata, nf = ifree.size();
    TRY(dev_upload(c, &P.ifree, ifree.data(), ifree.size()));
    TRY(dev_upload(c, &P.ishare, ishare.data(), ishare.size()));
    TRY(dev_upload(c, &P.share_src, share_src.data(), share_src.size()));
    TRY(dev_upload(c, &P.igprior, igprior.data(), igprior.size()));
    TRY(dev_upload(c, &P.params, pr->params, np));
    TRY(dev_upload(c, &P.pmin, pr->pmin, np));
    TRY(dev_upload(c, &P.pmax, pr->pmax, np));
    TRY(dev_upload(c, &P.pstep, pr->pstep, np));
    TRY(dev_upload(c, &P.prior, pr->prior, np));
    TRY(dev_upload(c, &P.priorlow, pr->priorlow, np));
    TRY(dev_upload(c, &P.priorup, pr->priorup, np));
    {
        std::vector<double> fmin(nf), fmax(nf), fstep(nf);
        for (size_t k = 0; k < nf; ++k) {
            fmin[k] = pr->pmin[ifree[k]];
            fmax[k] = pr->pmax[ifree[k]];
            fstep[k] = pr->pstep[ifree[k]];
        }
        TRY(dev_upload(c, &P.fmin, fmin.data(), nf));
        TRY(dev_upload(c, &P.fmax, fmax.data(), nf));
        TRY(dev_upload(c, &P.fstep, fstep.data(), nf));
    }
    {
        // Device copies padded to a multiple of 1024 points: x / A columns and data
        // repeat the last point, w = 1/uncert (IEEE division on the host, so the hot
        // loop multiplies) is 0 in the padding -- a padded point contributes exactly 0.
        // tiled plan (tiled.h tile_chisq_kernel): Gaussian chisq over at least
        // four 4096-point tiles, enough chains to amortise a tile, <= 16 coefficients.
        // Models the one-launch resident kernels take (design matrices of 5..11 columns,
        // polynomials of 2..4 coefficients) also from 8192 points on when the ensemble is
        // large: measured on MI355X (tools/mid_range.py) the resident plan costs ~6.5 us +
        // 0.26 ps per chain-point and iteration, the fused kernels ~1.2 us + 1.4 ps and
        // more once several workgroups share a CU (8192 x 768: 15.8 us fused, 7.4 tiled;
        // 16 000 x 2048: 54.7 / 12.0; 16 384 x 256: 7.7 / 7.3; 8192 x 256: 5.0 fused),
        // crossing near 4 M chain-points.
        const bool linear_model = pr->model_kind == MC3HIP_MODEL_LINEAR;
        // (design matrices of 2..15 columns go through the matrix cores, 2048-point tiles)
        const bool mfma_model = linear_model && pr->nmodel >= 2 && pr->nmodel <= 15 &&
                                true;
        const bool resident_model = linear_model ? mfma_model
                                                 : (pr->nmodel >= 2 && pr->nmodel <= 4);
        // (the mid-range rule was measured with the ONE-launch plan; where that plan cannot
        // run -- switched off, or a grid of tiles x columns of chains that is not co-resident
        // with an owner wave for every chain, the test decide_persist applies later -- the
        // fused kernels keep the shapes below 16 384 points.  An RCCL communicator attached
        // afterwards also forces three launches per iteration: not knowable here.)
        bool mid_resident = resident_model && !(pr->flags & MC3HIP_FLAG_NO_PERSIST) &&
                            !env_text("MC3HIP_NO_PERSIST") && !env_text("MC3HIP_NO_RESIDENT");
        if (mid_resident) {
            hipDeviceProp_t prop;
            const long long tile_pts = (pr->nmodel <= 4 && !mfma_model) ? 4096 : 2048;
            const long long nt = ((long long)nd + tile_pts - 1) / tile_pts;
            const long long cus = hipGetDeviceProperties(&prop, c->device) == hipSuccess
                                      ? prop.multiProcessorCount : 256;
            const long long maxcol = cus / nt;
            mid_resident = maxcol >= 1;
            if (mid_resident) {
                long long cb = ((long long)pr->nlocal + maxcol - 1) / maxcol;
                cb = (cb + 15) / 16 * 16;
                const long long ncol = ((long long)pr->nlocal + cb - 1) / cb;
                mid_resident = ncol * nt <= cus && (long long)pr->nlocal <= 8 * ncol * nt;
            }
        }
        // 2 049 ... 16 384 points of the plain strict DEMC shape with few chains: the tiles in
        // the registers of four (eight) waves per chain, kernels_lean.h lean_demc_wide_kernel --
        // 1.5-2 us per iteration where the streamed one-wave loop takes 2.5-3, the general chain
        // 4-7 and the one-launch tiled plan its flat ~6.7 (tools/mid_range.py).  MC3HIP_NO_WIDE.
        {
            const long long tiles = ((long long)nd + VLANES - 1) / VLANES;
            hipDeviceProp_t prop;
            const long long cus = hipGetDeviceProperties(&prop, c->device) == hipSuccess
                                      ? prop.multiProcessorCount : 256;
            const bool shape = !pr->wlike && !linear_model && pr->nmodel >= 2 && pr->nmodel <= 4 &&
                               pr->nmodel == pr->nparams && P.nfree == pr->nparams &&
                               P.ngprior == 0 && P.nshare == 0 &&
                               (pr->sampler == MC3HIP_DEMC || pr->sampler == MC3HIP_SNOOKER) &&
                               pr->thinning == 1 && pr->zlag == 0 && pr->fepsilon == 0.0 &&
                               pr->trace_iters == 0 && pr->nlocal == pr->nchains &&
                               !(pr->flags & (MC3HIP_FLAG_P2P | MC3HIP_FLAG_NO_PERSIST)) &&
                               !env_text("MC3HIP_NO_WIDE") && !env_text("MC3HIP_NO_LEAN") &&
                               !dev_env_text("MC3HIP_NO_FAST") && !env_text("MC3HIP_NO_PERSIST") &&
                               !env_text("MC3HIP_WPC");
            c->wide_tiles = 0;
            // (co-residency is decided by the occupancy query of the very instance --
            // decide_persist, which takes the choice back if the grid does not fit: two tiles
            // three workgroups of four waves per CU, four tiles two, eight tiles one)
            if (shape && tiles >= 2 && tiles <= 8 &&
                pr->nlocal <= (tiles <= 2 ? 3 : (tiles <= 4 ? 2 : 1)) * cus)
                c->wide_tiles = tiles <= 2 ? 2 : (tiles <= 4 ? 4 : 8);
            else if (shape && tiles > 8 && tiles <= 16 && pr->nlocal <= cus)
                c->wide_tiles = 16;
            // the same sizes (up to eight tiles) with fixed / shared coefficients, priors, the
            // support distribution, or the random walk: lean_*_body<.., NW = 4, NT> specialised
            // at run time (ctx.h jit_model_tag; without a compiler at hand: the general chain,
            // as before)
            const bool model_jit = linear_model ? (pr->nmodel >= 1 && pr->nmodel <= 16)
                                                : (pr->nmodel >= 2 && pr->nmodel <= 16);
            const bool shape_jit = !pr->wlike && model_jit && pr->nmodel == pr->nparams &&
                                   P.nfree >= 1 && pr->zlag == 0 && pr->trace_iters == 0 &&
                                   (pr->thinning == 1 || pr->sampler == MC3HIP_MRW ||
                                    ((pr->sampler == MC3HIP_SNOOKER ||
                                      pr->sampler == MC3HIP_DEMC) && pr->fepsilon == 0.0)) &&
                                   pr->nlocal == pr->nchains &&
                                   !(pr->flags & (MC3HIP_FLAG_P2P | MC3HIP_FLAG_NO_PERSIST)) &&
                                   !env_text("MC3HIP_NO_WIDE") && !env_text("MC3HIP_NO_LEAN") &&
                                   !dev_env_text("MC3HIP_NO_FAST") && !env_text("MC3HIP_NO_PERSIST") &&
                                   !env_text("MC3HIP_NO_JIT") && !env_text("MC3HIP_WPC") &&
                                   (pr->flags & MC3HIP_FLAG_JIT);
            c->wide_jit_tiles = 0;
            c->wide_jit_waves = 0;
            if (c->wide_tiles == 0 && shape_jit && tiles >= 2 && tiles <= 16) {
                // a thread's share of the tiles in doubles: NT tiles x 16 / NW points x (x, data,
                // weight -- or K columns, data, weight); up to ~98 leave the loop its registers
                // at two waves per SIMD.  Four waves per chain (two or more workgroups per CU)
                // where that fits, eight (one workgroup per CU: 256 chains) otherwise
                const long long per_point = linear_model ? pr->nmodel + 2 : 3;
                const int nt = tiles <= 2 ? 2 : (tiles <= 4 ? 4 : (tiles <= 8 ? 8 : 16));
                const long long d4 = (long long)nt * 4 * per_point, d8 = (long long)nt * 2 * per_point;
                if (nt <= 8 && d4 <= 98 &&
                    pr->nlocal <= (d4 <= 34 ? 3 : (d4 <= 60 ? 2 : 1)) * cus) {
                    c->wide_jit_tiles = nt;
                    c->wide_jit_waves = 4;
                } else if (d8 <= 98 && pr->nlocal <= cus) {
                    // (sixteen tiles, or what four waves do not hold: instead of the general
                    // chain or of the tiled plan -- chosen here, so only for a caller that will
                    // specialise)
                    c->wide_jit_tiles = nt;
                    c->wide_jit_waves = 8;
                }
            }
        }
        // (design matrices beyond what a loop specialised at run time keeps in registers -- ten
        // columns from 4 097 points, five from 8 193, twelve from 2 049 -- ran the general chain up to 16 383
        // points: 21-37 us per iteration of 256 chains at 8 192, where the one-launch tiled plan
        // through the matrix cores takes ~7)
        const bool big = c->wide_tiles == 0 && c->wide_jit_tiles == 0 &&
                         (nd >= 4 * 4096 || (mid_resident && nd >= 2 * 4096 &&
                                             (long long)nd * pr->nlocal >= 4194304LL) ||
                          (mid_resident && mfma_model && nd > 2048));
        const bool tiled = !pr->wlike && big && pr->nlocal >= 16 && pr->nmodel <= 16 &&
                           !env_text("MC3HIP_NO_TILED");
        P.tile = tiled ? ((pr->nmodel <= 4 && !mfma_model) ? 4096 : 2048) : 0;
        const size_t unit = tiled ? (size_t)P.tile : VLANES;
        const size_t npad = (nd + unit - 1) / unit * unit;
        P.npad = (int32_t)npad;
        P.ntiles = tiled ? (int32_t)(npad / unit) : 0;
        const size_t ncol = (pr->model_kind == MC3HIP_MODEL_LINEAR) ? (size_t)pr->nmodel : 1;
        std::vector<double> buf(npad * ncol);
        for (size_t k = 0; k < ncol; ++k)
            for (size_t i = 0; i < npad; ++i)
                buf[k * npad + i] = pr->indep[k * nd + (i < nd ? i : nd - 1)];
        TRY(dev_upload(c, &P.indep, buf.data(), npad * ncol));
        std::vector<double> d(npad), w(npad, 0.0);
        for (size_t i = 0; i < npad; ++i)
            d[i] = pr->data[i < nd ? i : nd - 1];
        for (size_t i = 0; i < nd; ++i)
            w[i] = pr->wlike ? 1.0 : 1.0 / pr->uncert[i];
        TRY(dev_upload(c, &P.data, d.data(), npad));
        TRY(dev_upload(c, &P.w, w.data(), npad));
        if (tiled && mfma_model) {
            // operand rows of the matrix-core tile kernels, weights folded in (core.h aw)
            std::vector<double> aw((ncol + 1) * npad);
            for (size_t k = 0; k < ncol; ++k)
                for (size_t i = 0; i < npad; ++i) {
                    volatile double prod = buf[k * npad + i] * w[i];    // (rounded: no fma)
                    aw[k * npad + i] = prod;
                }
            for (size_t i = 0; i < npad; ++i) {
                volatile double prod = d[i] * w[i];
                aw[ncol * npad + i] = prod;
            }
            TRY(dev_upload(c, &P.aw, aw.data(), aw.size()));
        }
    }
    // (a padded series, wl_nreal < ndata: the abscissae of the padding are the caller's)
    P.wl_nreal = (pr->wlike && pr->wl_nreal > 0) ? pr->wl_nreal : pr->ndata;
    if (pr->wlike && pr->model_kind == MC3HIP_MODEL_POLY && nd >= 2 && !dev_env_text("MC3HIP_NO_GRID") &&
        P.wl_nreal == pr->ndata) {
        // abscissae made by numpy.linspace (y = arange(num) * step + start, y[-1] = stop)?
        // Checked bit for bit; if so the wavelet kernel recomputes them (block_wavelet)
        const double x0 = pr->indep[0], xl = pr->indep[nd - 1];
        const double step = (xl - x0) / (double)(nd - 1);
        bool grid = step != 0.0 && std::isfinite(step);
        for (size_t i = 0; grid && i + 1 < nd; ++i) {
            volatile double prod = (double)i * step;         // (separately rounded, no fma)
            grid = (prod + x0) == pr->indep[i];
        }
        P.wl_grid = grid ? 1 : 0;
        P.wl_x0 = x0;
        P.wl_step = step;
        P.wl_xlast = xl;
    }
    if (pr->wlike && pr->ndata > 16384) {
        // pre-level values of the large-data path: two ping-pong buffers per workgroup; the
        // chisq-evaluation kernel runs at most WL_EVAL_BLOCKS vectors per launch
        const size_t blocks = (size_t)(pr->nlocal > WL_EVAL_BLOCKS ? pr->nlocal : WL_EVAL_BLOCKS);
        TRY(check_device_room(blocks * (3 * nd / 4) * sizeof(double),
                              "the wavelet likelihood's pre-level scratch (max(nlocal, 256) x "
                              "3/4 ndata doubles)"));
        TRY(dev_alloc(c, &P.wl_scratch, blocks * (3 * nd / 4)));
    }
    if (pr->wlike) {
        // the data in the blocked layout of the wavelet kernel, [j][t] = point B t + j
        const int T = 64 * choose_wpc(pr->nlocal, pr->ndata, 1);
        const int Ta = (pr->ndata / 2 < T) ? (pr->ndata / 2 > 0 ? pr->ndata / 2 : 1) : T;
        const int B = pr->ndata / Ta;
        std::vector<double> dt((size_t)B * T, 0.0);
        for (int t = 0; t < Ta; ++t)
            for (int j = 0; j < B; ++j)
                dt[(size_t)j * T + t] = pr->data[(size_t)B * t + j];
        TRY(dev_upload(c, &P.data_t, dt.data(), dt.size()));
    }
    const size_t zl = (size_t)pr->zlen, nl = (size_t)pr->nlocal;
    P.zs = (int32_t)(((nf + ZROW - 1) / ZROW) * ZROW);       // whole 128-byte lines per row
    const size_t zs = (size_t)P.zs;
    if (pr->flags & MC3HIP_FLAG_P2P) {
        // replicas other GPUs store into while kernels run: fine-grained (coherent at
        // system scope), own allocations so that they can be exported through IPC
        // Opening the IPC handle of an allocation whose size has bit 31 set (2..4 GiB,
        // 6..8 GiB, ...) from another process never returned (ROCm 7.2, dmabuf IPC: 2.5, 3.1
        // and 6.0 GiB hang; 1.9, 4.0 and 5.0 GiB are fine), so such a replica is rounded up to
        // the next multiple of 4 GiB.
        auto ipc_safe = [](size_t bytes) {
            return (bytes & (size_t(1) << 31)) ? ((bytes >> 32) + 1) << 32 : bytes;
        };
        void *pz = nullptr, *pl = nullptr;
        HIP_TRY_C(hipExtMallocWithFlags(&pz, ipc_safe(zl * zs * sizeof(double)),
                                        hipDeviceMallocFinegrained));
        c->allocs.push_back(pz);
        HIP_TRY_C(hipExtMallocWithFlags(&pl, ipc_safe(zl * sizeof(double)),
                                        hipDeviceMallocFinegrained));
        c->allocs.push_back(pl);
        P.Z = static_cast<double *>(pz);
        P.log_post = static_cast<double *>(pl);
    } else {
        TRY(dev_alloc(c, &P.Z, zl * zs));
        TRY(dev_alloc(c, &P.log_post, zl));
    }
    TRY(dev_alloc(c, &P.zchain, zl));
    TRY(dev_alloc(c, &P.X, nl * nf));
    TRY(dev_alloc(c, &P.chisq, nl));
    TRY(dev_alloc(c, &P.numaccept, nl));
    TRY(dev_alloc(c, &P.outbounds, nf));
    TRY(dev_alloc(c, &P.best_chisq, nl));
    TRY(dev_alloc(c, &P.best_iter, nl));
    TRY(dev_alloc(c, &P.bestx, nl * nf));
    {
        double *snap = nullptr;
        TRY(dev_alloc(c, &snap, (size_t)pr->nchains * zs));
        P.snap0 = snap;
    }
    if (P.trace_iters > 0) {
        TRY(dev_alloc(c, &P.trace_chisq, (size_t)P.trace_iters * nl));
        TRY(dev_alloc(c, &P.trace_flag, (size_t)P.trace_iters * nl));
    }
    if (P.tile > 0) {
        TRY(dev_alloc(c, &P.sc_nextp, nl * (size_t)SC_STRIDE));
        HIP_TRY_C(hipMemset(P.sc_nextp, 0, nl * (size_t)SC_STRIDE * sizeof(double)));
        TRY(dev_alloc(c, &P.sc_npk, nl * nf));
        TRY(dev_alloc(c, &P.sc_scal, nl * 2));
        TRY(dev_alloc(c, &P.sc_flag, nl));
        TRY(dev_alloc(c, &P.sc_part, nl * (size_t)P.ntiles));
        if ((pr->model_kind == MC3HIP_MODEL_LINEAR && P.tile == 2048) ||
            (pr->model_kind != MC3HIP_MODEL_LINEAR && P.tile == 4096 && pr->nmodel >= 2)) {
            // hand-off areas of the resident tiled plan, all-ones = "not there yet"
            P.rs_ntp = (P.ntiles + 15) / 16 * 16;
            TRY(dev_alloc(c, &P.rs_prop, 2 * nl * (size_t)RS_ROW));
            TRY(dev_alloc(c, &P.rs_part, nl * (size_t)P.rs_ntp));
            HIP_TRY_C(hipMemset(P.rs_prop, 0xff, 2 * nl * (size_t)RS_ROW * sizeof(double)));
            HIP_TRY_C(hipMemset(P.rs_part, 0xff, nl * (size_t)P.rs_ntp * sizeof(double)));
            TRY(dev_alloc(c, &P.rs_tag, nl));
            HIP_TRY_C(hipMemset(P.rs_tag, 0, nl * sizeof(unsigned long long)));
        }
        c->tile_kmax = pr->nmodel <= 4 ? 4 : pr->nmodel <= 8 ? 8 : pr->nmodel <= 12 ? 12 : 16;
        // chains per workgroup column (a multiple of the 16-chain block): one workgroup fits
        // a CU (128 KiB of LDS), so the kernel runs in rounds of `CUs` workgroups; pick the
        // column height with the fewest (rounds x blocks per workgroup), the tile load
        // counted as a fifth of a block.  C5: 49 tiles x 512 chains -> 112 chains per column,
        // 245 workgroups, one round of 7 blocks instead of two rounds of 4.
        int cus = 256;
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, c->device) == hipSuccess &&
                prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
        }
        int best_cb = 16;
        double best_cost = 1e300;
        for (int cb = 16; cb <= (int)((nl + 15) / 16 * 16); cb += 16) {
            const long long ncol = ((long long)nl + cb - 1) / cb;
            const long long rounds = ((long long)P.ntiles * ncol + cus - 1) / cus;
            const double cost = (double)rounds * (cb / 16 + 0.2);
            if (cost < best_cost) {
                best_cost = cost;
                best_cb = cb;
            }
        }
        c->tile_chain_block = best_cb;
    }
    P.test_abort = -1;
    if (const char *ta = env_text("MC3HIP_TEST_ABORT_AT"))
        P.test_abort = atoi(ta);
    P.zlag = pr->zlag;
    // test hook: rows for another rank's replica are held back (ticks of the 100 MHz clock)
    P.remote_delay = 0;
    if (const char *rd = env_text("MC3HIP_TEST_REMOTE_DELAY_NS"))
        P.remote_delay = atoi(rd) > 0 ? (atoi(rd) + 5) / 10 : 0;
    // test hook: one column of resident_async_kernel's tile waves held back before every block
    P.test_slow_col = P.test_slow_ticks = 0;
    if (const char *sc = env_text("MC3HIP_TEST_SLOW_COLUMN")) {
        int col = 0, ticks = 0;
        if (sscanf(sc, "%d:%d", &col, &ticks) == 2 && col >= 0 && ticks > 0) {
            P.test_slow_col = col;
            P.test_slow_ticks = ticks;
        }
    }
    {
        // checkpoint area: X, bestx [nl*nf]; chisq, best_chisq [nl]; numaccept,
        // best_iter [nl]; outbounds [nf] -- all 8-byte words
        char *ck = nullptr;
        TRY(dev_alloc(c, &ck, (2 * nl * nf + 4 * nl + nf) * 8));
        c->ckpt = ck;
    }
    TRY(dev_alloc(c, &c->ctl, 4));
    P.ctl = c->ctl;
    // (32 phase stamps; the rest is the timeline of one iteration that a -DMC3HIP_TIMELINE
    // build of a kernel writes: tools/timeline_c5.py)
    TRY(dev_alloc(c, &P.stamps, 65536));
    HIP_TRY_C(hipMemset(P.stamps, 0, 65536 * sizeof(unsigned long long)));
    HIP_TRY_C(hipMemset(c->ctl, 0, 4 * sizeof(long long)));
    // all-ones = "row not written yet" (the persistent kernel's hand-off sentinel)
    HIP_TRY_C(hipMemset(P.Z, 0xff, zl * zs * sizeof(double)));
    HIP_TRY_C(hipMemset(P.log_post, 0, zl * sizeof(double)));
    HIP_TRY_C(hipMemset(P.zchain, 0xff, zl * sizeof(int32_t)));      // -1, mcmc_driver.py:170
    HIP_TRY_C(hipMemset(P.numaccept, 0, nl * sizeof(long long)));
    HIP_TRY_C(hipMemset(P.outbounds, 0, nf * sizeof(unsigned long long)));
    HIP_TRY_C(hipMemset(P.bestx, 0, nl * nf * sizeof(double)));
    // the fills ran on the null stream; everything else runs on c->stream, which does not
    // synchronise with it
    HIP_TRY_C(hipDeviceSynchronize());
    TRY(dev_alloc(c, &c->model_out, nd));

    if (pr->model_kind == MC3HIP_MODEL_LINEAR)
        c->mt = MT_LINEAR;
    else if (pr->nmodel >= 2 && pr->nmodel <= 4)
        c->mt = pr->nmodel;
    else
        c->mt = MT_POLY_GENERIC;
    c->wpc = choose_wpc(pr->nlocal, pr->ndata, pr->wlike);
    c->smem = smem_bytes(64 * c->wpc, pr->ndata, pr->wlike);
    if (c->smem > 160 * 1024) {
        mc3hip_destroy(c);
        return fail(MC3HIP_ERR_ARG, "problem needs %zu bytes of LDS (> 160 KiB)", c->smem);
    }
    TRY(set_smem_attr(c));
    if (P.tile > 0) {
        if (tile_set_attr(c) != hipSuccess) {
            mc3hip_destroy(c);
            return fail(MC3HIP_ERR_HIP, "hipFuncSetAttribute failed for the tile kernel");
        }
        if (pr->thinning > 16)
            c->flags |= MC3HIP_FLAG_NO_GRAPH;       // keep the captured graph small
    }
    c->graph_nodes = (c->flags & MC3HIP_FLAG_NO_GRAPH) ? 0 : 256;
    if (env_text("MC3HIP_NO_GRAPH"))
        c->graph_nodes = 0;
    if (const char *g = env_text("MC3HIP_GRAPH_NODES"))
        c->graph_nodes = atoi(g) > 0 ? atoi(g) : 0;
#undef TRY
#undef HIP_TRY_C
    *out = c;
    return MC3HIP_OK;
}

void mc3hip_destroy(mc3hip_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    // (the graph that captured the all-gather goes BEFORE the communicator: RCCL counts the
    // captured launches as references to it, and ncclCommDestroy waits for the last one --
    // for ever, with more than one rank; found in round 6, the first time the collective ran
    // with two ranks, tools/rccl_ranks_as_hosts_try.py)
    if (c->cgraph_exec)
        (void)hipGraphExecDestroy(c->cgraph_exec);
    if (c->cgraph)
        (void)hipGraphDestroy(c->cgraph);
    c->cgraph_exec = nullptr;
    c->cgraph = nullptr;
    if (c->comm && g_rccl.CommDestroy)
        g_rccl.CommDestroy(c->comm);
    for (void *p : c->p2p_opened)
        (void)hipIpcCloseMemHandle(p);
    if (c->graph_exec)
        (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph)
        (void)hipGraphDestroy(c->graph);
    for (void *p : c->allocs)
        (void)hipFree(p);
    if (c->stage)
        (void)hipFree(c->stage);
    if (c->stage2)
        (void)hipFree(c->stage2);
    if (c->copy_stream)
        (void)hipStreamDestroy(c->copy_stream);
    if (c->ev0)
        (void)hipEventDestroy(c->ev0);
    if (c->ev1)
        (void)hipEventDestroy(c->ev1);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

int mc3hip_get_plan(const mc3hip_ctx *c, mc3hip_plan *plan)
{
    if (!c || !plan)
        return fail(MC3HIP_ERR_ARG, "null argument");
    plan->vlanes = VLANES;
    plan->tile = c->P.tile;
    plan->waves_per_chain = c->wpc;
    plan->graph_nodes = c->graph_nodes;
    plan->persistent = c->comm ? 0 : decide_persist(const_cast<mc3hip_ctx *>(c));
    plan->fast = (plan->persistent > 0 && c->fast)
                     ? ((lean_demc_eligible(c) || lean_demc_mc_eligible(c) ||
                         lean_snooker_eligible(c) || lean_mrw_eligible(c) ||
                         lean_demc_lag_eligible(c) || lean_rows_eligible(c) ||
                         jit_serves(c)) ? 2 : 1) : 0;
    plan->jit = (plan->fast == 2 && jit_serves(c)) ? 1 : 0;
    plan->capacity = plan->persistent > 0
                         ? (int32_t)(c->capacity < 0x7fffffff ? c->capacity : 0x7fffffff) : 0;
    if (plan->fast == 2 && c->p2p_ranks <= 1)
        plan->waves_per_chain = plan->jit ? jit_waves(c)
                                          : lean_waves(c);  // (helper waves share the data sum)
    if (plan->persistent > 0 && chainwave_runs(c, c->p2p_ranks > 1) && c->P.cw_lean)
        plan->fast = 3;                             // (wavelet: the hand-structured chain wave)
    if (plan->persistent > 0 && c->wl_one && c->mc_chains == 0 && c->p2p_ranks <= 1)
        plan->fast = 3;                             // (... and its one-wave form)
    if (plan->persistent > 0 && c->resident && c->P.rs_lean)
        plan->fast = 4;                             // (resident plans: the hand-structured owner)
    plan->tile_tree = tile_uses_mfma(c) ? 2 : 0;
    plan->resident = (plan->persistent > 0 && c->resident) ? 1 : 0;
    plan->comm_graph = c->cgraph_exec ? c->comm_graph_nodes : 0;
    plan->recoveries = c->recoveries;
    plan->chains_per_workgroup = (plan->persistent > 0 && c->mc_chains > 0) ? c->mc_chains : 1;
    plan->zlag = c->P.zlag;
    plan->remote_delay_ns = c->P.remote_delay * 10;
    plan->owner_waves = (plan->persistent > 0 && c->resident) ? c->resident_owner_waves : 0;
    return MC3HIP_OK;
}

// ---- run-time specialisation (include/mc3hip.h) -------------------------------------------
int mc3hip_jit_request(const mc3hip_ctx *c, char *buf, int32_t capacity)
{
    if (!c || !buf || capacity < 1)
        return fail(MC3HIP_ERR_ARG, "null argument");
    buf[0] = 0;
    // (what the library has an instance for, it runs: nothing to specialise -- but for the
    // random walk beyond 1024 points, whose library instance streams the tiles it cannot keep)
    if (fast_eligible(c) && !(c->wide_jit_tiles > 0 && c->P.sampler == MC3HIP_MRW))
        return MC3HIP_OK;
    const int tag = jit_model_tag(c);
    if (tag == 0)
        return MC3HIP_OK;
    const DevProblem &P = c->P;
    const bool plain = P.nfree == P.nparams && P.ngprior == 0 && P.nshare == 0;
    const char *fx = plain ? "false" : "true", *thin = P.thinning > 1 ? "true" : "false";
    const int nw = jit_waves(c);                 // (helper waves: the workgroup has 64 nw threads)
    int n = 0;
    if (c->wide_jit_tiles > 0) {
        // more than 1024 points: NT tiles in the registers of four waves
        const char *eps = P.fepsilon != 0.0 ? "true" : "false";
        const int nt = c->wide_jit_tiles;
        switch (P.sampler) {
        case MC3HIP_DEMC:      // <MT, P2P, FX, MULTI, BATCH, COURIER, EPS, NW, NT>
            if (P.thinning > 1)                  // <MT, P2P, FX, COURIER, NW, NT>
                n = snprintf(buf, (size_t)capacity,
                             "lean_demc_lag_body<%d, false, %s, false, %d, %d> /*threads=%d*/",
                             tag, fx, nw, nt, 64 * nw);
            else
            n = snprintf(buf, (size_t)capacity,
                         "lean_demc_body<%d, false, %s, false, false, false, %s, %d, %d> "
                         "/*threads=%d*/", tag, fx, eps, nw, nt, 64 * nw);
            break;
        case MC3HIP_SNOOKER:   // <MT, FX, THIN, MULTI, P2P, COURIER, EPS, NW, NT>
            n = snprintf(buf, (size_t)capacity,
                         "lean_snooker_body<%d, %s, %s, false, false, false, %s, %d, %d> "
                         "/*threads=%d*/", tag, fx, thin, eps, nw, nt, 64 * nw);
            break;
        default:               // <MT, FX, THIN, MULTI, P2P, COURIER, NW, NT>
            n = snprintf(buf, (size_t)capacity,
                         "lean_mrw_body<%d, %s, %s, false, false, false, %d, %d> "
                         "/*threads=%d*/", tag, fx, thin, nw, nt, 64 * nw);
            break;
        }
        if (n < 0 || n >= capacity) {
            buf[0] = 0;
            return fail(MC3HIP_ERR_ARG, "buffer of %d bytes too small for the request", capacity);
        }
        return MC3HIP_OK;
    }
    if (tag <= 4) {
        // the leftovers of the polynomials of 2..4 coefficients (ctx.h jit_model_tag): one wave
        const char *multi = P.npad > VLANES ? "true" : "false";
        switch (P.sampler) {
        case MC3HIP_DEMC:      // <MT, P2P, FX, MULTI, BATCH, COURIER, EPS>
            n = snprintf(buf, (size_t)capacity,
                         "lean_demc_body<%d, false, %s, %s, false, false, false>", tag, fx, multi);
            break;
        case MC3HIP_SNOOKER:   // <MT, FX, THIN, MULTI, P2P, COURIER, EPS>
            n = snprintf(buf, (size_t)capacity,
                         "lean_snooker_body<%d, %s, %s, %s, false, false, false>", tag, fx, thin,
                         multi);
            break;
        default:               // <MT, FX, THIN, MULTI, P2P, COURIER>
            n = snprintf(buf, (size_t)capacity, "lean_mrw_body<%d, %s, %s, %s, false, false>", tag,
                         fx, thin, multi);
            break;
        }
        if (n < 0 || n >= capacity) {
            buf[0] = 0;
            return fail(MC3HIP_ERR_ARG, "buffer of %d bytes too small for the request", capacity);
        }
        return MC3HIP_OK;
    }
    switch (P.sampler) {
    case MC3HIP_DEMC:
        if (P.thinning > 1 || P.zlag > 0)        // <MT, P2P, FX, COURIER>
            n = snprintf(buf, (size_t)capacity, "lean_demc_lag_body<%d, false, %s, false>", tag, fx);
        else if (nw > 1)                         // <MT, P2P, FX, MULTI, BATCH, COURIER, EPS, NW>
            n = snprintf(buf, (size_t)capacity,
                         "lean_demc_body<%d, false, %s, false, false, false, %s, %d> "
                         "/*threads=%d*/", tag, fx, P.fepsilon != 0.0 ? "true" : "false", nw,
                         64 * nw);
        else                                     // <MT, P2P, FX, MULTI, BATCH, COURIER, EPS>
            n = snprintf(buf, (size_t)capacity,
                         "lean_demc_body<%d, false, %s, false, false, false, %s>", tag, fx,
                         P.fepsilon != 0.0 ? "true" : "false");
        break;
    case MC3HIP_SNOOKER:                         // <MT, FX, THIN, MULTI, P2P, COURIER, EPS[, NW]>
        if (nw > 1)
            n = snprintf(buf, (size_t)capacity,
                         "lean_snooker_body<%d, %s, %s, false, false, false, %s, %d> "
                         "/*threads=%d*/", tag, fx, thin, P.fepsilon != 0.0 ? "true" : "false",
                         nw, 64 * nw);
        else
            n = snprintf(buf, (size_t)capacity,
                         "lean_snooker_body<%d, %s, %s, false, false, false, %s>", tag, fx, thin,
                         P.fepsilon != 0.0 ? "true" : "false");
        break;
    default:                                     // <MT, FX, THIN, MULTI, P2P, COURIER[, NW, NT]>
        if (nw > 1)
            n = snprintf(buf, (size_t)capacity,
                         "lean_mrw_body<%d, %s, %s, false, false, false, %d, 1> /*threads=%d*/",
                         tag, fx, thin, nw, 64 * nw);
        else
            n = snprintf(buf, (size_t)capacity, "lean_mrw_body<%d, %s, %s, false, false, false>",
                         tag, fx, thin);
        break;
    }
    if (n < 0 || n >= capacity) {
        buf[0] = 0;
        return fail(MC3HIP_ERR_ARG, "buffer of %d bytes too small for the request", capacity);
    }
    return MC3HIP_OK;
}

// Code objects handed over by mc3hip_jit_install, per device: loaded once, shared by every
// context that installs the same image, and never unloaded.  (Unloading with the context was
// seen -- tools/fuzz_hunt.py, one process creating and destroying contexts in a row, 3 runs in
// 16 -- to let the NEXT context's first allocation land on the module's address range while
// the pages were still mapped read-only: "Memory access fault ... Write access to a read-only
// page" in a kernel that writes nothing but its own buffers.  A handful of 20 KB images per
// process is what it costs.)
namespace {
struct LoadedModule {
    int device;
    std::string image;
    hipModule_t module;
};
std::mutex g_modules_lock;
std::vector<LoadedModule> g_modules;
}  // namespace

int mc3hip_jit_install(mc3hip_ctx *c, const void *code_object, int64_t bytes,
                       const char *kernel_name)
{
    if (!c || !code_object || bytes <= 0 || !kernel_name)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (c->refresh_done != 0 || c->ckpt_valid)
        return fail(MC3HIP_ERR_STATE, "a kernel is installed before the first mc3hip_run");
    if (jit_model_tag(c) == 0)
        return fail(MC3HIP_ERR_ARG, "this context's shape has no specialised loop "
                                    "(mc3hip_jit_request says which do)");
    HIP_TRY(hipSetDevice(c->device));
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipError_t e = hipSuccess;
    {
        std::lock_guard<std::mutex> hold(g_modules_lock);
        const std::string image(static_cast<const char *>(code_object), (size_t)bytes);
        for (const LoadedModule &m : g_modules)
            if (m.device == c->device && m.image == image)
                mod = m.module;
        if (!mod) {
            e = hipModuleLoadData(&mod, code_object);
            if (e != hipSuccess)
                return fail(MC3HIP_ERR_HIP, "hipModuleLoadData failed: %s", hipGetErrorString(e));
            g_modules.push_back(LoadedModule{c->device, image, mod});
        }
    }
    e = hipModuleGetFunction(&fn, mod, kernel_name);
    if (e != hipSuccess)
        return fail(MC3HIP_ERR_HIP, "kernel '%s' not in the code object: %s", kernel_name,
                    hipGetErrorString(e));
    c->jit_module = mod;                         // (shared: not this context's to unload)
    c->jit_fn = fn;
    c->jit_name = std::string("jit:") + kernel_name;
    c->persist_checked = false;                  // (the launch structure is decided anew)
    return MC3HIP_OK;
}

int mc3hip_set_population(mc3hip_ctx *c, const double *Z0, const double *log_post0,
                          const double *X0, const double *chisq0)
{
    if (!c || !Z0 || !log_post0)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if ((X0 == nullptr) != (chisq0 == nullptr))
        return fail(MC3HIP_ERR_ARG, "X0 and chisq0 must be given together");
    HIP_TRY(hipSetDevice(c->device));
    const DevProblem &P = c->P;
    const size_t nf = (size_t)P.nfree, M0 = (size_t)P.M0, nc = (size_t)P.nchains,
                 nl = (size_t)P.nlocal;
    RC_TRY(rows_to_device(c, P.Z, Z0, M0));
    HIP_TRY(hipMemcpy(P.log_post, log_post0, M0 * sizeof(double), hipMemcpyHostToDevice));
    // chains start from the first nchains rows of Z: chain.py:163-170
    const double *Xs = X0 ? X0 : Z0;
    std::vector<double> chi(nc);
    for (size_t i = 0; i < nc; ++i)
        chi[i] = chisq0 ? chisq0[i] : -2.0 * log_post0[i];
    RC_TRY(rows_to_device(c, const_cast<double *>(P.snap0), Xs, nc));
    HIP_TRY(hipMemcpy(P.X, Xs + (size_t)P.chain0 * nf, nl * nf * sizeof(double),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P.chisq, chi.data() + P.chain0, nl * sizeof(double),
                      hipMemcpyHostToDevice));
    std::vector<double> inf(nl, INFINITY);
    std::vector<long long> neg(nl, -1);
    HIP_TRY(hipMemcpy(P.best_chisq, inf.data(), nl * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P.best_iter, neg.data(), nl * sizeof(long long), hipMemcpyHostToDevice));
    c->populated = true;
    return MC3HIP_OK;
}

// copy the per-chain state to (save = true) or from the checkpoint area, stream-ordered
static int checkpoint_copy(mc3hip_ctx *c, bool save)
{
    const DevProblem &P = c->P;
    const size_t nl = (size_t)P.nlocal, nf = (size_t)P.nfree;
    // (one kernel: see checkpoint_kernel)
    CheckpointItems it;
    void *const ptrs[7] = {P.X, P.bestx, P.chisq, P.best_chisq, P.numaccept, P.best_iter,
                           P.outbounds};
    const size_t words[7] = {nl * nf, nl * nf, nl, nl, nl, nl, nf};
    size_t total = 0;
    for (int k = 0; k < 7; ++k) {
        it.ptr[k] = static_cast<unsigned long long *>(ptrs[k]);
        it.words[k] = words[k];
        total += words[k];
    }
    const int blocks = (int)((total + 255) / 256 < 64 ? (total + 255) / 256 : 64);
    MC3_LAUNCH(checkpoint_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, c->stream,
                       it, reinterpret_cast<unsigned long long *>(c->ckpt), save ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return MC3HIP_OK;
}

static int build_graph(mc3hip_ctx *c);

// RCCL exchange inside a hipGraph (multi-GPU fallback path): `comm_graph_nodes` refreshes
// per replay, each one (step kernel -> ring slot r) + (in-place ncclAllGather of slot r)
// + (ring_in kernel: the peers' rows to their place in Z).  The all-gather's buffers must
// have fixed addresses to be replayed, hence the ring instead of Z itself.  A failure to
// capture (an RCCL build that cannot) demotes the context to eager launches, once.
static int build_comm_graph(mc3hip_ctx *c)
{
    const size_t zs = (size_t)c->P.zs, slot = (size_t)c->P.nchains * zs;
    int G = c->graph_nodes;
    const size_t cap = (size_t(256) << 20) / (slot * sizeof(double));   // ring <= 256 MiB
    if ((size_t)G > cap)
        G = (int)cap;
    if (G < 2) {
        c->comm_graph_failed = true;
        return MC3HIP_OK;
    }
    if (!c->xring) {
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, (size_t)G * slot * sizeof(double)));
        c->allocs.push_back(p);
        c->xring = static_cast<double *>(p);
    }
    c->P.xring = c->xring;
    bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess;
    if (ok) {
        const int nb = (c->P.nchains * c->P.nfree + 255) / 256;
        for (int r = 0; r < G && ok; ++r) {
            launch_step(c, r);
            double *base = c->xring + (size_t)r * slot;
            ok = g_rccl.AllGather(base + (size_t)c->P.chain0 * zs, base,
                                  (size_t)c->P.nlocal * zs, NCCL_FLOAT64, c->comm,
                                  c->stream) == 0;
            MC3_LAUNCH(ring_in_kernel, dim3(nb > 64 ? 64 : nb), dim3(256), 0, c->stream,
                               c->P, r);
        }
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(c->stream, &g);   // always ends the capture
        ok = ok && e == hipSuccess && g != nullptr;
        if (ok)
            ok = hipGraphInstantiate(&c->cgraph_exec, g, nullptr, nullptr, 0) == hipSuccess;
        if (ok)
            c->cgraph = g;
        else if (g)
            (void)hipGraphDestroy(g);
    }
    c->P.xring = nullptr;           // eager launches exchange in place in Z
    if (!ok) {
        (void)hipGetLastError();
        c->comm_graph_failed = true;
        c->cgraph_exec = nullptr;
        return MC3HIP_OK;
    }
    c->comm_graph_nodes = G;
    return MC3HIP_OK;
}

// refreshes [R, Rend) through one launch per refresh (hipGraph batches + eager tail)
static int enqueue_per_refresh(mc3hip_ctx *c, int64_t R, int64_t Rend, long long run0)
{
    bool first = true;
    const bool use_graph = c->graph_nodes > 0 && c->comm == nullptr;
    if (use_graph && Rend - R >= c->graph_nodes && !c->graph_exec) {
        int rc = build_graph(c);
        if (rc)
            return rc;
    }
    const bool want_cgraph = c->graph_nodes > 0 && c->comm != nullptr && !c->comm_graph_failed;
    while (R < Rend) {
        int64_t left = Rend - R;
        MC3_LAUNCH(set_ctl_kernel, dim3(1), dim3(1), 0, c->stream, c->ctl, (long long)R,
                           first ? run0 : -1LL);
        first = false;
        if (use_graph && left >= c->graph_nodes) {
            HIP_TRY(hipGraphLaunch(c->graph_exec, c->stream));
            R += c->graph_nodes;
            continue;
        }
        if (want_cgraph && c->comm_warm && !c->cgraph_exec && !c->comm_graph_failed &&
            left >= 2) {
            int rc = build_comm_graph(c);
            if (rc)
                return rc;
        }
        if (c->cgraph_exec && left >= c->comm_graph_nodes) {
            HIP_TRY(hipGraphLaunch(c->cgraph_exec, c->stream));
            R += c->comm_graph_nodes;
            continue;
        }
        // eager: the whole remainder, or -- before the first capture -- one refresh, whose
        // collective lets RCCL set up its channels outside the capture
        if (want_cgraph && !c->comm_warm)
            left = 1;
        for (int64_t r = 0; r < left; ++r) {
            launch_step(c, (int)r);
            if (c->comm) {
                int rc = exchange_rows(c, R + r);
                if (rc)
                    return rc;
                c->comm_warm = true;
            }
        }
        R += left;
    }
    HIP_TRY(hipGetLastError());
    return MC3HIP_OK;
}

// Wait for the stream; if a persistent launch gave up (a wait for another chain's row
// timed out: its workgroups were not all resident), discard everything since the
// checkpoint and redo it with one launch per refresh.  Results are identical -- both
// paths run the same run_chain -- only slower; the context stays on that path.
static int settle(mc3hip_ctx *c)
{
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->persist_refs <= 0) {
        // (one launch per refresh: nothing can give up, what has run is final)
        if (c->p2p_ranks <= 1)
            c->confirmed_done = c->refresh_done;
        return MC3HIP_OK;
    }
    long long ab = 0;
    HIP_TRY(hipMemcpy(&ab, c->ctl + CTL_ABORT, sizeof ab, hipMemcpyDeviceToHost));
    if (ab == 0) {
        // single GPU: the launches since the checkpoint are complete.  Peer-to-peer: complete
        // HERE; another rank may have given up, so the checkpoint stays until the caller,
        // having agreed with the other ranks, starts the next run
        if (c->p2p_ranks > 1) {
            c->ckpt_settled = true;
        } else {
            c->ckpt_valid = false;
            c->confirmed_done = c->refresh_done;
        }
        return MC3HIP_OK;
    }
    if (c->p2p_ranks > 1)
        return fail(MC3HIP_ERR_ABORTED,
                    "persistent kernel: a wait for another rank's Z row timed out; every "
                    "rank must roll back (mc3hip_p2p_rollback) and redo refreshes [%lld,%lld)",
                    (long long)c->ckpt_R, (long long)c->refresh_done);
    if (!c->ckpt_valid)
        return fail(MC3HIP_ERR_STATE, "persistent kernel gave up and no checkpoint is pending");
    const DevProblem &P = c->P;
    const size_t nf = (size_t)P.nfree;
    const size_t row0 = (size_t)(P.M0 + c->ckpt_R * P.nchains);
    const size_t nrows = (size_t)((c->refresh_done - c->ckpt_R) * P.nchains);
    int rc = checkpoint_copy(c, false);
    if (rc)
        return rc;
    HIP_TRY(hipMemsetAsync(P.Z + row0 * (size_t)P.zs, 0xff,
                           nrows * (size_t)P.zs * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(P.zchain + row0, 0xff, nrows * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemsetAsync(P.log_post + row0, 0, nrows * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->ctl + CTL_ABORT, 0, sizeof(long long), c->stream));
    if (P.rs_prop) {
        HIP_TRY(hipMemsetAsync(P.rs_prop, 0xff, 2 * (size_t)P.nlocal * RS_ROW * sizeof(double),
                               c->stream));
        HIP_TRY(hipMemsetAsync(P.rs_part, 0xff, (size_t)P.nlocal * P.rs_ntp * sizeof(double),
                               c->stream));
    }
    c->resident = false;
    c->persist_refs = 0;            // this context stays on per-refresh launches
    c->persist_checked = true;
    c->fast = false;
    c->ckpt_valid = false;
    c->recoveries += 1;
    rc = enqueue_per_refresh(c, c->ckpt_R, c->refresh_done, c->last_run0);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->confirmed_done = c->refresh_done;
    return MC3HIP_OK;
}

static int build_graph(mc3hip_ctx *c)
{
    HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < c->graph_nodes; ++r)
        launch_step(c, r);
    hipError_t e = hipStreamEndCapture(c->stream, &c->graph);
    if (e != hipSuccess)
        return fail(MC3HIP_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    HIP_TRY(hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0));
    return MC3HIP_OK;
}

int mc3hip_run(mc3hip_ctx *c, int64_t nrefresh)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    if (!c->populated)
        return fail(MC3HIP_ERR_STATE, "mc3hip_set_population has not been called");
    if (nrefresh < 0)
        return fail(MC3HIP_ERR_ARG, "negative nrefresh");
    const DevProblem &P = c->P;
    if (P.M0 + (c->refresh_done + nrefresh) * P.nchains > c->zlen)
        return fail(MC3HIP_ERR_FULL, "Z holds %lld rows: cannot run %lld more refreshes after %lld",
                    (long long)c->zlen, (long long)nrefresh, (long long)c->refresh_done);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    int64_t R = c->refresh_done;
    const int64_t Rend = R + nrefresh;
    const long long run0 = R;
    c->last_run0 = run0;
    const int persist = c->comm ? 0 : decide_persist(c);
    if (persist > 0) {
        if (!c->ckpt_valid || c->ckpt_settled) {
            int rc = checkpoint_copy(c, true);
            if (rc)
                return rc;
            c->ckpt_R = R;
            c->ckpt_valid = true;
            c->ckpt_settled = false;
        }
        bool first = true;
        while (R < Rend) {
            const int64_t left = Rend - R;
            const int n = (int)(left < persist ? left : persist);
            MC3_LAUNCH(set_ctl_kernel, dim3(1), dim3(1), 0, c->stream, c->ctl,
                               (long long)R, first ? run0 : -1LL);
            if (c->resident)
                launch_resident(c, n);
            else if (c->fast)
                launch_fast(c, n);
            else
                launch_persist(c, n);
            R += n;
            first = false;
        }
    } else {
        int rc = enqueue_per_refresh(c, R, Rend, run0);
        if (rc)
            return rc;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    c->refresh_done = Rend;
    return MC3HIP_OK;
}

int mc3hip_sync(mc3hip_ctx *c)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    return settle(c);
}

// Streaming runs: the last `keep` refreshes move to the front of the chain rows, the rest of
// Z is unwritten again, the run carries on under a new Philox key (include/mc3hip.h).
int mc3hip_recycle(mc3hip_ctx *c, int64_t keep, uint64_t seed)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    if (!c->populated)
        return fail(MC3HIP_ERR_STATE, "mc3hip_set_population has not been called");
    DevProblem &P = c->P;
    if (P.sampler == MC3HIP_SNOOKER)
        return fail(MC3HIP_ERR_ARG, "snooker draws from the whole history (chain.py:197-205): "
                                    "its Z cannot be recycled");
    if (keep < 1 + (int64_t)P.zlag)
        return fail(MC3HIP_ERR_ARG, "keep=%lld: the next refresh looks %d refreshes back",
                    (long long)keep, 1 + P.zlag);
    {
        const int rc_settle = settle(c);
        if (rc_settle)
            return rc_settle;
    }
    const int64_t done = c->refresh_done;
    if (done < 2 * keep)
        return fail(MC3HIP_ERR_STATE, "only %lld refreshes done: nothing to recycle with keep=%lld",
                    (long long)done, (long long)keep);
    if (c->comm)
        RC_TRY(mc3hip_comm_gather_log_post(c));
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)P.nchains, zs = (size_t)P.zs;
    const size_t from = (size_t)(P.M0 + (done - keep) * P.nchains), to = (size_t)P.M0;
    const size_t rows = (size_t)keep * n, rest = (size_t)(done - keep) * n;
    HIP_TRY(hipMemcpyAsync(P.Z + to * zs, P.Z + from * zs, rows * zs * sizeof(double),
                           hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(P.log_post + to, P.log_post + from, rows * sizeof(double),
                           hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(P.zchain + to, P.zchain + from, rows * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, c->stream));
    // (unwritten again: the all-ones pattern the hand-off reads as "not there yet")
    HIP_TRY(hipMemsetAsync(P.Z + (to + rows) * zs, 0xff, rest * zs * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(P.zchain + to + rows, 0xff, rest * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemsetAsync(P.log_post + to + rows, 0, rest * sizeof(double), c->stream));
    if (P.rs_prop) {
        HIP_TRY(hipMemsetAsync(P.rs_prop, 0xff, 2 * (size_t)P.nlocal * RS_ROW * sizeof(double),
                               c->stream));
        HIP_TRY(hipMemsetAsync(P.rs_part, 0xff, (size_t)P.nlocal * P.rs_ntp * sizeof(double),
                               c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    // captured launches carry the problem (and its key) by value: capture again
    if (c->graph_exec) {
        (void)hipGraphExecDestroy(c->graph_exec);
        c->graph_exec = nullptr;
    }
    if (c->graph) {
        (void)hipGraphDestroy(c->graph);
        c->graph = nullptr;
    }
    if (c->cgraph_exec) {
        (void)hipGraphExecDestroy(c->cgraph_exec);
        c->cgraph_exec = nullptr;
    }
    if (c->cgraph) {
        (void)hipGraphDestroy(c->cgraph);
        c->cgraph = nullptr;
    }
    P.seed = seed;
    c->refresh_done = keep;
    c->confirmed_done = keep;       // (the rows moved: what mc3hip_get_rows_settled may read)
    c->ckpt_valid = false;
    c->ckpt_settled = false;
    c->logpost_gathered = keep;
    return MC3HIP_OK;
}

int mc3hip_last_run_ms(mc3hip_ctx *c, float *ms)
{
    if (!c || !ms)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (!c->timed)
        return fail(MC3HIP_ERR_STATE, "no run to time yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return MC3HIP_OK;
}

int mc3hip_get_rows(mc3hip_ctx *c, int64_t row0, int64_t nrows, double *Z, double *log_post,
                    int32_t *zchain)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    if (row0 < 0 || nrows < 0 || row0 + nrows > c->zlen)
        return fail(MC3HIP_ERR_ARG, "rows [%lld,%lld) outside Z", (long long)row0,
                    (long long)(row0 + nrows));
    {
        const int rc_settle = settle(c);
        if (rc_settle)
            return rc_settle;
    }
    if (Z)
        RC_TRY(rows_to_host(c, Z, c->P.Z + (size_t)row0 * (size_t)c->P.zs, (size_t)nrows));
    if (log_post)
        HIP_TRY(hipMemcpy(log_post, c->P.log_post + row0, (size_t)nrows * sizeof(double),
                          hipMemcpyDeviceToHost));
    if (zchain) {
        HIP_TRY(hipMemcpy(zchain, c->P.zchain + row0, (size_t)nrows * sizeof(int32_t),
                          hipMemcpyDeviceToHost));
        if (c->P.nlocal < c->P.nchains) {
            // a shard only marks its own rows; the chain of a written row is its
            // position in the refresh (chain.py:288 strided slot)
            const int64_t done_end = c->P.M0 + c->refresh_done * c->P.nchains;
            for (int64_t r = row0 > c->P.M0 ? row0 : c->P.M0; r < row0 + nrows && r < done_end; ++r)
                zchain[r - row0] = (int32_t)((r - c->P.M0) % c->P.nchains);
        }
    }
    return MC3HIP_OK;
}

/* Rows that can no longer change -- the initial population and the refreshes an mc3hip_sync
 * has confirmed (single GPU; a sharded run confirms by agreement of the ranks and uses
 * mc3hip_get_rows) -- read back on a stream of their own: the call neither waits for nor
 * disturbs a run that is in flight, so the host copy of a long run is made while the next
 * segment computes (mc3_amd/mcmc_driver.py).  MC3HIP_ERR_STATE for rows beyond the confirmed. */
int mc3hip_get_rows_settled(mc3hip_ctx *c, int64_t row0, int64_t nrows, double *Z, double *log_post,
                            int32_t *zchain)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    if (row0 < 0 || nrows < 0 || row0 + nrows > c->zlen)
        return fail(MC3HIP_ERR_ARG, "rows [%lld,%lld) outside Z", (long long)row0,
                    (long long)(row0 + nrows));
    if (c->p2p_ranks > 1 || c->comm != nullptr || c->P.nlocal < c->P.nchains)
        return fail(MC3HIP_ERR_STATE, "mc3hip_get_rows_settled serves one-GPU runs");
    if (row0 + nrows > c->P.M0 + c->confirmed_done * c->P.nchains)
        return fail(MC3HIP_ERR_STATE, "rows [%lld,%lld): only %lld refreshes are confirmed",
                    (long long)row0, (long long)(row0 + nrows), (long long)c->confirmed_done);
    HIP_TRY(hipSetDevice(c->device));
    if (!c->copy_stream)
        HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (Z) {
        const size_t nf = (size_t)c->P.nfree, zs = (size_t)c->P.zs;
        const size_t chunk = STAGE_MAX_DOUBLES / nf ? STAGE_MAX_DOUBLES / nf : 1;
        const double *padded = c->P.Z + (size_t)row0 * zs;
        for (size_t r = 0; r < (size_t)nrows; r += chunk) {
            const size_t m = (size_t)nrows - r < chunk ? (size_t)nrows - r : chunk;
            if (m * nf > c->stage2_doubles) {
                if (c->stage2)
                    HIP_TRY(hipFree(c->stage2));
                c->stage2 = nullptr;
                c->stage2_doubles = 0;
                void *p = nullptr;
                HIP_TRY(hipMalloc(&p, m * nf * sizeof(double)));
                c->stage2 = static_cast<double *>(p);
                c->stage2_doubles = m * nf;
            }
            const int blocks = (int)((m * nf + 255) / 256 < 4096 ? (m * nf + 255) / 256 : 4096);
            MC3_LAUNCH(pack_rows_kernel, dim3(blocks), dim3(256), 0, c->copy_stream, c->stage2,
                       padded + r * zs, (long long)m, (int)nf, (int)zs);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(Z + r * nf, c->stage2, m * nf * sizeof(double),
                                   hipMemcpyDeviceToHost, c->copy_stream));
            HIP_TRY(hipStreamSynchronize(c->copy_stream));
        }
    }
    if (log_post)
        HIP_TRY(hipMemcpyAsync(log_post, c->P.log_post + row0, (size_t)nrows * sizeof(double),
                               hipMemcpyDeviceToHost, c->copy_stream));
    if (zchain)
        HIP_TRY(hipMemcpyAsync(zchain, c->P.zchain + row0, (size_t)nrows * sizeof(int32_t),
                               hipMemcpyDeviceToHost, c->copy_stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
    return MC3HIP_OK;
}

int mc3hip_put_rows(mc3hip_ctx *c, int64_t row0, int64_t nrows, const double *Z,
                    const double *log_post)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    if (row0 < c->P.M0 || nrows < 0 || row0 + nrows > c->zlen)
        return fail(MC3HIP_ERR_ARG, "rows [%lld,%lld) outside the chain rows of Z",
                    (long long)row0, (long long)(row0 + nrows));
    {
        const int rc_settle = settle(c);
        if (rc_settle)
            return rc_settle;
    }
    if (Z)
        RC_TRY(rows_to_device(c, c->P.Z + (size_t)row0 * (size_t)c->P.zs, Z, (size_t)nrows));
    if (log_post)
        HIP_TRY(hipMemcpy(c->P.log_post + row0, log_post, (size_t)nrows * sizeof(double),
                          hipMemcpyHostToDevice));
    return MC3HIP_OK;
}

int mc3hip_get_state(mc3hip_ctx *c, double *X, double *chisq, int64_t *numaccept,
                     int64_t *outbounds, double *best_chisq, int64_t *best_iter, double *bestx,
                     int64_t *refresh_done)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    {
        const int rc_settle = settle(c);
        if (rc_settle)
            return rc_settle;
    }
    const DevProblem &P = c->P;
    const size_t nf = (size_t)P.nfree, nl = (size_t)P.nlocal;
    if (X)
        HIP_TRY(hipMemcpy(X, P.X, nl * nf * sizeof(double), hipMemcpyDeviceToHost));
    if (chisq)
        HIP_TRY(hipMemcpy(chisq, P.chisq, nl * sizeof(double), hipMemcpyDeviceToHost));
    if (numaccept)
        HIP_TRY(hipMemcpy(numaccept, P.numaccept, nl * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (outbounds)
        HIP_TRY(hipMemcpy(outbounds, P.outbounds, nf * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (best_chisq)
        HIP_TRY(hipMemcpy(best_chisq, P.best_chisq, nl * sizeof(double), hipMemcpyDeviceToHost));
    if (best_iter)
        HIP_TRY(hipMemcpy(best_iter, P.best_iter, nl * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (bestx)
        HIP_TRY(hipMemcpy(bestx, P.bestx, nl * nf * sizeof(double), hipMemcpyDeviceToHost));
    if (refresh_done)
        *refresh_done = c->refresh_done;
    return MC3HIP_OK;
}

int mc3hip_get_trace(mc3hip_ctx *c, int64_t iters, double *chisq, int32_t *flags)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    if (iters < 0 || iters > c->P.trace_iters)
        return fail(MC3HIP_ERR_ARG, "trace holds %d iterations", c->P.trace_iters);
    {
        const int rc_settle = settle(c);
        if (rc_settle)
            return rc_settle;
    }
    const size_t n = (size_t)iters * (size_t)c->P.nlocal;
    if (chisq)
        HIP_TRY(hipMemcpy(chisq, c->P.trace_chisq, n * sizeof(double), hipMemcpyDeviceToHost));
    if (flags)
        HIP_TRY(hipMemcpy(flags, c->P.trace_flag, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MC3HIP_OK;
}

static int chain_moments(mc3hip_ctx *c, int64_t nrefresh, int64_t burn, double *mean,
                         double *var)
{
    {
        const int rc_settle = settle(c);
        if (rc_settle)
            return rc_settle;
    }
    const size_t n = (size_t)c->P.nchains * (size_t)c->P.nfree;
    if (!c->moments)
        RC_TRY(dev_alloc(c, &c->moments, 2 * n));       // freed with the context
    double *dm = c->moments, *dv = dm + n;
    MC3_LAUNCH(chain_moments_kernel, dim3(c->P.nchains), dim3(256), 0, c->stream,
                       (const double *)c->P.Z, (long long)c->P.M0, c->P.nchains, c->P.nfree,
                       c->P.zs, (long long)nrefresh, (long long)burn, dm, dv);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(mean, dm, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(var, dv, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
        return fail(MC3HIP_ERR_HIP, "chain_moments: %s", hipGetErrorString(e));
    return MC3HIP_OK;
}

int mc3hip_chain_moments(mc3hip_ctx *c, int64_t burn, double *mean, double *var)
{
    if (!c || !mean || !var)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (burn < 0 || burn >= c->refresh_done)
        return fail(MC3HIP_ERR_ARG, "burn=%lld leaves no rows (%lld refreshes done)",
                    (long long)burn, (long long)c->refresh_done);
    return chain_moments(c, c->refresh_done, burn, mean, var);
}

int mc3hip_chain_moments_rows(mc3hip_ctx *c, int64_t nrefresh, int64_t burn, double *mean,
                              double *var)
{
    if (!c || !mean || !var)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (burn < 0 || burn >= nrefresh || c->P.M0 + nrefresh * c->P.nchains > c->zlen)
        return fail(MC3HIP_ERR_ARG, "burn=%lld of %lld refreshes (Z holds %lld rows)",
                    (long long)burn, (long long)nrefresh, (long long)c->zlen);
    return chain_moments(c, nrefresh, burn, mean, var);
}

int mc3hip_posterior_summary(mc3hip_ctx *c, int64_t nrefresh, int64_t burn, int32_t nranks,
                             const int64_t *ranks, double *order_stats, double *mean,
                             double *var, double *vmin, double *vmax)
{
    if (!c || !mean || !var || !vmin || !vmax || (nranks > 0 && (!ranks || !order_stats)))
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (nranks < 0 || nranks > SEL_MAXR)
        return fail(MC3HIP_ERR_ARG, "at most %d ranks per call", SEL_MAXR);
    if (nrefresh <= 0)
        nrefresh = c->refresh_done;
    if (burn < 0 || burn >= nrefresh || c->P.M0 + nrefresh * c->P.nchains > c->zlen)
        return fail(MC3HIP_ERR_ARG, "burn=%lld of %lld refreshes (Z holds %lld rows)",
                    (long long)burn, (long long)nrefresh, (long long)c->zlen);
    RC_TRY(settle(c));
    const DevProblem &P = c->P;
    const long long n = (long long)(nrefresh - burn) * P.nchains;
    const long long row0 = P.M0 + burn * P.nchains;
    for (int r = 0; r < nranks; ++r)
        if (ranks[r] < 0 || ranks[r] >= n)
            return fail(MC3HIP_ERR_ARG, "rank %lld outside [0,%lld)", (long long)ranks[r], n);
    const int nf = P.nfree;
    const int nblocks = (int)(n / 256 + 1 < 2048 ? n / 256 + 1 : 2048);
    // scratch of this call: keys [nf][n], block partials, results, selection state
    const size_t kbytes = (size_t)nf * (size_t)n * sizeof(uint64_t);
    const size_t pbytes = (size_t)nf * nblocks * 4 * sizeof(double);
    const size_t rbytes = (size_t)nf * 4 * sizeof(double);
    const size_t sbytes = (size_t)nf * sizeof(SelectState);
    const size_t hbytes = (size_t)nf * SEL_MAXR * 256 * sizeof(unsigned int);
    const size_t obytes = (size_t)nf * SEL_MAXR * sizeof(double);
    void *buf = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    RC_TRY(check_device_room(kbytes + pbytes + rbytes + sbytes + hbytes + obytes + 1024,
                             "the posterior summary's selection keys (8 bytes per row and free "
                             "parameter)"));
    HIP_TRY(hipMalloc(&buf, kbytes + pbytes + rbytes + sbytes + hbytes + obytes + 1024));
    struct Free { void *p; ~Free() { (void)hipFree(p); } } guard{buf};
    char *at = static_cast<char *>(buf);
    uint64_t *keys = reinterpret_cast<uint64_t *>(at); at += (kbytes + 255) & ~size_t(255);
    double *part = reinterpret_cast<double *>(at); at += (pbytes + 255) & ~size_t(255);
    double *res = reinterpret_cast<double *>(at); at += (rbytes + 255) & ~size_t(255);
    SelectState *st = reinterpret_cast<SelectState *>(at); at += (sbytes + 255) & ~size_t(255);
    unsigned int *hist = reinterpret_cast<unsigned int *>(at); at += (hbytes + 255) & ~size_t(255);
    double *ost = reinterpret_cast<double *>(at);
    MC3_LAUNCH(summary_pass1_kernel, dim3(nblocks), dim3(256), 0, c->stream,
                       (const double *)P.Z, row0, n, nf, P.zs, keys, part);
    MC3_LAUNCH(summary_finish1_kernel, dim3(1), dim3(256), 0, c->stream,
                       (const double *)part, nblocks, nf, n, res);
    MC3_LAUNCH(summary_pass2_kernel, dim3(nblocks), dim3(256), 0, c->stream,
                       (const double *)P.Z, row0, n, nf, P.zs, (const double *)res, part);
    MC3_LAUNCH(summary_finish2_kernel, dim3(1), dim3(256), 0, c->stream,
                       (const double *)part, nblocks, nf, n, res);
    if (nranks > 0) {
        std::vector<SelectState> h((size_t)nf);
        for (int k = 0; k < nf; ++k)
            for (int r = 0; r < SEL_MAXR; ++r) {
                h[k].prefix[r] = 0;
                h[k].rank[r] = r < nranks ? ranks[r] : 0;
            }
        HIP_TRY(hipMemcpyAsync(st, h.data(), sbytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(hist, 0, hbytes, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));          // (h goes out of scope below)
        for (int pass = 0; pass < 8; ++pass) {
            MC3_LAUNCH(select_hist_kernel, dim3(nblocks, nf), dim3(256), 0, c->stream,
                               (const uint64_t *)keys, n, nranks, pass,
                               (const SelectState *)st, hist);
            MC3_LAUNCH(select_pick_kernel, dim3(nf), dim3(256), 0, c->stream, nranks,
                               pass, st, hist);
        }
        MC3_LAUNCH(select_result_kernel, dim3(1), dim3(nf * SEL_MAXR), 0, c->stream,
                           (const SelectState *)st, nf, SEL_MAXR, ost);
    }
    HIP_TRY(hipGetLastError());
    std::vector<double> hres((size_t)nf * 4), host((size_t)nf * SEL_MAXR);
    HIP_TRY(hipMemcpyAsync(hres.data(), res, rbytes, hipMemcpyDeviceToHost, c->stream));
    if (nranks > 0)
        HIP_TRY(hipMemcpyAsync(host.data(), ost, obytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < nf; ++k) {
        mean[k] = hres[4 * k];
        var[k] = hres[4 * k + 1];
        vmin[k] = hres[4 * k + 2];
        vmax[k] = hres[4 * k + 3];
        for (int r = 0; r < nranks; ++r)
            order_stats[(size_t)r * nf + k] = host[(size_t)k * SEL_MAXR + r];
    }
    return MC3HIP_OK;
}

int mc3hip_get_chisq(mc3hip_ctx *c, int64_t row0, int64_t nrows, double *chisq)
{
    if (!c || !chisq)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (row0 < 0 || nrows < 0 || row0 + nrows > c->zlen)
        return fail(MC3HIP_ERR_ARG, "rows [%lld,%lld) outside Z", (long long)row0,
                    (long long)(row0 + nrows));
    RC_TRY(settle(c));
    const size_t chunk = STAGE_MAX_DOUBLES;
    for (size_t r = 0; r < (size_t)nrows; r += chunk) {
        const size_t m = (size_t)nrows - r < chunk ? (size_t)nrows - r : chunk;
        RC_TRY(stage_reserve(c, m));
        const int blocks = (int)((m + 255) / 256 < 4096 ? (m + 255) / 256 : 4096);
        MC3_LAUNCH(rows_chisq_kernel, dim3(blocks), dim3(256), 0, c->stream, c->P,
                           (long long)(row0 + r), (long long)m, c->stage);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(chisq + r, c->stage, m * sizeof(double), hipMemcpyDeviceToHost,
                               c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return MC3HIP_OK;
}

static int eval_reserve(mc3hip_ctx *c, int n)
{
    if (n <= c->eval_cap)
        return MC3HIP_OK;
    // grow-only scratch; freed with the context
    const int cap = n < 1024 ? 1024 : n;
    double *p = nullptr, *o = nullptr;
    RC_TRY(dev_alloc(c, &p, (size_t)cap * (size_t)c->P.nparams));
    RC_TRY(dev_alloc(c, &o, (size_t)cap));
    c->eval_params = p;
    c->eval_out = o;
    c->eval_cap = cap;
    return MC3HIP_OK;
}

int mc3hip_eval_chisq(mc3hip_ctx *c, const double *params, int32_t n, double *chisq_out)
{
    if (!c || !params || !chisq_out)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (n < 0)
        return fail(MC3HIP_ERR_ARG, "negative count");
    if (n == 0)
        return MC3HIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t np = (size_t)c->P.nparams;
    RC_TRY(eval_reserve(c, n));
    HIP_TRY(hipMemcpyAsync(c->eval_params, params, (size_t)n * np * sizeof(double),
                           hipMemcpyHostToDevice, c->stream));
    launch_eval(c, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(chisq_out, c->eval_out, (size_t)n * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MC3HIP_OK;
}

int mc3hip_init_population(mc3hip_ctx *c, int32_t kickoff, uint64_t seed, double *Z0,
                           double *log_post0, int64_t *filled_out, int64_t *failed_out)
{
    if (!c || !Z0 || !log_post0 || !filled_out || !failed_out)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (kickoff != 0 && kickoff != 1)
        return fail(MC3HIP_ERR_ARG, "kickoff must be 0 (normal) or 1 (uniform)");
    HIP_TRY(hipSetDevice(c->device));
    const DevProblem &P = c->P;
    const size_t nf = (size_t)P.nfree, M0 = (size_t)P.M0, ns = (size_t)P.nshare;
    // rounds of trials: generate + chisq in parallel, accept in order
    const int block = (int)(2 * M0 < 4096 ? 4096 : (2 * M0 > (1u << 20) ? (1u << 20) : 2 * M0));
    RC_TRY(eval_reserve(c, block));
    double *freep = nullptr, *dZ0 = nullptr, *dlp0 = nullptr, *stale = nullptr;
    int32_t *inside = nullptr;
    long long *st = nullptr;
    // (scratch of this call: one allocation, released on every way out)
    const size_t bytes = ((size_t)block * nf + M0 * nf + M0 + ns + 8) * sizeof(double) +
                         (size_t)block * sizeof(int32_t) + 64;
    void *buf = nullptr;
    HIP_TRY(hipMalloc(&buf, bytes));
    struct Free { void *p; ~Free() { (void)hipFree(p); } } guard{buf};
    freep = static_cast<double *>(buf);
    dZ0 = freep + (size_t)block * nf;
    dlp0 = dZ0 + M0 * nf;
    stale = dlp0 + M0;
    st = reinterpret_cast<long long *>(stale + ns + 1);
    inside = reinterpret_cast<int32_t *>(st + 4);
    {
        // the shared parameters' copies start as given (mcmc_driver.py:242: values = params)
        std::vector<double> h(ns + 1, 0.0);
        std::vector<double> hp((size_t)P.nparams);
        HIP_TRY(hipMemcpy(hp.data(), P.params, hp.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<int32_t> hs(ns ? ns : 1);
        if (ns)
            HIP_TRY(hipMemcpy(hs.data(), P.ishare, ns * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t q = 0; q < ns; ++q)
            h[q] = hp[(size_t)hs[q]];
        HIP_TRY(hipMemcpy(stale, h.data(), (ns + 1) * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemsetAsync(st, 0, 4 * sizeof(long long), c->stream));
    HIP_TRY(hipMemsetAsync(dZ0, 0, (M0 * nf + M0) * sizeof(double), c->stream));
    long long host_st[4] = {0, 0, 0, 0};
    long long t0 = 0;
    while ((size_t)host_st[0] < M0 && host_st[1] < 100 * (long long)M0) {
        MC3_LAUNCH(init_trials_kernel, dim3((block + 127) / 128), dim3(128), 0, c->stream,
                           P, kickoff, seed, t0, block, c->eval_params, freep, inside);
        launch_eval(c, block);
        MC3_LAUNCH(init_scan_kernel, dim3(1), dim3(64), 0, c->stream, P, block,
                           (const double *)c->eval_params, (const double *)freep,
                           (const int32_t *)inside, (const double *)c->eval_out, st, stale, dZ0,
                           dlp0);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host_st, st, sizeof host_st, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        t0 += block;
    }
    HIP_TRY(hipMemcpy(Z0, dZ0, M0 * nf * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(log_post0, dlp0, M0 * sizeof(double), hipMemcpyDeviceToHost));
    *filled_out = host_st[0];
    *failed_out = host_st[1];
    return MC3HIP_OK;
}

int mc3hip_eval_model(mc3hip_ctx *c, const double *params, double *model_out)
{
    if (!c || !params || !model_out)
        return fail(MC3HIP_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->eval_cap < 1) {
        double *p = nullptr, *o = nullptr;
        int rc = dev_alloc(c, &p, (size_t)1024 * c->P.nparams);
        if (rc)
            return rc;
        rc = dev_alloc(c, &o, (size_t)1024);
        if (rc)
            return rc;
        c->eval_params = p;
        c->eval_out = o;
        c->eval_cap = 1024;
    }
    HIP_TRY(hipMemcpyAsync(c->eval_params, params, (size_t)c->P.nparams * sizeof(double),
                           hipMemcpyHostToDevice, c->stream));
    launch_model(c);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(model_out, c->model_out, (size_t)c->P.ndata * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MC3HIP_OK;
}

int mc3hip_dwt_daub4(int32_t device, const double *in, int32_t n, int32_t inverse, double *out)
{
    if (!in || !out || n < 1)
        return fail(MC3HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    int nw = 1;
    while (nw < n)
        nw <<= 1;
    double *a = nullptr, *tmp = nullptr;
    HIP_TRY(hipMalloc(&a, (size_t)nw * sizeof(double)));
    if (hipMalloc(&tmp, (size_t)nw * sizeof(double)) != hipSuccess) {
        (void)hipFree(a);
        return fail(MC3HIP_ERR_HIP, "hipMalloc failed");
    }
    int rc = MC3HIP_OK;
    hipError_t e = hipMemset(a, 0, (size_t)nw * sizeof(double));
    if (e == hipSuccess)
        e = hipMemcpy(a, in, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        MC3_LAUNCH(daub4_kernel, dim3(1), dim3(1024), 0, 0, a, tmp, nw, inverse ? 1 : 0);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(out, a, (size_t)nw * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        rc = fail(MC3HIP_ERR_HIP, "dwt_daub4: %s", hipGetErrorString(e));
    (void)hipFree(a);
    (void)hipFree(tmp);
    return rc;
}

namespace {

// Device buffers of one standalone call, carved out of a per-device arena that is kept
// between calls (grow-only up to 256 MiB; a larger one is released when the call ends), so
// that mc3hip_chisq / residuals / dwt_chisq on small arrays pay for their copies and
// launches, not for allocations.  Calls on one device serialise on the arena's lock.
struct Arena {
    std::mutex lock;
    void *base = nullptr;
    size_t bytes = 0;
};
Arena g_arena[64];

struct Scratch {
    Arena *a = nullptr;
    size_t used = 0;
    // total_doubles: an upper bound of everything up() will be asked for
    Scratch(int device, size_t total_doubles, int nbuffers, hipError_t *e)
    {
        a = &g_arena[(device >= 0 && device < 64) ? device : 0];
        a->lock.lock();
        const size_t need = (total_doubles + 32 * (size_t)nbuffers + 32) * sizeof(double);
        if (need > a->bytes) {
            if (a->base)
                (void)hipFree(a->base);
            a->base = nullptr;
            a->bytes = 0;
            *e = hipMalloc(&a->base, need);
            if (*e == hipSuccess)
                a->bytes = need;
        }
    }
    ~Scratch()
    {
        if (a->bytes > (size_t(256) << 20)) {
            (void)hipFree(a->base);
            a->base = nullptr;
            a->bytes = 0;
        }
        a->lock.unlock();
    }
    double *up(const double *src, size_t n, hipError_t *e)
    {
        if (*e != hipSuccess)
            return nullptr;
        const size_t bytes = ((n ? n : 1) * sizeof(double) + 255) & ~size_t(255);
        if (used + bytes > a->bytes) {
            *e = hipErrorOutOfMemory;
            return nullptr;
        }
        double *p = reinterpret_cast<double *>(static_cast<char *>(a->base) + used);
        used += bytes;
        if (src && n)
            *e = hipMemcpy(p, src, n * sizeof(double), hipMemcpyHostToDevice);
        return p;
    }
};

// prior term on the host exactly as src_c/include/stats.h:96-106 (np is tiny)
static double host_priors(const double *off, const double *plow, const double *pup, int np)
{
    double pr = 0.0;
    for (int j = 0; j < np; ++j) {
        const double s = (off[j] > 0.0) ? off[j] / pup[j] : off[j] / plow[j];
        pr = pr + s * s;
    }
    return pr;
}

}  // namespace

int mc3hip_residuals(int32_t device, const double *model, const double *data,
                     const double *uncert, int32_t n, const double *off, const double *plow,
                     const double *pup, int32_t np, double *out)
{
    if (!model || !data || !uncert || !out || n < 0 || np < 0 || (np > 0 && (!off || !plow || !pup)))
        return fail(MC3HIP_ERR_ARG, "bad argument");
    if (n + np == 0)
        return MC3HIP_OK;
    HIP_TRY(hipSetDevice(device));
    hipError_t e = hipSuccess;
    Scratch S(device, 4 * ((size_t)n + np), 7, &e);
    double *dm = S.up(model, n, &e), *dd = S.up(data, n, &e), *du = S.up(uncert, n, &e);
    double *dof = S.up(off, np, &e), *dlo = S.up(plow, np, &e), *dup = S.up(pup, np, &e);
    double *dout = S.up(nullptr, (size_t)n + np, &e);
    if (e == hipSuccess) {
        const int nb = (n + np + 255) / 256;
        MC3_LAUNCH(residuals_kernel, dim3(nb > 1024 ? 1024 : nb), dim3(256), 0, 0,
                           (const double *)dm, (const double *)dd, (const double *)du, n,
                           (const double *)dof, (const double *)dlo, (const double *)dup, np, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(out, dout, ((size_t)n + np) * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail(MC3HIP_ERR_HIP, "residuals: %s", hipGetErrorString(e));
    return MC3HIP_OK;
}

int mc3hip_chisq(int32_t device, const double *model, const double *data, const double *uncert,
                 int32_t n, const double *off, const double *plow, const double *pup, int32_t np,
                 double *chisq_out)
{
    if (!model || !data || !uncert || !chisq_out || n < 1 || np < 0 ||
        (np > 0 && (!off || !plow || !pup)))
        return fail(MC3HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    hipError_t e = hipSuccess;
    Scratch S(device, 4 * (size_t)n + 1, 5, &e);
    double *dm = S.up(model, n, &e), *dd = S.up(data, n, &e), *du = S.up(uncert, n, &e);
    double *dres = S.up(nullptr, n, &e), *dout = S.up(nullptr, 1, &e);
    double total = 0.0;
    if (e == hipSuccess) {
        const int nb = (n + 255) / 256;
        MC3_LAUNCH(residuals_kernel, dim3(nb > 1024 ? 1024 : nb), dim3(256), 0, 0,
                           (const double *)dm, (const double *)dd, (const double *)du, n,
                           (const double *)nullptr, (const double *)nullptr,
                           (const double *)nullptr, 0, dres);
        MC3_LAUNCH(sumsq_kernel, dim3(1), dim3(1024), 0, 0, (const double *)dres, n,
                           (const double *)nullptr, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(&total, dout, sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail(MC3HIP_ERR_HIP, "chisq: %s", hipGetErrorString(e));
    *chisq_out = total + (np > 0 ? host_priors(off, plow, pup, np) : 0.0);
    return MC3HIP_OK;
}

int mc3hip_dwt_chisq(int32_t device, const double *params, int32_t nparams, const double *model,
                     const double *data, int32_t n, const double *off, const double *plow,
                     const double *pup, int32_t np, double *chisq_out)
{
    if (!params || !model || !data || !chisq_out || np < 0 || (np > 0 && (!off || !plow || !pup)))
        return fail(MC3HIP_ERR_ARG, "bad argument");
    if (nparams < 3)
        return fail(MC3HIP_ERR_ARG, "Wavelet chisq should have at least three parameters");
    if (n < 2 || (n & (n - 1)) || n > (1 << 24))
        return fail(MC3HIP_ERR_ARG, "wavelet likelihood needs power-of-two ndata (got %d)", n);
    HIP_TRY(hipSetDevice(device));
    const double gam = params[nparams - 3], sr = params[nparams - 2], sw = params[nparams - 1];
    int M = 0;
    while ((1 << M) < n)
        ++M;
    // per-scale weights, same arithmetic as the in-kernel table (block_wavelet)
    std::vector<double> inv(32, 0.0);
    const double sr2 = sr * sr, sw2 = sw * sw;
    double Lg = 0.0;
    const double sS2 = sr2 * mc3_det_exp2(-gam) * 0.72134752 + sw2;
    Lg = 2.0 * mc3_det_log(6.283185307179586 * sS2);
    for (int m = 1; m < M; ++m) {
        const double sW2 = sr2 * mc3_det_exp2(-gam * (double)m) + sw2;
        inv[m] = 1.0 / sW2;
        Lg = Lg + (double)(1 << m) * mc3_det_log(6.283185307179586 * sW2);
    }
    hipError_t e = hipSuccess;
    Scratch S(device, 4 * (size_t)n + 33, 6, &e);
    double *dm = S.up(model, n, &e), *dd = S.up(data, n, &e), *dinv = S.up(inv.data(), 32, &e);
    double *dres = S.up(nullptr, n, &e), *dtmp = S.up(nullptr, n, &e), *dout = S.up(nullptr, 1, &e);
    double T = 0.0, head[2] = {0.0, 0.0};
    if (e == hipSuccess) {
        const int nb = (n + 255) / 256;
        MC3_LAUNCH(diff_kernel, dim3(nb > 1024 ? 1024 : nb), dim3(256), 0, 0,
                           (const double *)dd, (const double *)dm, n, n, dres);   // data - model
        MC3_LAUNCH(daub4_kernel, dim3(1), dim3(1024), 0, 0, dres, dtmp, n, 0);
        MC3_LAUNCH(sumsq_kernel, dim3(1), dim3(1024), 0, 0, (const double *)dres, n,
                           (const double *)dinv, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(&T, dout, sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(head, dres, 2 * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail(MC3HIP_ERR_HIP, "dwt_chisq: %s", hipGetErrorString(e));
    const double Ssm = (head[0] * head[0]) / sS2 + (head[1] * head[1]) / sS2;
    *chisq_out = ((Ssm + Lg) + T) + (np > 0 ? host_priors(off, plow, pup, np) : 0.0);
    return MC3HIP_OK;
}

int mc3hip_kde_pdf(int32_t device, const double *sample, int32_t n, const double *x, int32_t m,
                   double cov, double *pdf)
{
    if (!sample || !x || !pdf || n < 1 || m < 1 || !(cov > 0.0))
        return fail(MC3HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    hipError_t e = hipSuccess;
    Scratch S(device, (size_t)n + 2 * (size_t)m, 3, &e);
    double *ds = S.up(sample, (size_t)n, &e), *dx = S.up(x, (size_t)m, &e);
    double *dp = S.up(nullptr, (size_t)m, &e);
    if (e == hipSuccess) {
        // scipy.stats.gaussian_kde: sum_i exp(-(x - s_i)^2 / (2 cov)) / (n sqrt(2 pi cov))
        const double norm = 1.0 / ((double)n * std::sqrt(2.0 * 3.14159265358979323846 * cov));
        MC3_LAUNCH(kde_kernel, dim3(m), dim3(256), 0, 0, (const double *)ds, n,
                           (const double *)dx, 0.5 / cov, norm, dp);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(pdf, dp, (size_t)m * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail(MC3HIP_ERR_HIP, "kde_pdf: %s", hipGetErrorString(e));
    return MC3HIP_OK;
}

int mc3hip_comm_unique_id(void *unique_id_128)
{
    if (!unique_id_128)
        return fail(MC3HIP_ERR_ARG, "null argument");
    int rc = load_rccl();
    if (rc)
        return rc;
    ncclUniqueId id;
    rc = g_rccl.GetUniqueId(&id);
    if (rc != 0)
        return fail(MC3HIP_ERR_COMM, "ncclGetUniqueId failed (%d)", rc);
    memcpy(unique_id_128, &id, sizeof id);
    return MC3HIP_OK;
}

int mc3hip_comm_init(mc3hip_ctx *c, const void *unique_id_128, int32_t rank, int32_t nranks)
{
    if (!c || !unique_id_128)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks)
        return fail(MC3HIP_ERR_ARG, "bad rank %d of %d", rank, nranks);
    // the in-place all-gather needs equal contiguous blocks in rank order
    if (c->P.nlocal * nranks != c->P.nchains || c->P.chain0 != rank * c->P.nlocal)
        return fail(MC3HIP_ERR_ARG,
                    "rank %d must own chains [%d,%d) of %d (equal contiguous blocks)", rank,
                    rank * c->P.nlocal, (rank + 1) * c->P.nlocal, c->P.nchains);
    int rc = load_rccl();
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, unique_id_128, sizeof id);
    rc = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (rc != 0) {
        c->comm = nullptr;
        return fail(MC3HIP_ERR_COMM, "ncclCommInitRank failed: %s",
                    g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    }
    c->rank = rank;
    c->nranks = nranks;
    return MC3HIP_OK;
}

int mc3hip_p2p_export(mc3hip_ctx *c, void *handles_128)
{
    if (!c || !handles_128)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (!(c->flags & MC3HIP_FLAG_P2P))
        return fail(MC3HIP_ERR_STATE, "context was not created with MC3HIP_FLAG_P2P");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    HIP_TRY(hipSetDevice(c->device));
    hipIpcMemHandle_t h[2];
    HIP_TRY(hipIpcGetMemHandle(&h[0], c->P.Z));
    HIP_TRY(hipIpcGetMemHandle(&h[1], c->P.log_post));
    memcpy(handles_128, h, sizeof h);
    return MC3HIP_OK;
}

int mc3hip_p2p_attach(mc3hip_ctx *c, int32_t rank, int32_t nranks, const void *all_handles)
{
    if (!c || !all_handles)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (!(c->flags & MC3HIP_FLAG_P2P))
        return fail(MC3HIP_ERR_STATE, "context was not created with MC3HIP_FLAG_P2P");
    if (nranks < 2 || nranks > 64 || rank < 0 || rank >= nranks)
        return fail(MC3HIP_ERR_ARG, "bad rank %d of %d", rank, nranks);
    if (c->P.nlocal * nranks != c->P.nchains || c->P.chain0 != rank * c->P.nlocal)
        return fail(MC3HIP_ERR_ARG,
                    "rank %d must own chains [%d,%d) of %d (equal contiguous blocks)", rank,
                    rank * c->P.nlocal, (rank + 1) * c->P.nlocal, c->P.nchains);
    if (c->comm)
        return fail(MC3HIP_ERR_STATE, "an RCCL communicator is already attached");
    if ((c->P.tile > 0 && !resident_shape(c)) || decide_persist(c) <= 0)
        return fail(MC3HIP_ERR_STATE,
                    "the peer-to-peer hand-off needs the persistent kernel, which this "
                    "problem cannot use (tiled plan, or too many chains to be co-resident)");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<double *> pz(nranks), pl(nranks);
    const hipIpcMemHandle_t *h = static_cast<const hipIpcMemHandle_t *>(all_handles);
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) {
            pz[r] = c->P.Z;
            pl[r] = c->P.log_post;
            continue;
        }
        void *z = nullptr, *l = nullptr;
        const bool dbg = dev_env_text("MC3HIP_DEBUG") != nullptr;
        if (dbg)
            fprintf(stderr, "[mc3hip rank %d] opening Z of rank %d\n", rank, r);
        HIP_TRY(hipIpcOpenMemHandle(&z, h[2 * r], hipIpcMemLazyEnablePeerAccess));
        c->p2p_opened.push_back(z);
        if (dbg)
            fprintf(stderr, "[mc3hip rank %d] opening log_post of rank %d\n", rank, r);
        HIP_TRY(hipIpcOpenMemHandle(&l, h[2 * r + 1], hipIpcMemLazyEnablePeerAccess));
        c->p2p_opened.push_back(l);
        if (dbg)
            fprintf(stderr, "[mc3hip rank %d] opened rank %d\n", rank, r);
        pz[r] = static_cast<double *>(z);
        pl[r] = static_cast<double *>(l);
    }
    double **dz = nullptr, **dl = nullptr;
    int rc = dev_alloc(c, &dz, (size_t)nranks);
    if (rc)
        return rc;
    rc = dev_alloc(c, &dl, (size_t)nranks);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(dz, pz.data(), nranks * sizeof(double *), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dl, pl.data(), nranks * sizeof(double *), hipMemcpyHostToDevice));
    c->P.peerZ = dz;
    c->P.peerLP = dl;
    c->P.npeers = nranks;
    c->P.rank = rank;
    c->p2p_ranks = nranks;
    // the grid must be co-resident for the instance that stores to the peers
    c->persist_checked = false;
    if (decide_persist(c) <= 0) {
        (void)mc3hip_p2p_detach(c);
        return fail(MC3HIP_ERR_STATE,
                    "the peer-to-peer instance of the persistent kernel cannot keep %d "
                    "workgroups resident", c->P.nlocal);
    }
    return MC3HIP_OK;
}

int mc3hip_comm_gather_log_post(mc3hip_ctx *c)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    if (!c->comm)
        return MC3HIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    // log_post rows are not needed by proposals, so they are exchanged lazily, and in bulk:
    // this rank's entries of all not-yet-exchanged refreshes are packed into one dense
    // buffer, ONE all-gather moves everybody's, a kernel scatters them (chunked by the
    // staging buffer: 64 MiB hold half a million refreshes of C3)
    const DevProblem &P = c->P;
    const size_t per_ref = (size_t)P.nlocal + (size_t)P.nchains;
    const int64_t chunk = (int64_t)(STAGE_MAX_DOUBLES / per_ref ? STAGE_MAX_DOUBLES / per_ref : 1);
    for (int64_t R = c->logpost_gathered; R < c->refresh_done; R += chunk) {
        const int64_t n = c->refresh_done - R < chunk ? c->refresh_done - R : chunk;
        RC_TRY(stage_reserve(c, (size_t)n * per_ref));
        double *send = c->stage, *recv = c->stage + (size_t)n * P.nlocal;
        const int nb = (int)(((size_t)n * P.nchains + 255) / 256 < 2048
                                 ? ((size_t)n * P.nchains + 255) / 256 : 2048);
        MC3_LAUNCH(logpost_pack_kernel, dim3(nb), dim3(256), 0, c->stream, P,
                           (long long)R, (long long)n, send);
        int rc = g_rccl.AllGather(send, recv, (size_t)n * P.nlocal, NCCL_FLOAT64, c->comm,
                                  c->stream);
        if (rc != 0)
            return fail(MC3HIP_ERR_COMM, "ncclAllGather(log_post) failed (%d)", rc);
        MC3_LAUNCH(logpost_unpack_kernel, dim3(nb), dim3(256), 0, c->stream, P,
                           (long long)R, (long long)n, (const double *)recv);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));      // the staging buffer is reused
    }
    c->logpost_gathered = c->refresh_done;
    return MC3HIP_OK;
}

int mc3hip_comm_info(mc3hip_ctx *c, int32_t *nranks, int32_t *rank)
{
    if (!c || !nranks || !rank)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (!c->comm)
        return fail(MC3HIP_ERR_STATE, "no RCCL communicator is attached");
    if (!g_rccl.CommCount || !g_rccl.CommUserRank)
        return fail(MC3HIP_ERR_COMM, "librccl.so lacks ncclCommCount / ncclCommUserRank");
    int n = 0, r = 0;
    if (g_rccl.CommCount(c->comm, &n) != 0 || g_rccl.CommUserRank(c->comm, &r) != 0)
        return fail(MC3HIP_ERR_COMM, "ncclCommCount / ncclCommUserRank failed");
    *nranks = n;
    *rank = r;
    return MC3HIP_OK;
}

int mc3hip_device_memory(int32_t device, int64_t *free_bytes, int64_t *total_bytes)
{
    if (!free_bytes || !total_bytes)
        return fail(MC3HIP_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f;
    *total_bytes = (int64_t)t;
    return MC3HIP_OK;
}

int mc3hip_comm_destroy(mc3hip_ctx *c)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    if (!c->comm)
        return MC3HIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->cgraph_exec)
        (void)hipGraphExecDestroy(c->cgraph_exec);
    if (c->cgraph)
        (void)hipGraphDestroy(c->cgraph);
    c->cgraph_exec = nullptr;
    c->cgraph = nullptr;
    c->comm_graph_nodes = 0;
    if (g_rccl.CommDestroy)
        g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
    c->comm_warm = false;
    c->nranks = 1;
    return MC3HIP_OK;
}

int mc3hip_p2p_detach(mc3hip_ctx *c)
{
    if (!c)
        return fail(MC3HIP_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (void *p : c->p2p_opened)
        (void)hipIpcCloseMemHandle(p);
    c->p2p_opened.clear();
    c->p2p_ranks = 0;
    c->P.npeers = 0;
    c->P.peerZ = nullptr;
    c->P.peerLP = nullptr;
    c->persist_checked = false;         // sized again for the instance that will run
    return MC3HIP_OK;
}

int mc3hip_p2p_rollback(mc3hip_ctx *c, int64_t *from_refresh, int64_t *to_refresh)
{
    if (!c || !from_refresh || !to_refresh)
        return fail(MC3HIP_ERR_ARG, "null argument");
    if (!(c->flags & MC3HIP_FLAG_P2P))
        return fail(MC3HIP_ERR_STATE, "context was not created with MC3HIP_FLAG_P2P");
    if (!c->ckpt_valid)
        return fail(MC3HIP_ERR_STATE, "no checkpoint is pending: nothing to roll back to");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const DevProblem &P = c->P;
    const size_t row0 = (size_t)(P.M0 + c->ckpt_R * P.nchains);
    const size_t nrows = (size_t)((c->refresh_done - c->ckpt_R) * P.nchains);
    RC_TRY(checkpoint_copy(c, false));
    // the caller guarantees that every rank's kernels have ended: nobody stores here any more
    HIP_TRY(hipMemsetAsync(P.Z + row0 * (size_t)P.zs, 0xff,
                           nrows * (size_t)P.zs * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(P.zchain + row0, 0xff, nrows * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemsetAsync(P.log_post + row0, 0, nrows * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->ctl + CTL_ABORT, 0, sizeof(long long), c->stream));
    if (P.rs_prop) {
        HIP_TRY(hipMemsetAsync(P.rs_prop, 0xff, 2 * (size_t)P.nlocal * RS_ROW * sizeof(double),
                               c->stream));
        HIP_TRY(hipMemsetAsync(P.rs_part, 0xff, (size_t)P.nlocal * P.rs_ntp * sizeof(double),
                               c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    *from_refresh = c->ckpt_R;
    *to_refresh = c->refresh_done;
    c->refresh_done = c->ckpt_R;
    c->ckpt_valid = false;
    c->ckpt_settled = false;
    c->recoveries += 1;
    // the peers stay mapped until the context goes, but are no longer stored to: from here
    // on the context runs one launch per refresh with an explicit exchange
    c->p2p_ranks = 0;
    c->P.npeers = 0;
    c->persist_refs = 0;
    c->persist_checked = true;
    c->fast = false;
    c->resident = false;
    return MC3HIP_OK;
}

/* diagnostic build only: the 16 phase stamps of the last launch (zeros otherwise) */
int mc3hip_debug_stamps(mc3hip_ctx *c, uint64_t *out16)
{
    if (!c || !out16)
        return fail(MC3HIP_ERR_ARG, "null argument");
    {
        const int rc_settle = settle(c);
        if (rc_settle)
            return rc_settle;
    }
    // (MC3HIP_STAMP_PAGE=1: the second group of sixteen, the resident tiled plan's phases;
    // pages 2..4095: a timeline build's records)
    const char *page = dev_env_text("MC3HIP_STAMP_PAGE");
    const int pg = (page && atoi(page) > 0 && atoi(page) < 4096) ? atoi(page) : 0;
    HIP_TRY(hipMemcpy(out16, c->P.stamps + 16 * (size_t)pg, 16 * sizeof(uint64_t),
                      hipMemcpyDeviceToHost));
    return MC3HIP_OK;
}

namespace {
// "<library offset of the handle> <device kernel name>" of an instance of this library: the
// offset is what `nm libmc3hip.so` lists the handle under (what a launch is given is the
// kernel's handle, a hidden data object of the kernel's name: dladdr knows the library's base,
// not the object's name), the name is the runtime's
static void ledger_label(const void *kernel, std::string *out)
{
    char buf[64];
    Dl_info info;
    unsigned long long off = 0;
    if (dladdr(kernel, &info) && info.dli_fbase)
        off = (unsigned long long)((const char *)kernel - (const char *)info.dli_fbase);
    snprintf(buf, sizeof buf, "0x%llx ", off);
    const char *name = hipKernelNameRefByPtr(kernel, nullptr);
    *out = std::string(buf) + (name ? name : "?");
}
}  // namespace

/* The launch ledger as text, one line per kernel instance this PROCESS has launched:
 * "<launches> <of which with no MC3HIP_* switch set> <offset of the kernel's handle in libmc3hip.so>
 * <device kernel name> [<switch seen> ...]" (run-time specialisations: offset 0x0, name
 * "jit:<kernel>").  Writes at most cap - 1 bytes and a
 * terminating 0; *needed = bytes the whole text takes (with the 0). */
int mc3hip_debug_launch_counts(char *buf, int64_t cap, int64_t *needed)
{
    if (!needed || (cap > 0 && !buf))
        return fail(MC3HIP_ERR_ARG, "null argument");
    std::string text;
    {
        std::lock_guard<std::mutex> hold(g_ledger_lock);
        for (const LedgerRow &r : g_ledger) {
            std::string label = "0x0 " + r.name;
            if (r.kernel)
                ledger_label(r.kernel, &label);
            text += std::to_string(r.launches) + " " + std::to_string(r.plain) + " " + label +
                    r.switches + "\n";
        }
    }
    *needed = (int64_t)text.size() + 1;
    if (cap > 0) {
        const size_t n = text.size() < (size_t)cap - 1 ? text.size() : (size_t)cap - 1;
        memcpy(buf, text.data(), n);
        buf[n] = 0;
    }
    return MC3HIP_OK;
}

/* The chain kernel this context launched last, as mc3hip_debug_launch_counts labels it
 * ("" before the first launch). */
int mc3hip_debug_last_kernel(mc3hip_ctx *c, char *buf, int64_t cap)
{
    if (!c || !buf || cap < 1)
        return fail(MC3HIP_ERR_ARG, "null argument");
    std::string text;
    if (c->last_kernel)
        ledger_label(c->last_kernel, &text);
    else if (c->jit_fn && !c->jit_name.empty() && c->refresh_done > 0)
        text = "0x0 " + c->jit_name;
    snprintf(buf, (size_t)cap, "%s", text.c_str());
    return MC3HIP_OK;
}

int mc3hip_selftest_math(int32_t device, int32_t op, const double *in, int32_t n, double *out)
{
    if (!in || !out || n < 1)
        return fail(MC3HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    hipError_t e = hipSuccess;
    Scratch S(device, 2 * (size_t)n, 2, &e);       // released on every way out
    double *di = S.up(in, (size_t)n, &e), *dout = S.up(nullptr, (size_t)n, &e);
    if (e == hipSuccess) {
        MC3_LAUNCH(selftest_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op,
                           (const double *)di, n, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail(MC3HIP_ERR_HIP, "selftest_math: %s", hipGetErrorString(e));
    return MC3HIP_OK;
}

int mc3hip_selftest_philox(int32_t device, const uint32_t *ctr, int32_t n, const uint32_t key[2],
                           uint32_t *out)
{
    if (!ctr || !out || !key || n < 1)
        return fail(MC3HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    hipError_t e = hipSuccess;
    Scratch S(device, 4 * (size_t)n, 2, &e);       // (16 bytes per counter = 2 doubles)
    uint32_t *dc = reinterpret_cast<uint32_t *>(
        S.up(reinterpret_cast<const double *>(ctr), (size_t)n * 2, &e));
    uint32_t *dout = reinterpret_cast<uint32_t *>(S.up(nullptr, (size_t)n * 2, &e));
    if (e == hipSuccess) {
        MC3_LAUNCH(selftest_philox_kernel, dim3((n + 255) / 256), dim3(256), 0, 0,
                           (const uint32_t *)dc, n, key[0], key[1], dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(out, dout, (size_t)n * 16, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return fail(MC3HIP_ERR_HIP, "selftest_philox: %s", hipGetErrorString(e));
    return MC3HIP_OK;
}

}  // extern "C"
#pragma GCC visibility pop
