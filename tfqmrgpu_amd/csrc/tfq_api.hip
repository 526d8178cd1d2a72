// C-ABI of libtfQMRgpu.so (include/tfqmrgpu.h, include/tfqmrgpu_ext.h) for MI355X.
// Mirrors the entry points of real-space/tfQMRgpu tfQMRgpu/source/tfqmrgpu.cu (same names,
// argument meaning and status codes); the implementation behind them is this library's own.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <new>
#include <vector>
#include <dlfcn.h>

#include "tfq_device.hpp"
#include "tfq_order.hpp"
#include "tfq_vec.hpp"
#include "tfq_switch.hpp"

using namespace tfq;

namespace tfq {

static inline char lower(char c) { return char(c | 32); }

DevPlan resolve(Plan const& p) {
    DevPlan d{};
    char* b = p.buffer;
    auto at = [&](Window const& w) { return (void*)(b + w.offset); };
    d.LM = p.LM; d.LN = p.LN; d.dbl = ('z' == p.precision);
    d.nCols = p.nCols; d.nnzbX = p.nnzbX; d.nnzbB = p.nnzbB; d.nnzbA = p.nnzbA;
    d.nChunks = uint32_t(p.chunks.col.size());
    static int const hashEnv = lab_switch("TFQMRGPU_HASHV3", 1);
    d.hashV3 = (p.v3IsHash && hashEnv) ? 1 : 0;   // (lab builds, TFQMRGPU_HASHV3=0: the multiply kernels read v3 also in hash mode)
    d.ilv = p.ilv;
    static int const antEnv = lab_switch("TFQMRGPU_A_STREAM", 1);
    d.aOnce = (p.aOnce && antEnv) ? 1 : 0;   // (lab builds, TFQMRGPU_A_STREAM=0: A operands always through the caches)
    d.m3 = p.threeProducts ? 1 : 0;
    d.fold = 0; d.foldCount = (uint32_t*)at(p.wFold); d.self = (DevPlan const*)at(p.wSelf);
    d.R = ('m' == p.precision) ? at(p.wR) : nullptr;
    d.x = at(p.wX); d.v4 = at(p.wV4); d.v5 = at(p.wV5); d.v6 = at(p.wV6); d.v7 = at(p.wV7);
    d.v8 = at(p.wV8); d.v9 = at(p.wV9); d.B = at(p.wB); d.A = at(p.wA); d.v3 = (float*)at(p.wV3);
    d.rho = at(p.wRho); d.alfa = at(p.wAlfa); d.beta = at(p.wBeta); d.c67 = at(p.wC67); d.eta = at(p.wEta);
    d.c67a = at(p.wC67a); d.eta2 = at(p.wEta2);
    d.z = (double*)at(p.wZ); d.d = (double*)at(p.wD); d.tau = (double*)at(p.wTau); d.var = (double*)at(p.wVar);
    d.invBn2 = (double*)at(p.wInvBn2); d.status = (int8_t*)at(p.wStatus); d.ctl = (Ctl*)at(p.wCtl);
    d.pz = (double*)at(p.wPz); d.pd = (double*)at(p.wPd); d.colrec = (double*)at(p.wColRec);
    d.colPart = (double*)at(p.wColPart); d.colSegMax = p.colSegMax;
    d.chunkFirst = (uint32_t*)at(p.wChunkFirst); d.chunkCol = (uint32_t*)at(p.wChunkCol);
    d.colChunkPtr = (uint32_t*)at(p.wColChunkPtr); d.colStart = (uint32_t*)at(p.wColStart);
    d.order = (uint32_t*)at(p.wOrder); d.bOfX = (uint32_t*)at(p.wBofX); d.starts = (uint32_t*)at(p.wStarts); d.pairs = (uint32_t*)at(p.wPairs);
    d.subset = (uint32_t*)at(p.wSubset); d.bColPtr = (uint32_t*)at(p.wBColPtr); d.bList = (uint32_t*)at(p.wBList);
    d.u2i = (uint32_t*)at(p.wU2I); d.rowI = (uint32_t*)at(p.wRowI); d.origCol = (int32_t*)at(p.wOrigCol);
    d.colBatch = p.colBatch.empty() ? nullptr : (uint8_t const*)at(p.wColBatch);
    d.orderB = p.colBatch.empty() ? nullptr : (uint32_t const*)at(p.wOrderB); d.nChunksB = uint32_t(p.chunks.orderB.size());
    return d;
}

// mixed precision: the double-precision side of the plan as a plan of its own (x, B, A in double, element order ilvZ, the chunk
// tables of the float plan) -- what the layout conversions and the refinement's multiply work on
static DevPlan resolveZ(Plan const& p) {
    DevPlan d = resolve(p);
    char* b = p.buffer;
    d.dbl = true; d.ilv = p.ilvZ; d.hashV3 = 0; d.R = nullptr;
    d.x = b + p.wXz.offset; d.B = b + p.wBz.offset; d.A = b + p.wAz.offset;
    return d;
}

// glibc rand() (TYPE_3 additive feedback generator, seed 1) restated so that the shadow vector of
// the reference CPU path (tfqmrgpu_linalg.hxx:799-802) can be reproduced in any process state
struct GlibcRand {
    uint32_t ring[31];
    int f = 3, b = 0;      // r[i] = r[i-3] + r[i-31]
    explicit GlibcRand(uint32_t seed = 1) {
        int32_t word = int32_t(seed);
        ring[0] = uint32_t(word);
        for (int i = 1; i < 31; ++i) {          // srandom_r: Park-Miller steps fill the state
            int32_t const hi = word / 127773, lo = word % 127773;
            word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            ring[i] = uint32_t(word);
        }
        for (int i = 0; i < 310; ++i) (void)next();   // srandom_r discards 10*31 outputs
    }
    int32_t next() {
        ring[f] += ring[b];
        uint32_t const out = ring[f] >> 1;
        f = (f + 1) % 31; b = (b + 1) % 31;
        return int32_t(out);
    }
};

static tfqmrgpuStatus_t hipCheck(hipError_t e, int code, int line) {
    return (hipSuccess == e) ? TFQMRGPU_STATUS_SUCCESS : err(code, line % 10000);
}
#define TFQ_HIP(call, code) { auto const st_ = hipCheck((call), (code), __LINE__); if (st_) return st_; }

// staging area for raw user blocks: the work vectors v4..v9 (free outside of solve)
struct Stage { char* ptr; size_t bytes; };
static Stage stage_of(Plan const& p) {
    return { p.buffer + p.wV4.offset, (p.wV9.offset + p.wV9.bytes) - p.wV4.offset };
}

// move blocks between a host array in the caller's layout and a native device array.  userDbl: precision of the caller's array,
// nativeDbl / ilv: precision and element order of the library's; `also`: a second library-side copy of the same blocks (the float A of a
// mixed-precision plan next to its double A)
struct Target { void* native; bool dbl; int ilv; };

// is `ptr` memory the GPU can read and write directly (hipMalloc, hipMallocManaged)?  Then setMatrix / getMatrix convert in place
// of the caller's array, no staging and no PCIe: the "same A, new B, solve again" loop of a caller whose B and X live on the
// device (README.md:97-104 of the reference asks for plan reuse; with host arrays X alone is 0.1 s of copies per 35 ms solve on P2).
static bool on_device(void const* ptr) {
    hipPointerAttribute_t a{};
    if (hipSuccess != hipPointerGetAttributes(&a, ptr)) { (void)hipGetLastError(); return false; }   // plain host memory: not an error of the caller
    return hipMemoryTypeDevice == a.type || hipMemoryTypeManaged == a.type || a.isManaged;
}

static tfqmrgpuStatus_t transfer_blocks(Plan& p, hipStream_t s, int direction, bool userDbl, Target const& to,
    void* host, uint32_t const* u2n, uint32_t nBlocks, int nR, int nC, int layout, bool trans, bool conj, Stage const* own = nullptr,
    Target const* also = nullptr)
{
    if (on_device(host)) {   // the caller's array is device memory: one conversion kernel straight from / into it, asynchronous on the stream
        launch_convert(direction, userDbl, to.dbl, to.native, host, u2n, 0, nBlocks, nR, nC, layout, trans, conj, to.ilv, s);
        if (also && 0 == direction) launch_convert(0, userDbl, also->dbl, also->native, host, u2n, 0, nBlocks, nR, nC, layout, trans, conj, also->ilv, s);
        TFQ_HIP(hipGetLastError(), TFQMRGPU_STATUS_LAUNCH_FAILED)
        return TFQMRGPU_STATUS_SUCCESS;
    }
    Stage const st = own ? *own : stage_of(p);
    size_t const blockBytes = size_t(2) * nR * nC * (userDbl ? 8 : 4);
    size_t const cap = st.bytes / blockBytes;
    if (cap < 1) return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED);
    for (uint32_t first = 0; first < nBlocks; ) {
        uint32_t const n = uint32_t(std::min<size_t>(cap, nBlocks - first));
        char* h = (char*)host + size_t(first) * blockBytes;
        if (0 == direction) {
            TFQ_HIP(hipMemcpyAsync(st.ptr, h, n * blockBytes, hipMemcpyHostToDevice, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
            launch_convert(0, userDbl, to.dbl, to.native, st.ptr, u2n, first, n, nR, nC, layout, trans, conj, to.ilv, s);
            if (also) launch_convert(0, userDbl, also->dbl, also->native, st.ptr, u2n, first, n, nR, nC, layout, trans, conj, also->ilv, s);
        } else {
            launch_convert(1, userDbl, to.dbl, to.native, st.ptr, u2n, first, n, nR, nC, layout, trans, conj, to.ilv, s);
            TFQ_HIP(hipMemcpyAsync(h, st.ptr, n * blockBytes, hipMemcpyDeviceToHost, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        }
        // the stage is reused by the next batch and the host array belongs to the caller
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        first += n;
    }
    TFQ_HIP(hipGetLastError(), TFQMRGPU_STATUS_LAUNCH_FAILED)
    return TFQMRGPU_STATUS_SUCCESS;
}

// ---- RCCL, loaded on first use so that single-GPU callers carry no dependency -------------------
struct UidByValue { char internal[128]; };   // ncclUniqueId
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, UidByValue, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(void const*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    bool load();
};
static Rccl g_rccl;
bool Rccl::load() {
    if (lib) return true;
    for (char const* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return false;
    GetUniqueId  = (decltype(GetUniqueId)) dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    CommDestroy  = (decltype(CommDestroy)) dlsym(lib, "ncclCommDestroy");
    AllReduce    = (decltype(AllReduce))   dlsym(lib, "ncclAllReduce");
    return GetUniqueId && CommInitRank && CommDestroy && AllReduce;
}
enum { kNcclDouble = 8, kNcclMax = 2 };

// max-reduce n doubles of device memory over all ranks (in place, on the solver's stream)
static tfqmrgpuStatus_t reduce_over_ranks(Handle& h, double* red, int n, hipStream_t s) {
    if (h.comm) {
        int const rc = g_rccl.AllReduce(red, red, size_t(n), kNcclDouble, kNcclMax, h.comm, s);
        return rc ? TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED) : TFQMRGPU_STATUS_SUCCESS;
    }
    if (h.reduceFn) {
        double v[4];
        TFQ_HIP(hipMemcpyAsync(v, red, n * sizeof(double), hipMemcpyDeviceToHost, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        h.reduceFn(h.reduceCtx, v, n);
        TFQ_HIP(hipMemcpyAsync(red, v, n * sizeof(double), hipMemcpyHostToDevice, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)
    }
    return TFQMRGPU_STATUS_SUCCESS;
}

// HIP events that are destroyed on every path out of a solve
struct EventList {
    std::vector<hipEvent_t> v;
    bool create(size_t n) {
        v.reserve(n);
        for (size_t i = 0; i < n; ++i) { hipEvent_t e; if (hipSuccess != hipEventCreate(&e)) return false; v.push_back(e); }
        return true;
    }
    ~EventList() { for (auto e : v) (void)hipEventDestroy(e); }
    hipEvent_t operator[](size_t i) const { return v[i]; }
};

// ---- the tfQMR driver -----------------------------------------------------------------------------
// Optional roctx ranges around the two phases of a solve, named like the reference's NVTX ranges (tfqmrgpu_core.hxx:29,
// 176-177,332; there compiled in with -DUSE_NVTX, here switched on with TFQMRGPU_ROCTX=1: libroctx64 is loaded on demand so
// that the library carries no dependency on the tracing runtime).  rocprofv3 --marker-trace shows them.
struct Roctx {
    int (*push)(char const*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        auto const v = std::getenv("TFQMRGPU_ROCTX");
        if (!v || 0 == std::atoi(v)) return;
        // the rocprofiler-sdk flavour: it is the one rocprofv3 listens to (a run that loaded the legacy libroctx64 under
        // rocprofv3 recorded no ranges and did not exit)
        for (char const* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1"}) {
            if (void* lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
                push = (decltype(push))dlsym(lib, "roctxRangePushA");
                pop = (decltype(pop))dlsym(lib, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
struct RoctxRange {
    Roctx const& r;
    RoctxRange(Roctx const& r_, char const* name) : r(r_) { if (r.push) (void)r.push(name); }
    ~RoctxRange() { if (r.pop) (void)r.pop(); }
};
static Roctx const& roctx() { static Roctx const r; return r; }

// Algorithm = reference tfqmrgpu::solve (tfqmrgpu_core.hxx:179-306), restructured:
//   dec35 | v6 | SpMM+v4+dot | dec34 | v5,nrm | decT | x,v7,v6,v7 | SpMM+v5+nrm+dot | decT | decide
//   [ | x | SpMM residual | column records | decide ]   <- only does work when the bound asks for a probe
// The host never waits for the iteration it has just enqueued: it keeps DEPTH iterations in flight
// and reads the control block of iteration `it` (copied to pinned memory behind it) before it
// enqueues iteration it+DEPTH.  Iterations enqueued after the solve has stopped cost a few empty
// launches.  Every rank enqueues the same number of iterations, so collectives always match.
// `d`: the plan's device pointers (d.R set: the right-hand side is the X-shaped vector R and the per-RHS scalars have been set up
// by the refinement, tfq_vec.hip); `early`: a refusal the caller has found already (it still has to travel through the ranks' vote);
// every bound of the history is multiplied by histScale (a refinement cycle's bounds are relative to ITS right-hand side).
struct SolveOutcome { Ctl last; double userFlops = 0; };
static tfqmrgpuStatus_t run_tfqmr(Handle& h, Plan& p, DevPlan const& dIn, double tol, int maxIt, tfqmrgpuStatus_t early, double histScale, SolveOutcome& out) {
    hipStream_t const s = (hipStream_t)h.stream;
    bool const multi = (h.comm != nullptr) || (h.reduceFn != nullptr);
    // small systems: the column operations and the decisions run in the producers' tails (one rank, built-in operator: a reduction
    // over ranks or a foreign multiply sits between the producer and the decision otherwise)
    bool const fold = p.foldOk && !multi && !p.opFn;
    DevPlan const d = [&] { DevPlan x = dIn; x.fold = fold ? 1 : 0; return x; }();
    if (fold && p.selfStale) {
        // the device-resident copy that the folded column operations read: taken again whenever a plan flag has changed since the last
        // one (hashV3, m3: setShadowVector / setShadowMode / setThreeProductMultiply; the copy of setBuffer predates the shadow vector),
        // so that device code never sees a flag that the host's DevPlan does not have
        DevPlan self = d; self.fold = 1; self.first = 0;
        TFQ_HIP(hipMemcpyAsync((void*)d.self, &self, sizeof self, hipMemcpyHostToDevice, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        p.selfStale = false;
    }
    constexpr int DEPTH = Plan::kDepth;
    constexpr int NK = TFQMRGPU_PROFILE_CLASSES;
    // How far the host runs ahead: 2 slots.  Every slot that is still queued when the solve stops costs ~14 empty
    // launches, and enqueuing a slot (~50 us) is never slower than executing one (>= 60 us even for tiny systems), so a
    // deeper queue only adds to the tail (measured: 4 -> 2 gains 8 % on the 2-iteration solves of config 3, 5 % on
    // 1000-block systems, 0.8 % on P2; 1 loses on small systems).  TFQMRGPU_DEPTH = 1..4 overrides.
    static int const depthEnv = lab_switch("TFQMRGPU_DEPTH", 0);
    int ahead = (depthEnv >= 1 && depthEnv <= DEPTH) ? depthEnv : 2;

    if (multi) {
        // One collective in front of every solve: {deepest queue any rank wants, some rank cannot start}.  Every rank must
        // enqueue the same number of slots (the collectives have to match), and a rank that returned here on its own would
        // leave its peers waiting in their first all-reduce for ever -- so the refusal travels through the same reduction.
        // (the buffer of the vote is allocated with the communicator / callback, tfqmrgpuExt_commInit: a rank that failed to
        //  allocate it HERE would leave before the collective and its peers would wait for ever)
        if (!h.voteBuf) return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED);
        double vote[2] = { double(ahead), early ? 1. : 0. };
        TFQ_HIP(hipMemcpyAsync(h.voteBuf, vote, sizeof vote, hipMemcpyHostToDevice, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        if (auto const st = reduce_over_ranks(h, h.voteBuf, 2, s)) return st;
        TFQ_HIP(hipMemcpyAsync(vote, h.voteBuf, sizeof vote, hipMemcpyDeviceToHost, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        ahead = std::min(DEPTH, std::max(1, int(vote[0])));
        if (!early && vote[1] > 0.) early = TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);   // a peer cannot start: nobody does
    }
    if (early) return early;

    // pinned ring + events live with the plan (hipHostMalloc / event creation cost more than a small solve)
    if (!p.ring) {
        TFQ_HIP(hipHostMalloc((void**)&p.ring, DEPTH * sizeof(Ctl), hipHostMallocDefault), TFQMRGPU_STATUS_ALLOCATION_FAILED)
        for (auto& e : p.ringEvent) {
            hipEvent_t ev_ = nullptr;
            if (hipSuccess != hipEventCreateWithFlags(&ev_, hipEventDisableTiming)) {
                for (auto& f : p.ringEvent) { if (f) (void)hipEventDestroy((hipEvent_t)f); f = nullptr; }
                (void)hipHostFree(p.ring); p.ring = nullptr;
                return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED);
            }
            e = (void*)ev_;
        }
    }
    Ctl* const ring = (Ctl*)p.ring;
    hipEvent_t ev[DEPTH];
    for (int i = 0; i < DEPTH; ++i) ev[i] = (hipEvent_t)p.ringEvent[i];
    // profiling: NK+1 timing events per in-flight iteration, event k sits in front of kernel class k
    int const prof = p.profiling;
    // level 2: only the fused multiplies are bracketed (class k needs events k and k + 1)
    auto const timed = [prof](int k) { return 1 == prof || (2 == prof && (TFQMRGPU_PROF_SPMM_V4_DOT == k || TFQMRGPU_PROF_SPMM_V5_NRM_DOT == k)); };
    EventList pev;
    if (prof && !pev.create(size_t(DEPTH) * (NK + 1))) return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED);

    { RoctxRange const range(roctx(), "tfQMR preparation"); TFQ_HIP(vec_launch(VEC_SETUP, d, tol, maxIt, s), TFQMRGPU_STATUS_LAUNCH_FAILED) }
    RoctxRange const range(roctx(), "tfQMR iterations");

    tfqmrgpuStatus_t fail = TFQMRGPU_STATUS_SUCCESS;

    // user-defined operator (tfqmrgpu_ext.h section 5): gather the operand into the caller's block order,
    // let the callback enqueue Y = A*X, apply the fused epilogue to the product
    auto const userOp = (tfqmrgpuOperator_t)p.opFn;
    double userFlops = 0;
    size_t const vecBytes = size_t(p.nnzbX) * 2 * p.LM * p.LN * (d.dbl ? 8 : 4);
    auto const up256 = [](size_t n) { return (n + 255) & ~size_t(255); };
    if (userOp && !p.opScratch) {
        size_t const bytes = 2 * up256(vecBytes) + up256(size_t(p.nnzbX) * 4) + up256(size_t(p.nnzbX) * 2);
        TFQ_HIP(hipMalloc((void**)&p.opScratch, bytes ? bytes : 256), TFQMRGPU_STATUS_ALLOCATION_FAILED)
        char* const q = p.opScratch + 2 * up256(vecBytes);
        std::vector<uint16_t> cu(p.nnzbX);      // compressed block column per block, caller's order
        for (uint32_t u = 0; u < p.nnzbX; ++u) cu[u] = p.colindx[u];
        TFQ_HIP(hipMemcpyAsync(q, p.i2u.data(), size_t(p.nnzbX) * 4, hipMemcpyHostToDevice, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipMemcpyAsync(q + up256(size_t(p.nnzbX) * 4), cu.data(), size_t(p.nnzbX) * 2, hipMemcpyHostToDevice, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)   // cu goes out of scope
    }
    auto multiply = [&](int epi, DevPlan const& d) {     // d: the plan with the first-iteration flag of the slot
        if (!userOp) { spmm_launch(epi, d, s); return; }
        char* const xu = p.opScratch; char* const yu = xu + up256(vecBytes);
        auto const i2u = (uint32_t const*)(yu + up256(vecBytes));
        auto const colU = (uint16_t const*)((char const*)i2u + up256(size_t(p.nnzbX) * 4));
        launch_convert(1, d.dbl, d.dbl, (EPI_RESIDUAL == epi) ? d.x : d.v6, xu, d.u2i, 0, p.nnzbX, p.LM, p.LN,
                       TFQMRGPU_LAYOUT_RRRRIIII, false, false, p.ilv, s);
        double fl = 0;
        auto const st = userOp(p.opCtx, yu, xu, colU, p.nnzbX, p.nCols, p.LM, p.LN, p.precision, (tfqmrgpuStream_t)s, &fl);
        if (st && !fail) fail = st;   // the slot is completed all the same: its reduction tells the other ranks
        userFlops += fl;
        epilogue_launch(epi, d, yu, i2u, s);
    };

    // part 0: all kernels of one iteration slot (most of them gate themselves off); 1: without the probe; 2: probe only
    static double const kOne = 1.;
    // the max-reduction of a slot; a rank that has failed marks the record, so that every rank stops at THAT slot (the failing rank
    // goes on completing slots until then, see the loop below; a rank whose device is lost cannot, its peers then wait in RCCL)
    auto reduce = [&](int what) {
        if (fail && hipSuccess != hipMemcpyAsync(&d.ctl->red[3 * what + 2], &kOne, sizeof kOne, hipMemcpyHostToDevice, s)) return;
        auto const st = reduce_over_ranks(h, &d.ctl->red[3 * what], 3, s);
        if (st && !fail) fail = st;
    };
    auto launches = [&](int slot, int part, bool firstIteration) {
        DevPlan const dSlot = [&] { DevPlan x = d; x.first = firstIteration ? 1 : 0; return x; }();
        auto mark = [&](int k) {
            if (!(k < NK && timed(k)) && !(k > 0 && timed(k - 1))) return;
            if (hipSuccess != hipEventRecord(pev[slot * (NK + 1) + k], s) && !fail) fail = TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
        };
        if (part != 2) {
            mark(TFQMRGPU_PROF_DEC35);            (void)vec_launch(VEC_DEC35, d, 0, 0, s);
            mark(TFQMRGPU_PROF_XPAY_V6);          (void)vec_launch(VEC_XPAY_V6, dSlot, 0, 0, s);
            mark(TFQMRGPU_PROF_SPMM_V4_DOT);      multiply(EPI_XPAY_DOT, dSlot);
            // (fold: dec34, decT, decT + decide and the probe's column records + decision run in the tails of the kernels in front of them)
            mark(TFQMRGPU_PROF_DEC34);            if (!fold) (void)vec_launch(VEC_DEC34, d, 0, 0, s);
            mark(TFQMRGPU_PROF_V5_NRM);           (void)vec_launch(VEC_V5_NRM, dSlot, 0, 0, s);
            mark(TFQMRGPU_PROF_DECT_C67);         if (!fold) (void)vec_launch(VEC_DECT_C67, d, 0, 0, s);
            mark(TFQMRGPU_PROF_X_V6_V7);          (void)vec_launch(VEC_X_V6_V7, dSlot, 0, 0, s);
            mark(TFQMRGPU_PROF_SPMM_V5_NRM_DOT);  multiply(EPI_AXPY_NRM_DOT, dSlot);
            mark(TFQMRGPU_PROF_DECT_FINAL);       if (!fold) (void)vec_launch(VEC_DECT_FIN, d, 0, 0, s);
            mark(TFQMRGPU_PROF_DECIDE);
            if (multi) {
                launch_decide(d, 1, s);
                reduce(0);
                launch_decide(d, 2, s);
            } else if (!fold) launch_decide(d, 0, s);
            mark(TFQMRGPU_PROF_PROBE);
        }
        if (part != 1) {
            (void)vec_launch(VEC_X_FLUSH, d, 0, 0, s);
            multiply(EPI_RESIDUAL, dSlot);
            if (!fold) (void)vec_launch(VEC_PROBE_COL, d, 0, 0, s);
            if (multi) {
                launch_probe_decide(d, 1, s);
                reduce(1);
                launch_probe_decide(d, 2, s);
            } else if (!fold) launch_probe_decide(d, 0, s);
        }
        mark(NK);
    };
    auto enqueue = [&](int slot, int part, bool firstIteration) {
        launches(slot, part, firstIteration);
        // a copy that did not start would be read as a stale "still running": both calls are checked
        if ((hipSuccess != hipMemcpyAsync(&ring[slot], d.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, s) ||
             hipSuccess != hipEventRecord(ev[slot], s)) && !fail) fail = TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
    };

    Ctl last{};
    last.state = (maxIt > 0) ? 0 : 3; last.residual2_reached = 1e300; last.iterations_needed = maxIt;
    int enq = 0, seen = 0;
    if (userOp) {
        // the callback's kernels cannot look at the control block, so nothing is enqueued ahead of a decision:
        // one host round trip per iteration (and one per probe), like the reference (tfqmrgpu_core.hxx:235-304)
        for (int it = 0; it < maxIt && !fail && 0 == last.state; ++it) {
            for (int part = 1; part <= 2 && !fail; ++part) {   // (a failing operator still completes its slot: see reduce())
                if (2 == part && !(0 == last.state && last.probe)) break;
                enqueue(0, part, 0 == it);
                if (hipSuccess != hipEventSynchronize(ev[0])) { fail = TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED); break; }
                last = ring[0];
                if (1 == part) p.boundHistory.push_back(last.max_bound2 * histScale);
            }
            if (prof) for (int k = 0; k < NK; ++k) {
                float ms = 0;
                if (timed(k) && hipSuccess == hipEventElapsedTime(&ms, pev[k], pev[k + 1])) {
                    p.profMs[k] += ms; p.profLaunches[k] += 1;
                    if (0 == it) { p.profFirstMs[k] += ms; p.profFirstLaunches[k] += 1; }
                }
            }
        }
    } else
    while (enq < std::min(ahead, maxIt)) { enqueue(enq % DEPTH, 0, 0 == enq); ++enq; }   // slot n runs iteration n (or nothing)
    // A rank that has failed on the way (fail != 0) does not leave on its own when there are peers: they have slots with
    // collectives enqueued ahead, which it must match.  It keeps completing slots -- reduce() marks every one of its records -- until
    // the control block shows state 4 (every rank stops at the slot whose reduction carried the mark) or the iterations run out.
    while (seen < enq && (!fail || multi)) {
        int const slot = seen % DEPTH;
        if (hipSuccess != hipEventSynchronize(ev[slot])) { if (!fail) fail = TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED); break; }   // the device is gone: nothing left to match
        int const nprobes_before = last.nprobes;
        last = ring[slot];
        ++seen;
        p.boundHistory.push_back(last.max_bound2 * histScale);
        if (prof) for (int k = 0; k < NK; ++k) {
            if (!timed(k)) continue;
            bool const gated = (TFQMRGPU_PROF_PROBE == k && last.nprobes == nprobes_before); // probe not requested
            float ms = 0;
            if (hipSuccess == hipEventElapsedTime(&ms, pev[slot * (NK + 1) + k], pev[slot * (NK + 1) + k + 1])) {
                if (gated) { p.profGatedMs[k] += ms; p.profGatedLaunches[k] += 1; }
                else {
                    p.profMs[k] += ms; p.profLaunches[k] += 1;
                    if (1 == seen) { p.profFirstMs[k] += ms; p.profFirstLaunches[k] += 1; }   // the slot of the first iteration
                }
            }
        }
        if (last.state != 0) break;
        if (enq < maxIt) { enqueue(enq % DEPTH, 0, false); ++enq; }
    }
    (void)hipStreamSynchronize(s);
    // iterations that were enqueued ahead and found the solve finished: their launches return at once
    if (prof) for (; seen < enq; ++seen) {
        int const slot = seen % DEPTH;
        for (int k = 0; k < NK; ++k) {
            float ms = 0;
            if (timed(k) && hipSuccess == hipEventElapsedTime(&ms, pev[slot * (NK + 1) + k], pev[slot * (NK + 1) + k + 1])) {
                p.profGatedMs[k] += ms; p.profGatedLaunches[k] += 1;
            }
        }
    }
    if (hipSuccess != hipGetLastError() && !fail) fail = TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
    if (fail) return fail;
    if (4 == last.state) return TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);   // another rank reported a failure; all ranks stopped at the same slot
    out.last = last; out.userFlops = userFlops;
    return TFQMRGPU_STATUS_SUCCESS;
}

// flop model of the reference: tfqmrgpu_linalg.hxx:587,625,684,703 and tfqmrgpu_blocksparse.hxx:198
struct FlopModel {
    double fMult, fDot, fNrm, fAxp;
    explicit FlopModel(Plan const& p) {
        double const blk = double(p.LM) * p.LN, nX = p.nnzbX;
        fMult = double(p.nPairs()) * 8. * p.LM * blk; fDot = nX * 8. * blk; fNrm = nX * 4. * blk; fAxp = nX * 8. * blk;
    }
    double solve(Ctl const& c) const { return fNrm + c.iteration * (2 * fMult + 2 * fDot + 2 * fNrm + 10 * fAxp) + c.nprobes * (fMult + fNrm); }
};

// Mixed precision 'm' (reference: dormant, tfqmrgpu.cu:42 "load float, multiply-accumulate double, store float"; documented as
// "start with float and converge double", tfqmrgpu.h:72).  Iterative refinement: x, B, A in double; per cycle r = b - A x in double,
// A d = r solved by the complex<float> tfQMR (its kernels unchanged; the right-hand side is the X-shaped R), x += d in double.  The
// float iteration moves half the bytes of the double one; a cycle gains the digits a float solve delivers, the refinement's own
// stopping test is max_rhs |b - A x| / |b| <= threshold in double.  maxIterations bounds the SUM of the inner iterations.
static tfqmrgpuStatus_t run_mixed(Handle& h, Plan& p, double tol, int maxIt) {
    hipStream_t const s = (hipStream_t)h.stream;
    SolveOutcome o;
    bool const multi = (h.comm != nullptr) || (h.reduceFn != nullptr);
    // ONE failure protocol for several ranks: the first collective of a mixed-precision solve is the refinement's
    // max-reduction of {max |r|^2/|b|^2, a value is not finite, a rank failed} (3 doubles) -- not the vote of run_tfqmr (2 doubles).  A
    // rank that cannot start (no buffer; a user-defined operator, which 'm' plans refuse) therefore takes part in THAT reduction with
    // the third value set: its peers read it in their cycle 0 and every rank leaves at the same collective, the refusing one with
    // its own status, the others with "a peer failed".  A rank that fails later, between two collectives, hands its failure to the
    // next one it would have entered: the vote of the inner solve (`early`), or the next refinement reduction (`mine`).
    auto const leave_marked = [&](tfqmrgpuStatus_t st) -> tfqmrgpuStatus_t {
        if (!multi) return st;
        if (!h.voteBuf) return st;                           // (allocated with the communicator / callback; without it no collective can be entered)
        double const mark[3] = { 0., 0., 1. };
        if (hipSuccess != hipMemcpyAsync(h.voteBuf, mark, sizeof mark, hipMemcpyHostToDevice, s)) return st;
        if (hipSuccess != hipStreamSynchronize(s)) return st;
        (void)reduce_over_ranks(h, h.voteBuf, 3, s);
        (void)hipStreamSynchronize(s);
        return st;
    };
    if (!p.buffer)  return leave_marked(TFQ_ERR(TFQMRGPU_POINTER_INVALID));
    if (p.opFn)     return leave_marked(TFQ_ERR(TFQMRGPU_NO_IMPLEMENTATION));  // user-defined operators: 'z' and 'c' only
    DevPlan const d = resolve(p), dz = resolveZ(p);
    FlopModel const fm(p);
    RefineArgs a{};
    a.d = d; a.xz = (double*)dz.x; a.Bz = (double const*)dz.B; a.Yz = (double const*)d.v4;   // A x lives in v4 ... v7, free between two inner solves
    a.bn2z = (double*)(p.buffer + p.wBn2z.offset); a.refine = (double*)(p.buffer + p.wRefine.offset); a.ilvZ = p.ilvZ;
    if (maxIt <= 0) TFQ_HIP(hipMemsetAsync(a.xz, 0, p.wXz.bytes, s), TFQMRGPU_STATUS_LAUNCH_FAILED)   // no iteration at all: x = 0 is the answer
    // what a float solve is asked for per cycle: a quarter of what is missing, but not more digits than float iterations deliver; a solve
    // that reaches its floor earlier ends itself (Ctl::stallStop) and the next cycle continues from the double residual
    double const kInnerFloor = 3e-5;
    int used = 0, strikes = 0, lastIts = 0;
    bool brokeDown = false, firstStalled = false;
    double res2 = 1e300, prev2 = 1e300, bestGain = 1.;
    tfqmrgpuStatus_t result = TFQMRGPU_STATUS_MAX_ITERATIONS;
    tfqmrgpuStatus_t mine = TFQMRGPU_STATUS_SUCCESS;     // this rank's own failure between two collectives: rides the next refinement reduction
    for (int cycle = 0; ; ++cycle) {
        if (cycle > 0) { spmm_apply(dz, dz.x, (void*)a.Yz, s); p.flops_performed += fm.fMult; }
        a.cycle = cycle; a.innerTol = 1e-4; a.innerMaxIt = std::max(0, maxIt - used);
        launch_refine_residual(a, s);
        p.flops_performed += fm.fNrm;
        if (!mine && hipSuccess != hipGetLastError()) mine = TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
        if (mine && multi) {   // (k_refine_max has written {.., .., 0}: this rank's mark behind it, on the same stream)
            static double const kOne = 1.;
            (void)hipMemcpyAsync(a.refine + 2, &kOne, sizeof kOne, hipMemcpyHostToDevice, s);
        }
        if (auto const st = reduce_over_ranks(h, a.refine, 3, s)) return mine ? mine : st;
        if (mine) return mine;                                              // every rank leaves behind this reduction
        double v[3];
        TFQ_HIP(hipMemcpyAsync(v, a.refine, sizeof v, hipMemcpyDeviceToHost, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        if (v[2] > 0.) return TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);      // a rank failed: every rank reads the same value and leaves here
        res2 = v[0];
        p.cycleResidual.push_back(std::sqrt(res2));
        p.refinementCycles = cycle;
        if (v[1] > 0.) { result = TFQMRGPU_STATUS_BREAKDOWN; break; }      // the residual is not finite
        if (res2 <= tol * tol) { result = TFQMRGPU_STATUS_SUCCESS; break; }
        if (used >= maxIt) break;
        double lastGain = 1.;
        if (cycle > 0 && prev2 > 0. && prev2 < 1e299) { lastGain = std::sqrt(res2 / prev2); bestGain = std::min(bestGain, lastGain); }   // what the last cycle delivered
        // (r04) the float floor of THIS plan: where its first float solve gave up by itself (Ctl::stallStop).  The next solve of the plan -- same A, new B, the
        // use the reference's README names (:97-104) -- asks its first cycle for twice that instead of searching for it again (P2: 9 -> 7 iterations)
        if (1 == cycle && firstStalled && lastGain < 1.) p.mixedFloor = lastGain;
        if (cycle > 0) {   // two cycles in a row gained less than a factor 2: give up (a breakdown of the last float solve is reported as one)
            strikes = (res2 > 0.25 * prev2) ? strikes + 1 : 0;
            if (strikes >= 2) { if (brokeDown) result = TFQMRGPU_STATUS_BREAKDOWN; break; }
        }
        prev2 = res2;
        // What this cycle is asked for.  The first cycle: a quarter of what is missing -- the floor of a float solve is not known beforehand (2.4e-4 on
        // P2, 2.4e-7 on a 32 x 32 stencil); a solve that reaches it ends itself (Ctl::stallStop) at the price of 2-3 iterations.  Later cycles know
        // what a cycle has delivered (bestGain): if what is missing is more than that, it is split evenly over the cycles it will take anyway, so
        // that none of them runs into its floor (P2: 9 + 9 + 5 -> 9 + 6 + 5 float iterations); systems whose first cycle delivers nearly everything keep
        // their two cycles.  Lab builds: TFQMRGPU_MIXED_SPLIT=0 a quarter of what is missing in every cycle, =2 an even split into cycles of at
        // most 3 digits from the first cycle on (P2: 7 + 7 + 5, but fast-converging systems then take three cycles: profiles/r03_mixed_policy.txt).
        static int const splitEnv = lab_switch("TFQMRGPU_MIXED_SPLIT", 1);
        double const need = tol / std::sqrt(res2);                                   // the factor still to gain, < 1
        double ask = 0.25 * need;
        if (2 == splitEnv) {
            int const n = std::max(1, int(std::ceil(-std::log10(need) / 3.0)));
            if (n > 1) ask = std::pow(need, 1. / n); else ask = 0.5 * need;
        } else if (1 == splitEnv && cycle > 0 && bestGain < 1.) {
            double const cap = std::min(0.5, 2. * bestGain);                         // what a cycle delivers without searching for its floor
            int const n = std::max(1, int(std::ceil(std::log(need) / std::log(cap))));
            if (n > 1) ask = std::pow(need, 1. / n);
        }
        static int const floorEnv = lab_switch("TFQMRGPU_MIXED_FLOOR", 1), predictEnv = lab_switch("TFQMRGPU_MIXED_PREDICT", 1);
        if (0 == cycle && floorEnv && p.mixedFloor > 0.) ask = std::max(ask, 2. * p.mixedFloor);   // (twice the floor: the last iterations in front of a floor are its slowest)
        // (r04) a LAST cycle that has less than two digits to gain does not search: the previous cycle's rate per iteration says how many iterations
        // that takes (+ 1), the float solve runs them without a probe of its own -- the refinement's residual in double decides anyway -- and stops
        // (P2's third cycle: 5 iterations and three float probes for a factor 4 -> 3 iterations and the one probe at the end)
        int innerIts = maxIt - used;
        bool predicted = false;
        if (predictEnv && cycle > 0 && need >= 0.01 && lastIts > 0 && lastGain < 0.5) {
            double const rate = std::pow(lastGain, 1. / lastIts);
            int const want = int(std::ceil(std::log(0.5 * need) / std::log(rate))) + 1;
            innerIts = std::min(innerIts, std::max(2, std::min(want, lastIts)));
            ask = 0.5 * need;
            predicted = true;
        }
        double const innerTol = std::min(0.5, std::max(kInnerFloor, ask));
        double const t2[2] = { innerTol * innerTol, predicted ? 0. : innerTol * innerTol * 1e4 };   // Ctl::tol2, Ctl::target_bound2 (every rank the same values; 0: no probe before the last iteration)
        // (a failure here travels through the vote in front of the inner solve: the peers are about to enter THAT collective)
        tfqmrgpuStatus_t const early = (hipSuccess == hipMemcpyAsync(&d.ctl->tol2, t2, sizeof t2, hipMemcpyHostToDevice, s))
                                       ? TFQMRGPU_STATUS_SUCCESS : TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
        // run_tfqmr's own protocol makes every rank come back at the same point: a refusal through its vote, a failure inside through the
        // third value of its slot records (state 4) -- so a status from it ends the refinement on every rank alike
        tfqmrgpuStatus_t early2 = early;
        if (predicted) {   // the inner solve's own iteration limit (k_refine_init_col has written maxIt - used)
            int32_t const lim = innerIts;
            if (!early2 && hipSuccess != hipMemcpyAsync(&d.ctl->maxIterations, &lim, sizeof lim, hipMemcpyHostToDevice, s)) early2 = TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
        }
        if (auto const st = run_tfqmr(h, p, d, innerTol, innerIts, early2, res2, o)) return st;
        used += o.last.iteration;
        lastIts = o.last.iteration;
        if (0 == cycle) firstStalled = (3 == o.last.state && o.last.iteration < innerIts);   // ended by itself at its floor, not at a limit
        p.cycleIterations.push_back(o.last.iteration);
        brokeDown = (2 == o.last.state);                                    // every right-hand side of this float solve broke down
        p.flops_performed += fm.solve(o.last) - fm.fNrm;                    // (|r|^2 of the set-up is the refinement's, counted above)
        launch_refine_update(a, s);
        p.flops_performed += 2. * p.nnzbX * p.LM * p.LN;
    }
    TFQ_HIP(hipGetLastError(), TFQMRGPU_STATUS_LAUNCH_FAILED)
    p.flops_performed_all += p.flops_performed;
    p.residuum_reached = std::sqrt(std::max(res2, 1.4e-76 * 1.4e-76));
    p.iterations_needed = (TFQMRGPU_STATUS_SUCCESS == result) ? used : maxIt;
    return result;
}

static tfqmrgpuStatus_t run_solve(Handle& h, Plan& p, double tol, int maxIt) {
    for (int k = 0; k < TFQMRGPU_PROFILE_CLASSES; ++k) { p.profLaunches[k] = 0; p.profMs[k] = 0; p.profGatedLaunches[k] = 0; p.profGatedMs[k] = 0; p.profFirstLaunches[k] = 0; p.profFirstMs[k] = 0; }
    p.boundHistory.clear(); p.cycleResidual.clear(); p.cycleIterations.clear(); p.refinementCycles = 0;
    p.iterations_needed = maxIt; p.flops_performed = 0;
    if ('m' == p.precision) return run_mixed(h, p, tol, maxIt);
    // what this rank can tell before it touches the device
    tfqmrgpuStatus_t early = TFQMRGPU_STATUS_SUCCESS;
    if (!p.buffer) early = TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    else if ('z' != p.precision && 'c' != p.precision) early = err(TFQMRGPU_PRECISION_MISSMATCH, __LINE__ % 10000, p.precision);
    SolveOutcome o;
    if (auto const st = run_tfqmr(h, p, early ? DevPlan{} : resolve(p), tol, maxIt, early, 1., o)) return st;
    Ctl const& last = o.last;
    FlopModel const fm(p);
    p.flops_performed = fm.solve(last);
    if (p.opFn) p.flops_performed += o.userFlops - (2. * last.iteration + last.nprobes) * fm.fMult;  // the operator's own count
    p.flops_performed_all += p.flops_performed;
    p.residuum_reached = std::sqrt(last.residual2_reached);
    p.iterations_needed = (1 == last.state) ? last.iterations_needed : maxIt;
    switch (last.state) {
        case 1: return TFQMRGPU_STATUS_SUCCESS;
        case 2: return TFQMRGPU_STATUS_BREAKDOWN;
        default: return TFQMRGPU_STATUS_MAX_ITERATIONS;
    }
}

static tfqmrgpuStatus_t upload(void* dst, void const* src, size_t bytes, hipStream_t s) {
    if (0 == bytes) return TFQMRGPU_STATUS_SUCCESS;
    TFQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
    return TFQMRGPU_STATUS_SUCCESS;
}

} // namespace tfq

// ====================================================================================================
extern "C" {

tfqmrgpuStatus_t tfqmrgpu_bsrsv_allowedBlockSizes(int32_t* number, int32_t* blockSizes, int const arrayLength) {
    // reference tfqmrgpu.cu:75-93, including its quirks: the output array is only cleared when
    // *number != 0 on entry, and a pair is stored only while 2*n < arrayLength
    if (nullptr == number) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (nullptr == blockSizes) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (0 != *number) for (int i = 0; i < arrayLength; ++i) blockSizes[i] = 0;
    int n = 0, i = 0;
    for (auto const& sz : kAllowedBlockSizes) {
        ++n;
        if (2 * n < arrayLength) { blockSizes[2 * i] = sz[0]; blockSizes[2 * i + 1] = sz[1]; ++i; }
    }
    *number = n;
    return (n == i) ? TFQMRGPU_STATUS_SUCCESS : TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_blockSizeMissing(int const ldA, int const ldB) {
    // reference tfqmrgpu.cu:95-106: 12 + char=ldA + line=ldB
    return blockSizeAllowed(ldA, ldB) ? 0 : err(TFQMRGPU_BLOCKSIZE_MISSING, ldB, ldA);
}

tfqmrgpuStatus_t tfqmrgpuCreateHandle(tfqmrgpuHandle_t* handle) {       // reference tfqmrgpu.cu:110-115
    if (nullptr == handle)  return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (nullptr != *handle) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    auto h = new (std::nothrow) Handle();
    if (!h) return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED);
    *handle = (tfqmrgpuHandle_t)h;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuDestroyHandle(tfqmrgpuHandle_t handle) {       // reference tfqmrgpu.cu:117-121
    if (nullptr == handle) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    auto h = (Handle*)handle;
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    if (h->voteBuf) (void)hipFree(h->voteBuf);
    delete h;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuSetStream(tfqmrgpuHandle_t handle, tfqmrgpuStream_t const streamId) { // tfqmrgpu.cu:124-128
    if (nullptr == handle) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    ((Handle*)handle)->stream = (void*)streamId;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuGetStream(tfqmrgpuHandle_t handle, tfqmrgpuStream_t* streamId) {     // tfqmrgpu.cu:130-134
    if (nullptr == handle || nullptr == streamId) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    *streamId = (tfqmrgpuStream_t)((Handle*)handle)->stream;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuCreateWorkspace(void** pBuffer, size_t const bytes, char const memType) { // tfqmrgpu.cu:682-694
    if (nullptr == pBuffer) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    hipError_t const e = ('m' == lower(memType)) ? hipMallocManaged(pBuffer, bytes) : hipMalloc(pBuffer, bytes);
    return (hipSuccess == e) ? TFQMRGPU_STATUS_SUCCESS : TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED);
}

tfqmrgpuStatus_t tfqmrgpuDestroyWorkspace(void* pBuffer) {              // tfqmrgpu.cu:696-698 (raw runtime code)
    return (tfqmrgpuStatus_t)hipFree(pBuffer);
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_createPlan(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t* plan, int const mb,
    int32_t const* bsrRowPtrA, int const nnzbA, int32_t const* bsrColIndA,
    int32_t const* bsrRowPtrX, int const nnzbX, int32_t const* bsrColIndX,
    int32_t const* bsrRowPtrB, int const nnzbB, int32_t const* bsrColIndB,
    int const indexOffset, int const echo)
{
    (void)handle;
    if (nullptr == plan)  return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (nullptr != *plan) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);     // tfqmrgpu.cu:161
    auto p = new (std::nothrow) Plan();
    if (!p) return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED);
    p->indexOffset = indexOffset;
    tfqmrgpuStatus_t st;
    try {   // no exception may cross the C boundary: a C or Fortran caller would be terminated instead of getting a status
        st = analyse(*p, mb, bsrRowPtrA, nnzbA, bsrColIndA, bsrRowPtrX, nnzbX, bsrColIndX,
                     bsrRowPtrB, nnzbB, bsrColIndB, indexOffset, echo);
    } catch (std::bad_alloc const&) { st = TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED); }
    if (st) { delete p; return st; }   // (the reference leaks the plan on its error paths)
    *plan = (tfqmrgpuBsrsvPlan_t)p;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_destroyPlan(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan) { // tfqmrgpu.cu:353-361
    (void)handle;
    auto p = asPlan(plan);
    if (!p) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (p->ring) { (void)hipHostFree(p->ring); for (auto e : p->ringEvent) (void)hipEventDestroy((hipEvent_t)e); }
    if (p->opScratch) (void)hipFree(p->opScratch);
    p->magic = 0;
    delete p;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_bufferSize(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan,
    int const ldA, int const blockDim, int const ldB, int const RhsBlockDim, char const precision, size_t* pBufferSizeInBytes)
{
    // reference tfqmrgpu.cu:364-412
    int const LM = ldA, LN = ldB;
    if (LM != blockDim)    return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (LM > LN)           return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (LN != RhsBlockDim) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    auto p = asPlan(plan);
    if (!p || !handle) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    char prec;
    switch (lower(precision)) {            // f,c -> c ; m -> m ; everything else -> z  (tfqmrgpu.cu:383-390)
        case 'f': case 'c': prec = 'c'; break;
        case 'm': prec = 'm'; break;
        default:  prec = 'z';
    }
    if (nullptr == pBufferSizeInBytes) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (!blockSizeAllowed(LM, LN)) return err(TFQMRGPU_BLOCKSIZE_MISSING, LN, LM); // tfqmrgpu.cu:70
    // 'm': accepted here AND solved (the reference accepts it here and refuses it in solve, tfqmrgpu.cu:42-44, 386): float vectors for
    // the iteration plus x, B and A in double (tfq_plan.cpp: layoutBuffer)
    tfqmrgpuStatus_t st;
    try { st = layoutBuffer(*p, LM, LN, prec); }
    catch (std::bad_alloc const&) { return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED); }
    if (p->opScratch) { (void)hipFree(p->opScratch); p->opScratch = nullptr; }   // sized for the previous block shape
    p->precision = prec;
    p->buffer = nullptr;
    *pBufferSizeInBytes = p->bufferBytes;
    return st;
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_setBuffer(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, void* const pBuffer) {
    // reference tfqmrgpu.cu:415-450: register the buffer, create v3, upload the index lists
    if (nullptr == pBuffer) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    auto p = asPlan(plan); auto h = (Handle*)handle;
    if (!p || !h) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (0 == p->LM) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);        // bufferSize has not been called
    hipStream_t const s = (hipStream_t)h->stream;
    p->buffer = (char*)pBuffer;
    auto at = [&](Window const& w) { return (void*)(p->buffer + w.offset); };
    auto up = [&](Window const& w, void const* src, size_t bytes) { return upload(at(w), src, bytes, s); };
    tfqmrgpuStatus_t st;
    auto const& c = p->chunks;
    std::vector<int32_t> orig(p->original_bsrColIndX);
    for (auto& o : orig) o -= p->indexOffset;
    if ((st = up(p->wChunkFirst, c.first.data(), c.first.size() * 4))) return st;
    if ((st = up(p->wChunkCol, c.col.data(), c.col.size() * 4))) return st;
    if ((st = up(p->wOrder, c.order.data(), c.order.size() * 4))) return st;
    if ((st = up(p->wColChunkPtr, c.colPtr.data(), c.colPtr.size() * 4))) return st;
    if ((st = up(p->wColStart, p->colStart.data(), p->colStart.size() * 4))) return st;
    if ((st = up(p->wOrigCol, orig.data(), orig.size() * 4))) return st;
    if (!p->colBatch.empty() && (st = up(p->wColBatch, p->colBatch.data(), p->colBatch.size()))) return st;
    if (!p->colBatch.empty() && (st = up(p->wOrderB, c.orderB.data(), c.orderB.size() * 4))) return st;
    if ((st = up(p->wBofX, p->bOfX.data(), p->bOfX.size() * 4))) return st;
    if ((st = up(p->wStarts, p->starts_i.data(), p->starts_i.size() * 4))) return st;
    if ((st = up(p->wPairs, p->pairs_i.data(), p->pairs_i.size() * 4))) return st;
    if ((st = up(p->wSubset, p->subset_i.data(), p->subset_i.size() * 4))) return st;
    if ((st = up(p->wBColPtr, p->bColPtr.data(), p->bColPtr.size() * 4))) return st;
    if ((st = up(p->wBList, p->bList.data(), p->bList.size() * 4))) return st;
    if ((st = up(p->wU2I, p->u2i.data(), p->u2i.size() * 4))) return st;
    if ((st = up(p->wRowI, p->rowI.data(), p->rowI.size() * 4))) return st;
    TFQ_HIP(hipMemsetAsync(at(p->wCtl), 0, p->wCtl.bytes, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
    TFQ_HIP(hipMemsetAsync(at(p->wFold), 0, p->wFold.bytes, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
    TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)     // host vectors above go out of scope
    DevPlan const d = resolve(*p);
    {   // the device-resident copy that the folded column operations read (pointers and sizes only: nothing in it changes with a solve)
        static_assert(sizeof(DevPlan) <= 1024, "window wSelf");
        DevPlan self = d; self.fold = 1;
        TFQ_HIP(hipMemcpyAsync(at(p->wSelf), &self, sizeof self, hipMemcpyHostToDevice, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)
    }
    if (TFQMRGPU_SHADOW_GLIBC_RAND == p->shadowMode) {
        size_t const n = size_t(p->nnzbX) * 2 * p->LM * p->LN;
        std::vector<float> v3(n);
        GlibcRand rng(1);
        float const denom = 1. / 2147483647;                            // tfqmrgpu_linalg.hxx:799-801
        for (size_t i = 0; i < n; ++i) v3[i] = rng.next() * denom;
        st = transfer_blocks(*p, s, 0, false, Target{d.v3, false, p->ilv}, v3.data(), d.u2i, p->nnzbX, p->LM, p->LN, TFQMRGPU_LAYOUT_RRRRIIII, false, false);
        if (st) return st;
        p->v3IsHash = false; p->selfStale = true;
    } else {
        p->v3IsHash = true; p->selfStale = true;
        launch_shadow_hash(d, s);
        if (hipSuccess != hipGetLastError()) return TFQ_ERR(TFQMRGPU_STATUS_RANDOM_GEN_FAILED);
    }
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_getBuffer(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, void** pBuffer) { // tfqmrgpu.cu:453-462
    (void)handle;
    auto p = asPlan(plan);
    if (!p || !pBuffer) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    *pBuffer = (void*)p->buffer;
    return (nullptr == *pBuffer) ? TFQ_ERR(TFQMRGPU_POINTER_INVALID) : TFQMRGPU_STATUS_SUCCESS;
}

// reference tfqmrgpu::set_or_getMatrix, tfqmrgpu.cu:467-603
static tfqmrgpuStatus_t set_or_get(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, char const var, void* values,
    char const precision, char const transposition, tfqmrgpuDataLayout_t const layout, bool const is_get)
{
    switch (layout) {
        case TFQMRGPU_LAYOUT_RRRRIIII: case TFQMRGPU_LAYOUT_RIRIRIRI: case TFQMRGPU_LAYOUT_RRIIRRII: break;
        default: return err(TFQMRGPU_DATALAYOUT_UNKNOWN, layout % 10000);          // line field = layout
    }
    bool conj = false, trans = false;
    char const tr = lower(transposition);   // '*' | 32 == '*'
    switch (tr) {
        case 'h': case 'c': conj = true; trans = true; break;
        case '*': conj = true; break;
        case 't': trans = true; break;
        case 'n': break;
        default: return err(TFQMRGPU_TANSPOSITION_UNKNOWN, __LINE__ % 10000, tr);
    }
    auto p = asPlan(plan); auto h = (Handle*)handle;
    if (!p || !h) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (0 == p->LM) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    uint32_t nnzb = 0; int nR = p->LM, nC = p->LN; int which = 0;
    switch (lower(var)) {
        case 'a': nnzb = p->nnzbA; nC = p->LM; trans = !trans; which = 0; if (!is_get) p->mixedFloor = 0; break; // A is stored transposed (tfqmrgpu.cu:514-517); (a new operator: the remembered float floor goes)
        case 'b': nnzb = p->nnzbB; which = 1; break;
        case 'x': nnzb = p->nnzbX; which = 2; break;
        default: return err(TFQMRGPU_VARIABLENAME_UNKNOWN, __LINE__ % 10000, var);
    }
    if (nnzb < 1) return TFQMRGPU_STATUS_SUCCESS;
    if (nullptr == p->buffer) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    bool const mixed = ('m' == p->precision);
    bool const is_double = ('z' == p->precision);
    bool const user_double = ('z' == lower(precision));
    // 'z' plans take double data, every other plan float data (tfqmrgpu.cu:538-542); a mixed-precision plan takes either: its copies of
    // A, B and X are double (A also float), the values are converted on the way
    if (!mixed && user_double != is_double) return err(TFQMRGPU_PRECISION_MISSMATCH, __LINE__ % 10000, precision);
    if (nullptr == values) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    hipStream_t const s = (hipStream_t)h->stream;
    DevPlan const d = mixed ? resolveZ(*p) : resolve(*p);
    Target const to{ (0 == which) ? d.A : (1 == which) ? d.B : d.x, d.dbl, d.ilv };
    Target const floatA{ p->buffer + p->wA.offset, false, p->ilv };              // mixed: the inner solves multiply with A in float
    uint32_t const* u2n = (2 == which) ? d.u2i : nullptr;
    auto const st = transfer_blocks(*p, s, is_get ? 1 : 0, user_double, to, values, u2n, nnzb, nR, nC, layout, trans, conj, nullptr,
                                    (mixed && 0 == which && !is_get) ? &floatA : nullptr);
    if (!st && 1 == which) p->haveB = true;
    return st;
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_setMatrix(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, char const var,
    void const* val, char const precision, int const ld, int const d2, char const trans, tfqmrgpuDataLayout_t const layout)
{
    (void)ld; (void)d2;   // ignored by the reference as well (tfqmrgpu.cu:615-616)
    return set_or_get(handle, plan, var, (void*)val, precision, trans, layout, false);
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_getMatrix(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, char const var,
    void* val, char const precision, int const ld, int const d2, char const trans, tfqmrgpuDataLayout_t const layout)
{
    (void)ld; (void)d2;
    if ('x' != lower(var)) return err(TFQMRGPU_UNDOCUMENTED_ERROR, __LINE__ % 10000, var);  // only X (tfqmrgpu.cu:635-643)
    return set_or_get(handle, plan, var, val, precision, trans, layout, true);
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_solve(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, double const threshold, int const maxIterations) {
    auto p = asPlan(plan); auto h = (Handle*)handle;                    // tfqmrgpu.cu:648-661
    if (!p || !h) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    return run_solve(*h, *p, threshold, maxIterations);
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_getInfo(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan,
    double* residuum_reached, int32_t* iterations_needed, double* flops_performed, double* flops_performed_all)
{
    (void)handle;                                                        // tfqmrgpu.cu:663-679
    auto p = asPlan(plan);
    if (!p) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    int any = 0;
    if (residuum_reached)    { ++any; *residuum_reached = p->residuum_reached; }
    if (iterations_needed)   { ++any; *iterations_needed = p->iterations_needed; }
    if (flops_performed)     { ++any; *flops_performed = p->flops_performed; }
    if (flops_performed_all) { ++any; *flops_performed_all = p->flops_performed_all; }
    return any ? TFQMRGPU_STATUS_SUCCESS : TFQMRGPU_STATUS_NO_INFO_PASSED;
}

} // extern "C"

// one-call drivers, reference tfqmrgpu.cu:702-821
template <typename real_t>
static tfqmrgpuStatus_t bsrsv_onecall(int mb, int ldA, int ldB,
    int32_t const* rowPtrA, int nnzbA, int32_t const* colIndA, real_t const* Amat, char transA,
    int32_t const* rowPtrX, int nnzbX, int32_t const* colIndX, real_t* Xmat, char transX,
    int32_t const* rowPtrB, int nnzbB, int32_t const* colIndB, real_t const* Bmat, char transB,
    int32_t* iterations, float* residual, int indexOffset, int echo)
{
    char const zoc = (sizeof(real_t) == 8) ? 'z' : 'c';
    char const* const me = (zoc == 'z') ? "tfqmrgpu_bsrsv_z" : "tfqmrgpu_bsrsv_c";
    tfqmrgpuStatus_t stat;
    tfqmrgpuHandle_t handle = nullptr;
    tfqmrgpuBsrsvPlan_t plan = nullptr;
    void* buffer = nullptr;
    size_t bytes = 0;
    auto cleanup = [&]() {      // (the reference returns early and leaks on errors)
        if (buffer) tfqmrgpuDestroyWorkspace(buffer);
        if (plan) tfqmrgpu_bsrsv_destroyPlan(handle, plan);
        if (handle) tfqmrgpuDestroyHandle(handle);
    };
#define TFQ_STEP(call, name) stat = (call); if (stat) { if (echo > 0) std::printf("# %s: %s returned %d\n", me, name, stat); cleanup(); return stat; }
    if (echo > 0) std::printf("# %s: mb= %d, ldA= %d, ldB= %d, iterations= %d, residual= %.1e\n", me, mb, ldA, ldB,
                              iterations ? *iterations : 200, residual ? *residual : 1e-9);
    TFQ_STEP(tfqmrgpuCreateHandle(&handle), "tfqmrgpuCreateHandle")
    TFQ_STEP(tfqmrgpuSetStream(handle, 0), "tfqmrgpuSetStream")
    TFQ_STEP(tfqmrgpu_bsrsv_createPlan(handle, &plan, mb, rowPtrA, nnzbA, colIndA, rowPtrX, nnzbX, colIndX,
                                       rowPtrB, nnzbB, colIndB, indexOffset, echo), "tfqmrgpu_bsrsv_createPlan")
    TFQ_STEP(tfqmrgpu_bsrsv_bufferSize(handle, plan, ldA, ldA, ldB, ldB, zoc, &bytes), "tfqmrgpu_bsrsv_bufferSize")
    TFQ_STEP(tfqmrgpuCreateWorkspace(&buffer, bytes, 'd'), "tfqmrgpuCreateWorkspace")
    TFQ_STEP(tfqmrgpu_bsrsv_setBuffer(handle, plan, buffer), "tfqmrgpu_bsrsv_setBuffer")
    TFQ_STEP(tfqmrgpu_bsrsv_setMatrix(handle, plan, 'A', Amat, zoc, ldA, ldA, transA, TFQMRGPU_LAYOUT_RIRIRIRI), "tfqmrgpu_bsrsv_setMatrix('A')")
    TFQ_STEP(tfqmrgpu_bsrsv_setMatrix(handle, plan, 'B', Bmat, zoc, ldB, ldA, transB, TFQMRGPU_LAYOUT_RIRIRIRI), "tfqmrgpu_bsrsv_setMatrix('B')")
    double const threshold = residual ? *residual : 1e-9;
    int const maxiter = iterations ? *iterations : 200;
    TFQ_STEP(tfqmrgpu_bsrsv_solve(handle, plan, threshold, maxiter), "tfqmrgpu_bsrsv_solve")
    double residuum = 0, flops = 0, flops_all = 0; int32_t needed = 0;
    TFQ_STEP(tfqmrgpu_bsrsv_getInfo(handle, plan, &residuum, &needed, &flops, &flops_all), "tfqmrgpu_bsrsv_getInfo")
    if (echo > 1) std::printf("# tfQMRgpu needed %d iterations to converge to %.1e using %g GFlop\n", needed, residuum, flops * 1e-9);
    if (residual) *residual = float(residuum);
    if (iterations) *iterations = needed;
    TFQ_STEP(tfqmrgpu_bsrsv_getMatrix(handle, plan, 'X', Xmat, zoc, ldB, ldA, transX, TFQMRGPU_LAYOUT_RIRIRIRI), "tfqmrgpu_bsrsv_getMatrix")
#undef TFQ_STEP
    cleanup();
    return TFQMRGPU_STATUS_SUCCESS;
}

extern "C" {

tfqmrgpuStatus_t tfqmrgpu_bsrsv_z(int mb, int ldA, int ldB,
    int32_t const* rowPtrA, int nnzbA, int32_t const* colIndA, double const* Amat, char transA,
    int32_t const* rowPtrX, int nnzbX, int32_t const* colIndX, double* Xmat, char transX,
    int32_t const* rowPtrB, int nnzbB, int32_t const* colIndB, double const* Bmat, char transB,
    int32_t* iterations, float* residual, int indexOffset, int echo)
{
    return bsrsv_onecall<double>(mb, ldA, ldB, rowPtrA, nnzbA, colIndA, Amat, transA, rowPtrX, nnzbX, colIndX, Xmat, transX,
                                 rowPtrB, nnzbB, colIndB, Bmat, transB, iterations, residual, indexOffset, echo);
}

tfqmrgpuStatus_t tfqmrgpu_bsrsv_c(int mb, int ldA, int ldB,
    int32_t const* rowPtrA, int nnzbA, int32_t const* colIndA, float const* Amat, char transA,
    int32_t const* rowPtrX, int nnzbX, int32_t const* colIndX, float* Xmat, char transX,
    int32_t const* rowPtrB, int nnzbB, int32_t const* colIndB, float const* Bmat, char transB,
    int32_t* iterations, float* residual, int indexOffset, int echo)
{
    return bsrsv_onecall<float>(mb, ldA, ldB, rowPtrA, nnzbA, colIndA, Amat, transA, rowPtrX, nnzbX, colIndX, Xmat, transX,
                                rowPtrB, nnzbB, colIndB, Bmat, transB, iterations, residual, indexOffset, echo);
}

// ---- extensions (include/tfqmrgpu_ext.h) --------------------------------------------------------------

tfqmrgpuStatus_t tfqmrgpuExt_planView(tfqmrgpuBsrsvPlan_t plan, tfqmrgpuPlanView_t* v) {
    auto p = asPlan(plan);
    if (!p || !v) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    v->nRows = p->nRows; v->nCols = p->nCols; v->nnzbA = p->nnzbA; v->nnzbX = p->nnzbX; v->nnzbB = p->nnzbB;
    v->nPairs = p->nPairs();
    v->pairs = p->pairs.data(); v->starts = p->starts.data(); v->subset = p->subset.data();
    v->colindx = p->colindx.data(); v->original_bsrColIndX = p->original_bsrColIndX.data();
    v->LM = p->LM; v->LN = p->LN; v->precision = p->precision;
    return TFQMRGPU_STATUS_SUCCESS;
}

int32_t tfqmrgpuExt_getBoundHistory(tfqmrgpuBsrsvPlan_t plan, double* bound2, int32_t capacity) {
    auto p = asPlan(plan);
    if (!p) return -1;
    auto const n = int32_t(p->boundHistory.size());
    for (int32_t i = 0; i < std::min(n, capacity); ++i) if (bound2) bound2[i] = p->boundHistory[i];
    return n;
}

tfqmrgpuStatus_t tfqmrgpuExt_setProfiling(tfqmrgpuBsrsvPlan_t plan, int on) {
    auto p = asPlan(plan);
    if (!p) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    p->profiling = (on < 0 || on > 2) ? 1 : on;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_getProfile(tfqmrgpuBsrsvPlan_t plan, int64_t* launches, double* milliseconds) {
    auto p = asPlan(plan);
    if (!p || !launches || !milliseconds) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    for (int k = 0; k < TFQMRGPU_PROFILE_CLASSES; ++k) { launches[k] = p->profLaunches[k]; milliseconds[k] = p->profMs[k]; }
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_getProfileGated(tfqmrgpuBsrsvPlan_t plan, int64_t* launches, double* milliseconds) {
    auto p = asPlan(plan);
    if (!p || !launches || !milliseconds) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    for (int k = 0; k < TFQMRGPU_PROFILE_CLASSES; ++k) { launches[k] = p->profGatedLaunches[k]; milliseconds[k] = p->profGatedMs[k]; }
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_getMultiplyKernel(tfqmrgpuBsrsvPlan_t plan, char* name, int32_t capacity) {
    auto p = asPlan(plan);
    if (!p || !name || capacity < 1) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (!p->buffer) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    DevPlan d = resolve(*p);
    d.fold = (p->foldOk && !p->opFn) ? 1 : 0;     // (one rank: what run_tfqmr decides)
    std::snprintf(name, size_t(capacity), "%s", spmm_kernel_family(d));
    return TFQMRGPU_STATUS_SUCCESS;
}
tfqmrgpuStatus_t tfqmrgpuExt_getProfileFirst(tfqmrgpuBsrsvPlan_t plan, int64_t* launches, double* milliseconds) {
    auto p = asPlan(plan);
    if (!p || !launches || !milliseconds) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    for (int k = 0; k < TFQMRGPU_PROFILE_CLASSES; ++k) { launches[k] = p->profFirstLaunches[k]; milliseconds[k] = p->profFirstMs[k]; }
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_setThreeProductMultiply(tfqmrgpuBsrsvPlan_t plan, int on) {
    auto p = asPlan(plan);
    if (!p) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    p->threeProducts = (0 != on); p->selfStale = true;
    return TFQMRGPU_STATUS_SUCCESS;
}

int32_t tfqmrgpuExt_getRefinementHistory(tfqmrgpuBsrsvPlan_t plan, double* residual, int32_t* iterations, int32_t capacity) {
    auto p = asPlan(plan);
    if (!p) return -1;
    auto const n = int32_t(p->cycleResidual.size());
    for (int32_t i = 0; i < std::min(n, capacity); ++i) {
        if (residual) residual[i] = p->cycleResidual[i];
        if (iterations) iterations[i] = (size_t(i) < p->cycleIterations.size()) ? p->cycleIterations[i] : 0;   // the last entry has no solve behind it
    }
    return n;
}

tfqmrgpuStatus_t tfqmrgpuExt_setShadowMode(tfqmrgpuBsrsvPlan_t plan, int mode) {
    auto p = asPlan(plan);
    if (!p) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (mode != TFQMRGPU_SHADOW_HASH && mode != TFQMRGPU_SHADOW_GLIBC_RAND) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    p->shadowMode = mode;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_setShadowVector(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, float const* v3) {
    auto p = asPlan(plan); auto h = (Handle*)handle;
    if (!p || !h || !v3 || !p->buffer) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    DevPlan const d = resolve(*p);
    p->v3IsHash = false; p->selfStale = true;
    return transfer_blocks(*p, (hipStream_t)h->stream, 0, false, Target{d.v3, false, p->ilv}, (void*)v3, d.u2i, p->nnzbX, p->LM, p->LN,
                           TFQMRGPU_LAYOUT_RRRRIIII, false, false);
}

tfqmrgpuStatus_t tfqmrgpuExt_getShadowVector(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, float* v3) {
    auto p = asPlan(plan); auto h = (Handle*)handle;
    if (!p || !h || !v3 || !p->buffer) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    DevPlan const d = resolve(*p);
    return transfer_blocks(*p, (hipStream_t)h->stream, 1, false, Target{d.v3, false, p->ilv}, (void*)v3, d.u2i, p->nnzbX, p->LM, p->LN,
                           TFQMRGPU_LAYOUT_RRRRIIII, false, false);
}

tfqmrgpuStatus_t tfqmrgpuExt_getWorkVector(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, int which, void* values) {
    auto p = asPlan(plan); auto h = (Handle*)handle;
    if (!p || !h || !values || !p->buffer) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if ('z' != p->precision && 'c' != p->precision) return err(TFQMRGPU_PRECISION_MISSMATCH, __LINE__ % 10000, p->precision);
    DevPlan const d = resolve(*p);
    void* v = nullptr;
    switch (which) {
        case 1: v = d.x; break;  case 4: v = d.v4; break;  case 5: v = d.v5; break;  case 6: v = d.v6; break;
        case 7: v = d.v7; break; case 8: v = d.v8; break;  case 9: v = d.v9; break;
        default: return err(TFQMRGPU_VARIABLENAME_UNKNOWN, __LINE__ % 10000, char('0' + (which & 7)));
    }
    // the usual staging area IS the work vectors: this getter brings its own
    Stage st{nullptr, std::min<size_t>(p->S, size_t(64) << 20)};
    st.bytes = std::max(st.bytes, size_t(2) * p->LM * p->LN * ('z' == p->precision ? 8 : 4));
    if (hipSuccess != hipMalloc((void**)&st.ptr, st.bytes)) return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED);
    auto const status = transfer_blocks(*p, (hipStream_t)h->stream, 1, 'z' == p->precision, Target{v, 'z' == p->precision, p->ilv}, values, d.u2i, p->nnzbX, p->LM, p->LN,
                                        TFQMRGPU_LAYOUT_RRRRIIII, false, false, &st);
    (void)hipFree(st.ptr);
    return status;
}

// ---- prepared launch order of the stand-alone multiply (tfq_order.cpp) -----------------------------------------------------------
namespace { struct MultiplyOrder { uint32_t magic = 0x6f726472u; uint32_t nY = 0; uint32_t* perm = nullptr; }; }

tfqmrgpuStatus_t tfqmrgpuExt_multiplyPrepare(tfqmrgpuHandle_t handle, char precision, int lm, int ln, uint32_t nnzbY,
    uint32_t const* starts_d, uint32_t const* pairs_d, int mode, void** order)
{
    auto h = (Handle*)handle;
    if (!h || !starts_d || !pairs_d || !order) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    *order = nullptr;
    if (!blockSizeAllowed(lm, ln)) return err(TFQMRGPU_BLOCKSIZE_MISSING, ln, lm);
    uint32_t const ch = multiply_blocks_per_work_group(precision, lm, ln);
    if (0 == ch || 0 == nnzbY || mode <= 0) return TFQMRGPU_STATUS_SUCCESS;      // nothing to prepare: a null order is the caller's order
    hipStream_t const s = (hipStream_t)h->stream;
    try {
        std::vector<uint32_t> starts(size_t(nnzbY) + 1);
        TFQ_HIP(hipMemcpyAsync(starts.data(), starts_d, starts.size() * 4, hipMemcpyDeviceToHost, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        std::vector<uint32_t> pairs(size_t(starts[nnzbY]) * 2);
        TFQ_HIP(hipMemcpyAsync(pairs.data(), pairs_d, pairs.size() * 4, hipMemcpyDeviceToHost, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        TFQ_HIP(hipStreamSynchronize(s), TFQMRGPU_STATUS_LAUNCH_FAILED)
        for (uint32_t y = 0; y < nnzbY; ++y) if (starts[y + 1] < starts[y]) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
        auto const perm = multiply_order(nnzbY, starts.data(), pairs.data(), ch, mode, uint32_t(lm * lm), uint32_t(lm * ln));
        auto* o = new MultiplyOrder;
        o->nY = nnzbY;
        if (hipSuccess != hipMalloc((void**)&o->perm, size_t(nnzbY) * 4)) { delete o; return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED); }
        if (hipSuccess != hipMemcpyAsync(o->perm, perm.data(), size_t(nnzbY) * 4, hipMemcpyHostToDevice, s) || hipSuccess != hipStreamSynchronize(s)) {
            (void)hipFree(o->perm); delete o; return TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
        }
        *order = o;
    } catch (std::bad_alloc const&) { return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED); }
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_multiplyRelease(void* order) {
    auto o = (MultiplyOrder*)order;
    if (!o) return TFQMRGPU_STATUS_SUCCESS;
    if (o->magic != 0x6f726472u) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (o->perm) (void)hipFree(o->perm);
    o->magic = 0; delete o;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_multiplyOrdered(tfqmrgpuHandle_t handle, char precision, int lm, int ln,
    uint32_t nnzbY, uint32_t const* starts_d, uint32_t const* pairs_d, void const* A_d, void const* X_d, void* Y_d, void const* order)
{
    auto h = (Handle*)handle;
    if (!h || !starts_d || !pairs_d || !A_d || !X_d || !Y_d) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    auto o = (MultiplyOrder const*)order;
    if (o && (o->magic != 0x6f726472u || o->nY != nnzbY)) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);   // an order belongs to ONE listing
    return launch_multiply(precision, lm, ln, nnzbY, starts_d, pairs_d, A_d, X_d, Y_d, (hipStream_t)h->stream, o ? o->perm : nullptr);
}

tfqmrgpuStatus_t tfqmrgpuExt_multiply(tfqmrgpuHandle_t handle, char precision, int lm, int ln,
    uint32_t nnzbY, uint32_t const* starts_d, uint32_t const* pairs_d, void const* A_d, void const* X_d, void* Y_d)
{
    auto h = (Handle*)handle;
    if (!h || !starts_d || !pairs_d || !A_d || !X_d || !Y_d) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    return launch_multiply(precision, lm, ln, nnzbY, starts_d, pairs_d, A_d, X_d, Y_d, (hipStream_t)h->stream);
}

tfqmrgpuStatus_t tfqmrgpuExt_applyOperator(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, int repetitions) {
    auto p = asPlan(plan); auto h = (Handle*)handle;
    if (!p || !h || !p->buffer) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (p->opFn) return TFQ_ERR(TFQMRGPU_NO_IMPLEMENTATION);       // a user-defined operator is the caller's to apply
    hipStream_t const s = (hipStream_t)h->stream;
    bool const mixed = ('m' == p->precision);
    DevPlan const d = mixed ? resolveZ(*p) : resolve(*p);           // mixed: the double side (x, A), the product in v4 ... v7
    void* const y = mixed ? d.v4 : d.v9;                            // the work vectors are free outside of a solve
    for (int r = 0; r < std::max(1, std::abs(repetitions)); ++r) spmm_apply(d, d.x, y, s);
    // repetitions < 0: |repetitions| launches and nothing else (timing: X stays as it is, the product is left in the work vector)
    if (repetitions >= 0) TFQ_HIP(hipMemcpyAsync(d.x, y, mixed ? p->wXz.bytes : p->S, hipMemcpyDeviceToDevice, s), TFQMRGPU_STATUS_LAUNCH_FAILED)
    TFQ_HIP(hipGetLastError(), TFQMRGPU_STATUS_LAUNCH_FAILED)
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_commUniqueId(char id[128]) {
    if (!id) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (!g_rccl.load()) return TFQ_ERR(TFQMRGPU_NO_IMPLEMENTATION);
    return g_rccl.GetUniqueId(id) ? TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED) : TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_commInit(tfqmrgpuHandle_t handle, int nranks, int rank, char const id[128]) {
    auto h = (Handle*)handle;
    if (!h || !id) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (!g_rccl.load()) return TFQ_ERR(TFQMRGPU_NO_IMPLEMENTATION);
    UidByValue u; std::memcpy(u.internal, id, 128);
    void* comm = nullptr;
    if (g_rccl.CommInitRank(&comm, nranks, u, rank)) return TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
    h->comm = comm; h->nranks = nranks; h->rank = rank;
    if (!h->voteBuf) TFQ_HIP(hipMalloc((void**)&h->voteBuf, 4 * sizeof(double)), TFQMRGPU_STATUS_ALLOCATION_FAILED)   // for the vote in front of every solve
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_commDestroy(tfqmrgpuHandle_t handle) {
    auto h = (Handle*)handle;
    if (!h) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->nranks = 1; h->rank = 0;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_setOperator(tfqmrgpuBsrsvPlan_t plan, tfqmrgpuOperator_t multiply, void* ctx) {
    auto p = asPlan(plan);
    if (!p) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    p->opFn = (void*)multiply; p->opCtx = ctx;
    return TFQMRGPU_STATUS_SUCCESS;
}

tfqmrgpuStatus_t tfqmrgpuExt_setReduceCallback(tfqmrgpuHandle_t handle, tfqmrgpuReduceMax_t fn, void* ctx) {
    auto h = (Handle*)handle;
    if (!h) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    h->reduceFn = fn; h->reduceCtx = ctx;
    if (fn && !h->voteBuf) TFQ_HIP(hipMalloc((void**)&h->voteBuf, 4 * sizeof(double)), TFQMRGPU_STATUS_ALLOCATION_FAILED)
    return TFQMRGPU_STATUS_SUCCESS;
}

} // extern "C"
