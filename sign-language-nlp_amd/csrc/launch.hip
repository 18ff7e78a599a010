// launch.hip -- the thread-local launch recorder behind zlaunch() and a recordable zero-fill (see launch.hpp).
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "launch.hpp"

namespace slnlp {

static std::mutex gen_mutex;
static std::unordered_map<const float*, unsigned long long> gen_table;

unsigned long long params_generation(const float* params) {
    std::lock_guard<std::mutex> lk(gen_mutex);
    auto it = gen_table.find(params);
    return it == gen_table.end() ? 0ull : it->second;
}
unsigned long long bump_params_generation(const float* params) {
    std::lock_guard<std::mutex> lk(gen_mutex);
    return ++gen_table[params];
}

void destroy_sync(int plan_wants_sync) {
    if (plan_wants_sync) (void)hipDeviceSynchronize();
}

// ---- one kernel sequence per device (launch.hpp, StepScope)
static std::atomic<int> g_stream_policy{1};
constexpr int MAX_DEV = 64;
struct DevSeq {
    std::recursive_mutex mu;            // recursive: a host thread's nested scopes
    hipStream_t last = nullptr;         // stream of the last step enqueued on this device
    bool any = false;
    hipEvent_t ev = nullptr;
};
static DevSeq g_seq[MAX_DEV];
static thread_local int tl_scope_depth = 0;
static thread_local int tl_stream_policy = -1;     // -1: the process-wide policy; 0 / 1: this host thread's own (slnlp_set_thread_stream_policy)

StepScope::StepScope(hipStream_t st) : st_(st) {
    const int policy = tl_stream_policy >= 0 ? tl_stream_policy : g_stream_policy.load(std::memory_order_relaxed);
    if (tl_scope_depth++ > 0 || !policy || recording()) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
    dev_ = dev;
    outer_ = true;
    DevSeq& q = g_seq[dev];
    q.mu.lock();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) return;               // a captured step is replayed later, under its own scope
    if (q.any && q.last != st) {
        // the tail of the other stream: everything enqueued there so far (nobody else enqueues: we hold the mutex)
        bool ok = q.ev || hipEventCreateWithFlags(&q.ev, hipEventDisableTiming) == hipSuccess;
        if (ok && hipEventRecord(q.ev, q.last) == hipSuccess) {
            if (hipStreamWaitEvent(st, q.ev, 0) != hipSuccess) {
                set_error("StepScope: cannot order stream %p behind stream %p: %s", (void*)st, (void*)q.last, hipGetErrorString(hipGetLastError()));
                rc = SLNLP_ERR_LAUNCH;
            }
        } else {
            (void)hipGetLastError();    // the other stream is gone (destroying it waited for its work): nothing to order against
        }
    }
}

StepScope::~StepScope() {
    --tl_scope_depth;
    if (!outer_) return;
    DevSeq& q = g_seq[dev_];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st_, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap == hipStreamCaptureStatusNone) { q.last = st_; q.any = true; }
    q.mu.unlock();
}

static thread_local Recorder* tl_recorder = nullptr;

Recorder* current_recorder() { return tl_recorder; }
void set_recorder(Recorder* r) { tl_recorder = r; }

int record_op(const void* fn, dim3 grid, dim3 block, size_t lds, int kind, const void* args, size_t bytes, const char* what) {
    Recorder* r = tl_recorder;
    SLNLP_CHECK_ARG(r, "record_op: no recorder installed");
    RecOp op;
    op.fn = fn;
    op.grid = grid;
    op.block = block;
    op.lds = lds;
    op.kind = kind;
    op.args.assign((const char*)args, (const char*)args + bytes);
    op.what = what;
    r->ops.push_back(std::move(op));
    return 0;
}

__device__ __forceinline__ void fill_zero_body(uint4* __restrict__ p, long n16) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
SLNLP_ZKERNEL(fill_zero_kernel, 256, fill_zero_body)

int fill_zero(void* p, size_t bytes, hipStream_t st) {
    SLNLP_CHECK_ARG(p && bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0, "fill_zero: needs a 16-byte aligned range");
    if (bytes == 0) return 0;
    const long n16 = (long)(bytes / 16);
    int grid = ceil_div(n16, 256);
    if (grid > 2048) grid = 2048;
    return zlaunch(fill_zero_kernel, dim3(grid), 256, 0, st, "fill_zero", (uint4*)p, n16);
}

}  // namespace slnlp

// ---- launch timer (launch.hpp)
namespace slnlp {
struct TimedLaunch { hipEvent_t e0 = nullptr, e1 = nullptr; int blocks = 0, njobs = 0, geo = 0; bool done = false; };
static std::mutex g_timer_mu;
static std::vector<TimedLaunch> g_timer_recs;
static std::atomic<int> g_timer_on{0};
static int g_timer_max = 0;

int launch_timer_begin(hipStream_t st) {
    if (!g_timer_on.load(std::memory_order_relaxed)) return -1;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (cap != hipStreamCaptureStatusNone) return -1;
    std::lock_guard<std::mutex> lk(g_timer_mu);
    if ((int)g_timer_recs.size() >= g_timer_max) return -1;
    TimedLaunch t;
    if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess || hipEventRecord(t.e0, st) != hipSuccess) {
        (void)hipGetLastError();
        if (t.e0) (void)hipEventDestroy(t.e0);
        if (t.e1) (void)hipEventDestroy(t.e1);
        return -1;
    }
    g_timer_recs.push_back(t);
    return (int)g_timer_recs.size() - 1;
}
void launch_timer_end(int rec, hipStream_t st, int blocks, int njobs, int geo) {
    std::lock_guard<std::mutex> lk(g_timer_mu);
    if (rec < 0 || rec >= (int)g_timer_recs.size()) return;
    TimedLaunch& t = g_timer_recs[rec];
    t.blocks = blocks; t.njobs = njobs; t.geo = geo;
    t.done = hipEventRecord(t.e1, st) == hipSuccess;
}
}  // namespace slnlp

extern "C" int slnlp_launch_timer_start(int max_records) {
    std::lock_guard<std::mutex> lk(slnlp::g_timer_mu);
    if (!slnlp::g_timer_recs.empty() || max_records < 1) {
        slnlp::set_error("launch_timer_start: %s", max_records < 1 ? "max_records < 1" : "a previous run was not read (slnlp_launch_timer_stop)");
        return SLNLP_ERR_INVALID_ARG;
    }
    slnlp::g_timer_max = max_records;
    slnlp::g_timer_recs.reserve(max_records);
    slnlp::g_timer_on.store(1, std::memory_order_relaxed);
    return 0;
}
extern "C" int slnlp_launch_timer_stop(slnlp_timed_launch* out, int max_out) {
    slnlp::g_timer_on.store(0, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(slnlp::g_timer_mu);
    int n = 0;
    for (auto& t : slnlp::g_timer_recs) {
        float ms = -1.f;
        if (t.done && hipEventSynchronize(t.e1) == hipSuccess && hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) { (void)hipGetLastError(); ms = -1.f; }
        if (out && n < max_out && ms >= 0.f) {
            out[n].blocks = t.blocks; out[n].njobs = t.njobs; out[n].geometry = t.geo; out[n].us = ms * 1e3f;
            ++n;
        }
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    slnlp::g_timer_recs.clear();
    return n;
}

// ---- how many split-bf16 passes the gradient products of the plane GEMM take (slnlp.h: slnlp_set_backward_passes)
namespace slnlp {
static int env_passes(const char* name, int dflt) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : dflt;
    return v == 2 || v == 3 ? v : dflt;
}
static std::atomic<int> g_wgrad_passes{env_passes("SLNLP_WGRAD_PASSES", 2)};
static std::atomic<int> g_dgrad_passes{env_passes("SLNLP_DGRAD_PASSES", 2)};
int wgrad_passes() { return g_wgrad_passes.load(std::memory_order_relaxed); }
int dgrad_passes() { return g_dgrad_passes.load(std::memory_order_relaxed); }
}  // namespace slnlp

extern "C" int slnlp_set_backward_passes(int wgrad, int dgrad) {
    if ((wgrad != 2 && wgrad != 3) || (dgrad != 2 && dgrad != 3)) {
        slnlp::set_error("set_backward_passes: wgrad = %d, dgrad = %d (each 2 or 3)", wgrad, dgrad);
        return SLNLP_ERR_INVALID_ARG;
    }
    slnlp::g_wgrad_passes.store(wgrad, std::memory_order_relaxed);
    slnlp::g_dgrad_passes.store(dgrad, std::memory_order_relaxed);
    return 0;
}
extern "C" int slnlp_get_backward_passes(int* wgrad, int* dgrad) {
    if (wgrad) *wgrad = slnlp::wgrad_passes();
    if (dgrad) *dgrad = slnlp::dgrad_passes();
    return 0;
}

extern "C" int slnlp_set_thread_stream_policy(int serialise) {
    slnlp::tl_stream_policy = serialise < 0 ? -1 : (serialise ? 1 : 0);
    return 0;
}

extern "C" int slnlp_set_stream_policy(int serialise) {
    slnlp::g_stream_policy.store(serialise ? 1 : 0, std::memory_order_relaxed);
    return 0;
}
