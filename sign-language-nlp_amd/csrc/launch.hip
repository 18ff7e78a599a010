// launch.hip -- the thread-local launch recorder behind zlaunch() and a recordable zero-fill (see launch.hpp).
#include <atomic>
#include <mutex>
#include <unordered_map>

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

static std::atomic<int> g_destroy_sync{1};
void destroy_sync() {
    if (g_destroy_sync.load(std::memory_order_relaxed)) (void)hipDeviceSynchronize();
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

extern "C" int slnlp_set_destroy_sync(int on) {
    slnlp::g_destroy_sync.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}
