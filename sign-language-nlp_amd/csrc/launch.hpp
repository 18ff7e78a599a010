// launch.hpp -- one way to launch every kernel of the step, so that K independent fits can share ONE launch.
//
// The reference's only parallelism is many small independent fits (/root/reference/main.py:70-78: GridSearchCV farms
// (candidate x fold) fits out to workers).  One batch-50 fit cannot fill 256 CUs -- its decoder stages are 50-row
// kernels -- and fits on separate streams only reach 1.26x (the command path serialises small dispatches).  So fits of
// one shape advance in LOCKSTEP through one launch sequence:
//
//  * every kernel takes its arguments as ONE trivially-copyable pack `P` plus an optional device table:
//        kernel(P a0, const P* tab)   ->   args = tab ? tab[blockIdx.z] : a0
//    A plain launch passes (a0, nullptr): nothing changes.  A lockstep launch passes the table of the K fits' packs
//    and grid.z = K: block (x, y, z) does for fit z exactly what block (x, y) of that fit's own launch would do,
//    so every fit's results are bit-identical to its solo run.
//  * the grouped GEMM kernels already run a list of independent jobs per launch; a lockstep launch concatenates the
//    K fits' job lists (device-resident job table + a block -> job map).
//  * a Recorder (thread-local) turns the plan code into a launch PROGRAM: while it is installed, zlaunch() /
//    the group launchers append {kernel, grid, block, LDS, argument bytes} instead of launching.  The lockstep
//    driver (lockstep.hip) records each fit's step once, merges the K programs call site by call site, uploads the
//    tables once and from then on replays the merged program: ~one launch per call site for all K fits.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "common.hpp"

namespace slnlp {

// ------------------------------------------------------------------ argument pack (a POD tuple) ----
template <size_t I, class T>
struct PackLeaf {
    T v;
};
template <class Seq, class... T>
struct PackImpl;
template <size_t... I, class... T>
struct PackImpl<std::index_sequence<I...>, T...> : PackLeaf<I, T>... {};
template <class... T>
using Pack = PackImpl<std::index_sequence_for<T...>, T...>;

// Pointers that arrive through a device-resident table are generic ("flat") to the compiler; kernel-argument pointers
// are known to be global.  None of our kernel arguments ever points to LDS or scratch, so every pointer field is passed
// through an explicit global address-space cast: both paths then compile to global_load / global_store (flat accesses
// also tie vmcnt to lgkmcnt and defeat counted waits).  The empty asm keeps the compiler from folding the cast pair
// away (it does otherwise, and everything stays flat); "+s": kernel arguments are wave-uniform, they stay in SGPRs.
template <class T>
__device__ __forceinline__ T as_global(T v) { return v; }
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
    typedef __attribute__((address_space(1))) T* gp;
    gp g = (gp)p;
    asm("" : "+s"(g));
    return (T*)g;
}
__device__ __forceinline__ PlaneOut as_global(PlaneOut po) {
    po.hi = as_global(po.hi);
    po.lo = as_global(po.lo);
    po.q8 = as_global(po.q8);
    return po;
}

// the per-direction argument structs travel inside kernel argument packs: their pointers get the same global
// address-space treatment as bare pointer arguments (declared before PackOf so its calls see them)
__device__ __forceinline__ slnlp_rnn_cell_dir as_global(slnlp_rnn_cell_dir d) {
    d.xproj = as_global(d.xproj); d.hproj = as_global(d.hproj); d.h = as_global(d.h); d.c = as_global(d.c);
    d.hprev_save = as_global(d.hprev_save); d.cprev_save = as_global(d.cprev_save); d.acts = as_global(d.acts);
    d.hn_save = as_global(d.hn_save); d.out = as_global(d.out);
    return d;
}
__device__ __forceinline__ slnlp_rnn_cell_bwd_dir as_global(slnlp_rnn_cell_bwd_dir d) {
    d.dh_state = as_global(d.dh_state); d.dc_state = as_global(d.dc_state); d.dout = as_global(d.dout); d.acts = as_global(d.acts);
    d.cprev_save = as_global(d.cprev_save); d.hprev_save = as_global(d.hprev_save); d.hn_save = as_global(d.hn_save);
    d.dgx = as_global(d.dgx); d.dgh = as_global(d.dgh); d.carry = as_global(d.carry); d.dh_extra = as_global(d.dh_extra);
    return d;
}

template <class F>
struct PackOf;
template <class... T>
struct PackOf<void (*)(T...)> {
    using type = Pack<T...>;
    using seq = std::index_sequence_for<T...>;
    template <void (*Body)(T...), size_t... I>
    __device__ __forceinline__ static void call(const type& p, std::index_sequence<I...>) {
        Body(as_global(static_cast<const PackLeaf<I, T>&>(p).v)...);
    }
};

// SLNLP_ZKERNEL(name, THREADS, body): the __global__ entry `name` around the __device__ function `body`.
#define SLNLP_ZKERNEL(name, THREADS, body)                                                                          \
    __global__ __launch_bounds__(THREADS) void name(typename ::slnlp::PackOf<decltype(&body)>::type a0,              \
                                                    const typename ::slnlp::PackOf<decltype(&body)>::type* tab) {    \
        using PO = ::slnlp::PackOf<decltype(&body)>;                                                                 \
        typename PO::type a;                                                                                         \
        if (tab) a = tab[blockIdx.z];                                                                                \
        else a = a0;                                                                                                 \
        ::slnlp::probe_kernel_begin();                                                                               \
        PO::template call<&body>(a, typename PO::seq{});                                                             \
        ::slnlp::probe_kernel_end();                                                                                 \
    }

// ------------------------------------------------------------------ recording ----
enum RecKind { REC_Z = 0, REC_PLANE_GROUP = 1, REC_GEMM_GROUP = 2 };

struct RecOp {
    const void* fn = nullptr;
    dim3 grid, block;
    size_t lds = 0;
    int kind = REC_Z;
    std::vector<char> args;        // REC_Z: the pack; group kinds: the by-value group parameter struct
    const char* what = "";
};

struct Recorder {
    std::vector<RecOp> ops;
};

Recorder* current_recorder();                 // nullptr: launches go to the stream
void set_recorder(Recorder* r);               // thread-local
inline bool recording() { return current_recorder() != nullptr; }

int record_op(const void* fn, dim3 grid, dim3 block, size_t lds, int kind, const void* args, size_t bytes, const char* what);

template <class P>
struct PackMaker;
template <size_t... I, class... T>
struct PackMaker<PackImpl<std::index_sequence<I...>, T...>> {
    template <class... A>
    static PackImpl<std::index_sequence<I...>, T...> make(A&&... a) {
        static_assert(sizeof...(A) == sizeof...(T), "zlaunch: argument count does not match the kernel's parameter list");
        PackImpl<std::index_sequence<I...>, T...> p{};
        ((static_cast<PackLeaf<I, T>&>(p).v = static_cast<T>(std::forward<A>(a))), ...);
        return p;
    }
};
template <class P, class... A>
P make_pack_from(A&&... a) {
    return PackMaker<P>::make(std::forward<A>(a)...);
}

template <class P, class... A>
int zlaunch(void (*kern)(P, const P*), dim3 grid, int threads, size_t lds, hipStream_t st, const char* what, A&&... a) {
    static_assert(std::is_trivially_copyable<P>::value, "kernel argument packs must be trivially copyable");
    // deduce the body's parameter types from the pack: P is PackImpl<seq, T...>
    P p = make_pack_from<P>(std::forward<A>(a)...);
    if (recording()) return record_op((const void*)kern, grid, dim3(threads), lds, REC_Z, &p, sizeof(P), what);
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, p, (const P*)nullptr);
    SLNLP_CHECK_LAUNCH(what);
    return 0;
}

// What a destroy call does about work still in flight.  Default: hipDeviceSynchronize() -- the plan's buffers belong to the
// caller, who may free them next.  A plan (or lockstep group) whose owner switched it off -- slnlp_{tf,rnn}[_lockstep]_
// set_destroy_sync(handle, 0), per object, never process-wide -- skips the wait: for callers whose buffers come from a
// STREAM-ORDERED allocator on the stream the plan ran on (torch's caching allocator: a freed block is only handed to later
// work of that stream), where the device-wide wait would stall every other host thread's queued work behind each plan that
// goes away.
void destroy_sync(int plan_wants_sync);

// ONE kernel sequence per device, by default.  Round 2 measured that fits on two hardware queues changed each other's results;
// round 3 found the cause (tools/probes/PACKED_FP32_REPORT.md): packed fp32 VALU instructions with an op_sel
// half-select compute wrongly in lanes 48-63 while a workgroup of ANOTHER kernel runs MFMA on the same CU -- a property of
// generated code, not of these sources; this library is built without them (Makefile, checked by disassembly in tests/).  The
// ordering stays as the library's default because it is free with one stream and gives a C-API caller that brings several
// streams solo bits even from a library someone rebuilt with other flags: every top-level step entry point opens a StepScope, which
//   * holds a per-device mutex while the step's launches are enqueued (host threads take turns, whole steps at a time),
//   * when the previous step of this device was enqueued on ANOTHER stream, records an event at that stream's tail and makes
//     this step's stream wait for it (the common case -- one shared stream -- costs a mutex and nothing on the GPU).
// Scopes nest (slnlp_tf_train_step -> slnlp_tf_forward ...): only the outermost acts.  Skipped while a launch recorder is
// installed (nothing is launched) and on a capturing stream.  Opting out is scoped to the HOST THREAD that owns a stream of its
// own (slnlp_set_thread_stream_policy(0): the grid search's worker threads); slnlp_set_stream_policy(0) is the process-wide
// switch (probes).  Still required whatever the policy: nothing -- DPP reductions instead of ds_bpermute and the StepScope itself
// are defensive; the build without packed fp32 is what the results depend on.
struct StepScope {
    explicit StepScope(hipStream_t st);
    ~StepScope();
    StepScope(const StepScope&) = delete;
    StepScope& operator=(const StepScope&) = delete;
    int rc = 0;
  private:
    bool outer_ = false;
    int dev_ = 0;
    hipStream_t st_ = nullptr;
};

// the fused Adam update's constants, when a lockstep group trains with Adam instead of SGD-momentum (lockstep.hip)
struct LsAdam { float beta1, beta2, eps, weight_decay; };

// zero `bytes` (a multiple of 16, 16-B aligned) with a kernel of ours: recordable, unlike hipMemsetAsync
int fill_zero(void* p, size_t bytes, hipStream_t st);

// Launch timer (slnlp_launch_timer_start / _stop): while it runs, every plane-GEMM group launch issued as a plain launch (not
// recorded, not under graph capture) is bracketed by two HIP events on its stream, so bench.py can time the dominant kernel
// where it lives -- between the other kernels of a train step, caches as the step leaves them -- not only back to back.
int launch_timer_begin(hipStream_t st);            // -> record index, or -1 when the timer is off / full / the stream is capturing
void launch_timer_end(int rec, hipStream_t st, int blocks, int njobs, int geo);

// split-bf16 passes of the plane GEMM's gradient products (2: the dY operand enters with its bf16 head only; 3: full split) --
// the process-wide DEFAULT FOR NEW PLANS: a plan copies it at creation and never looks again (slnlp_set_backward_passes, launch.hip)
int wgrad_passes();
int dgrad_passes();

}  // namespace slnlp
