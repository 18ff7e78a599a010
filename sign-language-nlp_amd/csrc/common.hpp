// Shared device/host helpers for the slnlp gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/slnlp.h"

namespace slnlp {

// ---------------------------------------------------------------- errors ----
// Thread-local last-error string; entry points return a non-zero code and
// never abort ("returns a code -- never aborts").
void set_error(const char* fmt, ...);

#define SLNLP_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            ::slnlp::set_error(__VA_ARGS__);       \
            return SLNLP_ERR_INVALID_ARG;          \
        }                                          \
    } while (0)

#define SLNLP_CHECK_LAUNCH(what)                                              \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) {                                              \
            ::slnlp::set_error("%s: %s", what, hipGetErrorString(e__));       \
            return SLNLP_ERR_LAUNCH;                                          \
        }                                                                     \
    } while (0)

#define SLNLP_TRY(expr)            \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != 0) return rc__; \
    } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// One-time initialisation per device (hipFuncSetAttribute applies per device) that the fits_per_gpu host threads of a rank
// may all reach at once: the first caller runs `init` under a mutex, everyone else waits for it or takes the lock-free path.
struct DeviceOnce {
    std::mutex mu;
    std::atomic<bool> done[64];
    DeviceOnce() { for (auto& d : done) d.store(false, std::memory_order_relaxed); }
    template <class F>
    int run(F&& init) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (done[dev].load(std::memory_order_acquire)) return 0;
        std::lock_guard<std::mutex> lk(mu);
        if (done[dev].load(std::memory_order_relaxed)) return 0;
        const int rc = init();
        if (rc == 0) done[dev].store(true, std::memory_order_release);
        return rc;
    }
};

// ------------------------------------------------- probe builds (tools/probes) ----
// -DSLNLP_PROBE_FENCES=k builds a library whose kernels bracket themselves with agent-scope fences: bit 0 = every wave starts
// with an acquire (invalidates its CU's L1 and the XCD L2's non-local lines), bit 1 = every wave ends with a release (writes the
// XCD L2's dirty lines back).  Never the shipped build (k = 0: both helpers compile to nothing): it exists to tell whether the
// multi-queue nondeterminism is a cache-maintenance gap at kernel boundaries.
#ifndef SLNLP_PROBE_FENCES
#define SLNLP_PROBE_FENCES 0
#endif
__device__ __forceinline__ void probe_kernel_begin() {
#if SLNLP_PROBE_FENCES & 1
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void probe_kernel_end() {
#if SLNLP_PROBE_FENCES & 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ------------------------------------------------------------- dropout ------
// Counter-based Threefry4x32-12 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- the
// 12-round form is the shortest Threefry-4x32 the paper reports as passing BigCrush; known-answer vectors of the 20-round form
// pin the numpy restatement in tests/test_dropout_cpu.py, which the GPU masks are compared with bit for bit).  A dropout site
// is a logical [R, C] tensor.  One call yields 128 bits = EIGHT 16-bit lots and serves the 4 x 2 block of elements
// {rows 4 r4 .. 4 r4 + 3} x {columns c, c + 16} (bit 4 of c clear):
//     element (r, c):  counter = {cc, r >> 2, 0, 0},  cc = c without its bit 4 = ((c >> 5) << 4) | (c & 15)
//                      key     = {seed + phi * step (low, high word), site, 0}
//                      lot     = 16-bit field 4 * (c >> 4 & 1) + (r & 3) of the 128 bits (field f = half f & 1 of word f >> 1)
//                      kept iff lot >= thr16,  thr16 = round(p * 65536):  P[drop] = thr16 / 65536, within 2^-17 of p
// -- the 4 rows of a 16 x 16 MFMA accumulator column AND the same rows of the column tile next to it, which the lane that owns
// the first also owns in every MFMA kernel here (two column tiles per wave at least), so a call is spent on eight elements.
// History (rounds 1-4): Philox4x32-10, 32-bit lots, four elements per call -- 40 quarter-rate 32-bit multiplies per call; the
// dropout epilogue was a third of a forward plane-GEMM workgroup's life in a lockstep step (profiles/r04_lockstep_plane_timeline.txt).
// Threefry is adds, rotates and xors only (88 full-rate instructions per call).  The kept values are scaled by 1 / (1 - p)
// exactly as nn.Dropout does (the reference: positional_encoding.py:25,49, nn.Transformer's dropout at transformer.py:40-45);
// nothing pins the reference's stream (CPU mt19937 there).  Masks are never stored: the backward kernels regenerate them from
// (seed, step, site, r, c); slnlp_dropout_mask materialises any site's mask.
struct RngState {
    unsigned long long seed;
    unsigned long long step;
};

// the Threefry key schedule of (seed, step, site): ONE 16-byte load of the device-resident {seed, step} and a handful of scalar
// operations.  Kernels that draw in a loop make it once in front of the loop: written inside, the load sits in every iteration
// with a full wait behind it (the loop's LDS stores may alias the state as far as the compiler knows).
struct DropKey { unsigned ks[5]; };
__device__ __forceinline__ DropKey dropout_key(const unsigned long long* rng, int site) {
    const unsigned long long k = rng[0] + 0x9E3779B97F4A7C15ull * rng[1];
    DropKey K;
    K.ks[0] = (unsigned)k; K.ks[1] = (unsigned)(k >> 32); K.ks[2] = (unsigned)site; K.ks[3] = 0u;
    K.ks[4] = 0x1BD11BDAu ^ K.ks[0] ^ K.ks[1] ^ K.ks[2] ^ K.ks[3];
    return K;
}

#ifndef SLNLP_THREEFRY_ROUNDS
#define SLNLP_THREEFRY_ROUNDS 12
#endif
__device__ __forceinline__ uint4 threefry4x32(unsigned c0, unsigned c1, const DropKey& K) {
    constexpr int ROT[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
    unsigned x0 = c0 + K.ks[0], x1 = c1 + K.ks[1], x2 = K.ks[2], x3 = K.ks[3];      // counter words 2, 3 are zero
#pragma unroll
    for (int r = 0; r < SLNLP_THREEFRY_ROUNDS; ++r) {
        if ((r & 1) == 0) {
            x0 += x1; x1 = __builtin_rotateleft32(x1, ROT[r & 7][0]) ^ x0;
            x2 += x3; x3 = __builtin_rotateleft32(x3, ROT[r & 7][1]) ^ x2;
        } else {
            x0 += x3; x3 = __builtin_rotateleft32(x3, ROT[r & 7][0]) ^ x0;
            x2 += x1; x1 = __builtin_rotateleft32(x1, ROT[r & 7][1]) ^ x2;
        }
        if ((r & 3) == 3) {
            const int s = (r + 1) >> 2;
            x0 += K.ks[s % 5]; x1 += K.ks[(s + 1) % 5]; x2 += K.ks[(s + 2) % 5]; x3 += K.ks[(s + 3) % 5] + (unsigned)s;
        }
    }
    return make_uint4(x0, x1, x2, x3);
}

// the call's column coordinate of column c (bit 4 of c removed) and which half of the call's lots c takes (bit 4 of c)
__device__ __forceinline__ unsigned drop_cc(unsigned c) { return ((c >> 5) << 4) | (c & 15u); }
__device__ __forceinline__ int drop_half(unsigned c) { return (int)((c >> 4) & 1u); }

// the eight lots of rows 4 r4 .. 4 r4 + 3 x columns {c, c + 16} with cc = drop_cc(c)
__device__ __forceinline__ uint4 dropout_bits8(const DropKey& key, unsigned r4, unsigned cc) { return threefry4x32(cc, r4, key); }

// keep-threshold: keep iff lot >= thr  (P[drop] = thr / 2^16)
__host__ __device__ __forceinline__ unsigned dropout_threshold(float p) {
    const double t = (double)p * 65536.0 + 0.5;
    return t >= 65535.0 ? 65535u : (unsigned)t;
}

// lot of (row r & 3, column half `half`): field 4 * half + r.  Written as a 64-bit shift on purpose: a chain of selects over the
// vector's elements is turned into a dynamically indexed extract by the optimiser, which the backend then serves from scratch or
// LDS (every kernel that called it with a runtime row grew a private segment and ran a fifth slower); with static arguments the
// shift folds into one v_lshrrev / v_and.
__device__ __forceinline__ unsigned pick_lot(const uint4& v, int half, int r) {
    const unsigned long long lo = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
    const unsigned long long hi = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
    return (unsigned)((half ? hi : lo) >> (16 * (r & 3))) & 0xFFFFu;
}

// single-element form (non-MFMA kernels)
__device__ __forceinline__ bool dropout_keep(const DropKey& key, unsigned r, unsigned c, unsigned thr) {
    const uint4 b = dropout_bits8(key, r >> 2, drop_cc(c));
    return pick_lot(b, drop_half(c), (int)(r & 3u)) >= thr;
}
__device__ __forceinline__ bool dropout_keep(const unsigned long long* rng, int site,
                                             unsigned r, unsigned c, unsigned thr) {
    return dropout_keep(dropout_key(rng, site), r, c, thr);
}

// ------------------------------------------------------- bf16 plane sinks ----
// Producers of GEMM operands also emit the value as bf16 hi / lo planes (same row stride as the fp32
// tensor) so the consuming GEMM can stage them by LDS-DMA with no conversion (gemm_planes.hip).
// hi = truncated upper half of the fp32 word, lo = rne_bf16(x - hi): hi + lo == x to ~2^-16.
struct PlaneOut {
    unsigned short* hi = nullptr;
    unsigned short* lo = nullptr;
    unsigned char* q8 = nullptr;   // precision 8: the value as OCP e4m3 (scale 1), operand of the fp8 forward GEMMs
};
// fp32 -> OCP e4m3fn (gfx950's fp8), saturating at +-448 (a NaN stays a NaN)
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
// x = hi + lo, hi = bf16(x) ROUNDED TO NEAREST (v_cvt_pk_bf16_f32; a NaN stays a NaN), lo = bf16(x - hi): |lo| <= 2^-9 |x|, and the head
// alone is an unbiased bf16 image of x -- what the two-pass gradient products (gemm_planes.hip, precision 2) contract with.
// (Rounds 1-3 truncated: hi = the top 16 bits, |lo| <= 2^-8 |x|, and a head that is 0.2 % short on average.)
__device__ __forceinline__ void split_bf16(float x, unsigned short& h, unsigned short& l) {
    const __bf16 hb = (__bf16)x;
    h = __builtin_bit_cast(unsigned short, hb);
    const __bf16 b = (__bf16)(x - (float)hb);
    l = __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned short head_bf16(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ void store_planes1(const PlaneOut& po, long idx, float v) {
    if (!po.hi) return;
    unsigned short h, l;
    split_bf16(v, h, l);
    po.hi[idx] = h;
    po.lo[idx] = l;
    if (po.q8) po.q8[idx] = (unsigned char)(pack_fp8x4(v, 0.f, 0.f, 0.f) & 0xFFu);
}
__device__ __forceinline__ void store_planes4(const PlaneOut& po, long idx, float4 v) {   // idx % 4 == 0
    if (!po.hi) return;
    unsigned short h[4], l[4];
    split_bf16(v.x, h[0], l[0]); split_bf16(v.y, h[1], l[1]); split_bf16(v.z, h[2], l[2]); split_bf16(v.w, h[3], l[3]);
    uint2 w;
    w.x = h[0] | ((unsigned)h[1] << 16); w.y = h[2] | ((unsigned)h[3] << 16);
    *reinterpret_cast<uint2*>(po.hi + idx) = w;
    w.x = l[0] | ((unsigned)l[1] << 16); w.y = l[2] | ((unsigned)l[3] << 16);
    *reinterpret_cast<uint2*>(po.lo + idx) = w;
    if (po.q8) *reinterpret_cast<unsigned*>(po.q8 + idx) = pack_fp8x4(v.x, v.y, v.z, v.w);
}

// ------------------------------------------------------------- wave ops -----
// Cross-lane reductions WITHOUT the LDS crossbar: DPP row operations inside a row of 16 lanes and v_readlane across the four
// rows -- VALU only (`__shfl_xor` compiles to ds_bpermute_b32 on gfx950, an LDS-unit instruction).  History: round 3 suspected
// ds_bpermute of the multi-queue nondeterminism and moved every reduction here; that changed nothing -- the cause turned out to
// be packed fp32 VALU instructions (the library is built without them, Makefile).  The DPP forms stay
// because they are no slower and keep reductions off the LDS pipe the GEMM tiles load; they are not a required mitigation.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {      // every lane has a valid source for the controls used here
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;          // quad_perm [1,0,3,2] / [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;   // lane i <-> 7 - i within 8 / 15 - i within 16
__device__ __forceinline__ float lane_bcast(float v, int lane) {   // `lane` wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// sum / max over the 16 lanes of a DPP row (lanes 16k .. 16k+15); every lane of the row gets the result
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);      // quads hold one value each: mirroring inside 8 lanes swaps the two quads
    v += dpp_mov<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
    v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
    return v;
}
// over the whole wave (all 64 lanes active); every lane gets the result
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

}  // namespace slnlp

// ------------------------------------------------- internal kernel launchers
namespace slnlp {
int gemm(const slnlp_gemm_args& a, hipStream_t s);
int gemm_group(const slnlp_gemm_args* jobs, int njobs, hipStream_t s, unsigned wide_mask = 0);   // fp32-operand jobs, one launch (gemm.hip)
int gemm_planes(const slnlp_gemm_args& a, hipStream_t s);
int gemm_rows(const slnlp_gemm_args& a, hipStream_t s);
int gemm_rows_bwd(const slnlp_gemm_args& dgrad, const slnlp_gemm_args& wgrad, hipStream_t s);   // dX = dY W and dW = dY^T x, db in one launch     // B-row products on k-major planes, register-direct (gemm_rows.hip)
int gemm_planes_init();
// up to 4 independent plane GEMMs in ONE launch, optional deterministic split-K per job (gemm_planes.hip)
int gemm_planes_group(const slnlp_gemm_args* jobs, const int* split_k, int njobs, void* scratch, size_t scratch_bytes,
                      hipStream_t s, bool defer_reduce = false);   // defer_reduce: a lone split-K job's slices meet in a second launch
size_t gemm_group_scratch_bytes(const slnlp_gemm_args* jobs, const int* split_k, int njobs);
// the gradient pair of one dY over plane operands: dW = dY^T x (split-K) and dX = dY W -- one grouped launch, or a launch each when
// both are large (gemm_planes.hip: gemm_planes_wd_plan holds the rule; split <= WD_MAX_SPLITK)
constexpr int WD_MAX_SPLITK = 8;
struct WdPlan { int split, separate; };
WdPlan gemm_planes_wd_plan(const slnlp_gemm_args& wgrad, const slnlp_gemm_args& dgrad);
int gemm_planes_wd(const slnlp_gemm_args& wgrad, const slnlp_gemm_args& dgrad, void* scratch, size_t scratch_bytes, hipStream_t s);
struct QuantRow { long off; int K, pad; };   // one weight row of a precision-8 plan: offset into the arena (floats), length
int quant_rows_fp8(const float* x, int64_t ld, int R, int K, unsigned char* q, int64_t ldq, float* scale, const void* row_table,
                   hipStream_t st);   // row_table: device array of {long offset (floats); int K; int pad} or null
int split_planes(const float* x, int64_t ld, int R, int C, unsigned short* hi, unsigned short* lo, int64_t ldp, hipStream_t st);
int embed_fwd(const int64_t* ids, int64_t ld_ids, int B, int S, int E, int V, const float* table, const float* pe,
              float* out, float scale, float drop_p, int drop_site, const unsigned long long* rng, int64_t nan_idx,
              hipStream_t st, PlaneOut po = {}, unsigned char* keep_mask = nullptr);   // keep_mask: [B*S*E/4] bytes or null
int embed_bwd(const int64_t* ids, int64_t ld_ids, int B, int S, int E, int V, const float* dx, float* dtable,
              float scale, int64_t zero_row, float drop_p, int drop_site, const unsigned long long* rng, void* scratch,
              hipStream_t st, const unsigned char* keep_mask = nullptr);   // keep bits recorded by embed_fwd, or null: regenerate
size_t embed_bwd_scratch_bytes(int B, int S, int E);
int attn_self_fwd(const float* qkv, const int64_t* ids, int64_t ld_ids, int64_t pad_idx, int causal, int B, int S,
                  int H, int dh, float* ctx, float* probs, float drop_p, int drop_site,
                  const unsigned long long* rng, hipStream_t st, PlaneOut po = {});
int attn_self_bwd(const float* qkv, const float* probs, const float* dctx, int B, int S, int H, int dh, float* dqkv,
                  float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st, PlaneOut po = {},
                  float* long_scratch = nullptr);   // S > 64: attn_long_scratch_bytes(B, S, H) bytes
size_t attn_long_scratch_bytes(int B, int S, int H);
int attn_cross_fwd(const float* q, const float* kv, int64_t ld_kv, int B, int S, int H, int dh, float* ctx,
                   float* probs, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st);
int attn_cross_bwd(const float* q, const float* kv, int64_t ld_kv, const float* probs, const float* dctx, int B,
                   int S, int H, int dh, float* dq, float* dkv, int64_t ld_dkv, float drop_p, int drop_site,
                   const unsigned long long* rng, hipStream_t st, PlaneOut po = {});
// decoder cross-attention for a target of length 1 without K / V projections of the memory (attention_mem.hip); the
// per-head weight products around these run as batched GEMM jobs (tf_plan.hip)
int xmem_fwd(const float* qk, const float* mem, const float* bv, int B, int S, int H, int dh, float* mbar, float* psum, float* probs,
             float* ctx0, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st);
int xmem_bwd(const float* mem, const float* bv, const float* probs, const float* psum, const float* qk, const float* dmbar,
             const float* dctx, int B, int S, int H, int dh, float* dsc, float* dqk, float* dcp, float* dbv, float* dmem, int accumulate,
             float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st);
int layernorm_fwd(const float* x, const float* gamma, const float* beta, int rows, int E, float eps, float* y,
                  float* stats, hipStream_t st, PlaneOut po = {});
int ln_bwd_blocks(int rows);
bool ln_bwd_fused(int full_rows);     // the row kernel of a batch this tall also sums (dgamma, dbeta) per chunk (elementwise.hip)
int layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stats, int rows, int E,
                  const float* add_to_dx, float* dx, float* dx_drop, float drop_p, int drop_site,
                  const unsigned long long* rng, float* partial, int* nblk_out, int full_rows, hipStream_t st,
                  PlaneOut po_dx = {}, PlaneOut po_drop = {});
int attn_init();
int ln_param_reduce(const slnlp_ln_reduce_entry* table_dev, int n, int max_E, hipStream_t st);
// (dgamma, dbeta) chunk sums of n LayerNorms (elementwise.hip: ln_param_partial_kernel) from a device table, or of one (`single`,
// host memory; table == nullptr).  rows_enc / rows_dec = this batch's rows, full_* = the plan's full batch (chunk geometry).
struct LnPartialEntry {
    const float* dy;
    const float* x;
    const float* stats;
    float* partial;
    int dec;                // 1: a decoder LayerNorm (rows = rows_dec), 0: an encoder one (rows = rows_enc)
    int pad;
};
int ln_param_partial(const LnPartialEntry* table_dev, const LnPartialEntry* single_host, int n, int E, int rows_enc, int rows_dec,
                     int full_rows_enc, int full_rows_dec, hipStream_t st);
int ln_partial_chunk(int rows);
int lsm_nll(const float* logits, int64_t ld_logits, const int64_t* y, int B, int V, int64_t ignore_index, float* logp,
            float* loss, float* dlogits, int64_t ld_dlogits, float* row_scratch, hipStream_t st, hipStream_t loss_st,
            float* logp2 = nullptr, const int* logp2_row = nullptr, float* loss_hist = nullptr, const int* hist_idx = nullptr);
int lsm_bwd(const float* logp, const float* dlogp, int B, int V, float* dlogits, int64_t ld_dlogits, hipStream_t st);
int rnn_cell_fwd(int lstm, const slnlp_rnn_cell_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths, float fill,
                 int64_t ld_out, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st);
int rnn_step_fwd(int lstm, const slnlp_rnn_step_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths, float fill,
                 int64_t ld_out, float drop_p, int drop_site, const unsigned long long* rng, int precision, hipStream_t st);
int rnn_layer_init();
int rnn_layer_fwd(int lstm, const slnlp_rnn_layer_dir* dirs, int ndir, int B, int Hd, int S, const int64_t* lengths,
                  float fill, int64_t ld_out, float drop_p, int drop_site, const unsigned long long* rng, int precision,
                  unsigned* bar, int* err, int* launched, hipStream_t st);
int rnn_cell_bwd(int lstm, const slnlp_rnn_cell_bwd_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths,
                 int64_t ld_dout, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st);
bool rnn_step_bwd_covers(int B, int Hd);
int rnn_step_bwd_init();
int rnn_step_bwd(int lstm, const slnlp_rnn_step_bwd_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths, int64_t ld_dout,
                 float drop_p, int drop_site, const unsigned long long* rng, int precision, hipStream_t st);
int bahdanau_fwd(const float* q, const float* pk, const float* val, const float* we, const int64_t* ids,
                 int64_t ld_ids, int64_t pad, int B, int S, int Hd, float* alphas, float* ctx, hipStream_t st);
int bahdanau_bwd(const float* q, const float* pk, const float* val, const float* we, const float* alphas,
                 const float* dctx, int B, int S, int Hd, float* dq, float* dpk, float* dval, float* dwe_part,
                 float* dwe, hipStream_t st);
int add_rows(const float* in, int64_t ld_in, float* out, int64_t ld_out, int R, int C, int accumulate, hipStream_t st);
int tanh_bwd(const float* dy, const float* y, float* out, int64_t n, hipStream_t st);
int clip_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, const float* lr_dev,
                  float momentum, float max_norm, float* partials, float* norm_out, unsigned long long* rng,
                  hipStream_t st, PlaneOut wp = {}, int64_t wp_begin = 0, int64_t wp_end = -1);   // planes written for floats [wp_begin, wp_end); -1: to the end
int clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr_dev,
                   float beta1, float beta2, float eps, float weight_decay, float max_norm, float* partials, float* norm_out,
                   unsigned long long* rng, float* step_f, hipStream_t st, PlaneOut wp = {}, int64_t wp_begin = 0, int64_t wp_end = -1);
// Which version of a parameter arena a plan's derived data (bf16 weight planes) was made from: every optimizer step and
// every slnlp_*_params_changed() call moves the arena to a new generation (process-wide table keyed by the arena pointer,
// because several plans -- one per sequence length -- may share one arena).
unsigned long long params_generation(const float* params);
unsigned long long bump_params_generation(const float* params);
}  // namespace slnlp
