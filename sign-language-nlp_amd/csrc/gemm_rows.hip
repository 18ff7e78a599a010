// gemm_rows.hip -- the decoder's B-row products on PRE-SPLIT operands, register-direct.
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ),  A = Ahi + Alo, B = Bhi + Blo  (bf16 planes, both k-major)
//
// The reference decodes ONE target position (transformer.py:82-87: tgt of length 1), so every Linear of its decoder is a
// [batch rows x E] x [E x E]^T product -- 50 rows at the headline config -- on a dependent chain of ~150 launches per step:
// 2 % of the step's FLOPs, 29 % of its time (r04).  Such a launch lasts as long as ONE workgroup does, and gemm.hip's workgroup
// spends its 5.8 us converting fp32 tiles to bf16 hi / lo on their way into LDS: two barriers and a convert / ds_write pass per
// 64-k step, four steps deep even with the K sum on two thread groups (profiles/r04_gemm_timeline.txt).  Here nothing is
// converted and nothing is staged:
//  * operands arrive as bf16 hi / lo planes written by their producers (the optimizer kernels for the weights; LayerNorm, the
//    embedding and this kernel's own epilogue for the activations), zero-padded to 64 rows / 64 k;
//  * a k-major plane row IS the MFMA fragment layout (lane (row l & 15, k-octet l >> 4) reads 16 contiguous bytes), so every
//    wave loads its own fragments straight into registers and issues its MFMAs as they land: no LDS staging, no barrier, no
//    VALU in the K loop; one memory round trip per launch;
//  * ONE MFMA tile (16 rows x 16 columns) per workgroup, 512 threads = 8 waves, wave w = the 64-k tiles w, w + 8, ...: every
//    byte of the workgroup's panels is loaded by one wave, each tile's partial product is computed from zero, and the partials
//    are added in tile order through LDS -- C = ((P_0 + P_1) + P_2) + ... -- this kernel's definition of the K sum
//    (gemm_rows_body: why the tile is this small);
//  * the epilogue (bias -> activation -> gate -> dropout -> residual; fp32 and / or planes out) runs in the accumulator layout;
//    its bias / gate / residual values and the dropout key are requested behind the operand loads, before the first MFMA.
// A SLNLP_ZKERNEL: K fits in lockstep share one launch through grid.z (launch.hpp).
#include <atomic>

#include "common.hpp"
#include "gemm_jobs.hpp"
#include "launch.hpp"

namespace slnlp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct RowsParams {
    slnlp_gemm_args a;
    unsigned drop_thr;
    float drop_scale;
};
__device__ __forceinline__ RowsParams as_global(RowsParams p) {
    launder(p.a);
    return p;
}

#if SLNLP_PROBE_FENCES == 256
// timeline probe build (tools/probes/probe_gemm_timeline.py): wave 0 of every workgroup records 100 MHz timestamps of its phases
constexpr int RTS_MAX = 1 << 14, RTS_W = 6;   // words: entry, first tile's operands landed, partials stored, K sum done, end, {grid, block}
__device__ unsigned long long g_rts[RTS_MAX][RTS_W];
__device__ unsigned g_rts_n;
#define RTS_MARK(slot) do { if (threadIdx.x == 0) rts[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RTS_MARK(slot) do { } while (0)
#endif

constexpr int RT_THREADS = 512, RT_WAVES = 8;
constexpr int RT_MAX_KTILES = 16;             // K <= 1024

// Work split.  Such a launch is bound by what ONE compute unit can load (about 33 GB/s of 16-byte fragment loads): a workgroup
// of a 64 x 16 tiling pulls the whole activation panel plus its weight columns -- 160 KB at K = 512, in 4.8 us whatever the kernel
// does with them (the fp32-operand kernel and a register-direct plane kernel with that tiling both measured 5.8 us per workgroup:
// profiles/r05_gemm_timeline.txt).  So the tile of a solo fit's launch is ONE MFMA tile, 16 rows x 16 columns, on 4 x N / 16
// workgroups (128 for a [50 x 512] product instead of 32): 64 KB and 3.1 us per workgroup.  K fits in lockstep fill the chip anyway
// and pay for bytes instead: their merged launch takes 64 x 16 or 64 x 32 tiles (5 / 3 MB of panel reads per product instead of 8).
// Geometry (MT x NT MFMA tiles per workgroup) is picked per launch (rows_geo) and NEVER changes a result:
// WAVE w takes the 64-k tiles w, w + 8, ... and computes each tile's partial product P_t from zero -- per 32-k step MT A and NT B
// fragments, hi and lo, every byte of the workgroup's panels loaded by exactly one wave.  The partials meet in LDS and are added
// in TILE ORDER -- C = ((P_0 + P_1) + P_2) + ... -- which is the DEFINITION of this kernel's K sum: one value per product however
// many waves, workgroups or fits share the launch, whatever the tile.
template <int NSPLIT, int MT, int NT>
__device__ __forceinline__ void gemm_rows_body(RowsParams p) {
    static_assert(MT * NT <= RT_WAVES, "one epilogue tile per wave");
    extern __shared__ __attribute__((aligned(16))) float part[];      // [ktiles][MT x NT tiles][64 lanes][4]
    const slnlp_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if SLNLP_PROBE_FENCES == 256
    unsigned long long rts[RTS_W] = {0, 0, 0, 0, 0, 0};
    struct RtsFlush {
        unsigned long long* t;
        __device__ ~RtsFlush() {
            if (threadIdx.x == 0) {
                t[4] = __builtin_amdgcn_s_memrealtime();
                t[5] = ((unsigned long long)(gridDim.x * gridDim.y) << 32) | (unsigned)(blockIdx.y * gridDim.x + blockIdx.x);
                const unsigned i = atomicAdd(&g_rts_n, 1u) & (unsigned)(RTS_MAX - 1);
                for (int k = 0; k < RTS_W; ++k) g_rts[i][k] = t[k];
            }
        }
    } rts_flush{rts};
    RTS_MARK(0);
#endif
    const int bn0 = blockIdx.x * 16 * NT, bm0 = blockIdx.y * 16 * MT;
    const int M = g.M, N = g.N;
    const int ktiles = (g.K + 63) >> 6;
    // this lane's fragment rows and its k-octet inside a 32-k step.  The planes are zero-padded to multiples of 64 rows and weight
    // planes are followed by more of the arena; a tile that starts past the last row block re-reads the last valid tile instead
    // (its products belong to rows >= M / columns >= N, which nothing stores)
    const int am_last = ((M + 15) / 16 - 1) * 16, bn_last = ((N + 15) / 16 - 1) * 16;
    const long koct = 8 * (lane >> 4);
    const unsigned short *ah[MT], *al[MT], *bh[NT], *bl[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long off = (long)(min(bm0 + 16 * i, am_last) + (lane & 15)) * g.lda_p + koct;
        ah[i] = g.A_hi + off; al[i] = g.A_lo + off;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const long off = (long)(min(bn0 + 16 * j, bn_last) + (lane & 15)) * g.ldb_p + koct;
        bh[j] = g.B_hi + off; bl[j] = g.B_lo + off;
    }

    // what the epilogue will want: wave w < MT * NT runs it for MFMA tile (w / NT, w % NT); the lane holds rows gm0 .. gm0 + 3 of column gn
    const int ei = wave / NT, ej = wave % NT;
    const int gm0 = bm0 + 16 * ei + ((lane >> 4) << 2), gn = bn0 + 16 * ej + (lane & 15);
    const bool live = wave < MT * NT && gn < N && gm0 < M;
    float bias = 0.f, gt[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    DropKey dkey = {};
    auto epilogue_loads = [&]() {
        if (!live) return;
        if (g.bias) bias = g.bias[gn];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = gm0 + r < M ? gm0 + r : gm0;
            if (g.gate) gt[r] = g.gate[(long)gm * g.ldg + gn];
            if (g.resid) rs[r] = g.resid[(long)gm * g.ldr + gn];
        }
        if (g.drop_p > 0.f) dkey = dropout_key(g.rng, g.drop_site);
    };

    for (int t = wave; t < ktiles; t += RT_WAVES) {
        bf16x8 fa[2][MT], la[2][MT], fb[2][NT], lb[2][NT];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = t * 64 + kk * 32;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                fb[kk][j] = *reinterpret_cast<const bf16x8*>(bh[j] + k);
                if (NSPLIT == 3) lb[kk][j] = *reinterpret_cast<const bf16x8*>(bl[j] + k);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                fa[kk][i] = *reinterpret_cast<const bf16x8*>(ah[i] + k);
                if (NSPLIT == 3) la[kk][i] = *reinterpret_cast<const bf16x8*>(al[i] + k);
            }
        }
        // (requested behind the first tile's operands, in front of its MFMAs: vector-memory operations retire in order, so a slow
        //  residual line in FRONT of the operands would hold the first MFMA's counted wait -- 0.4 us on gemm.hip's 16-wide tile)
        if (t == wave) epilogue_loads();
#if SLNLP_PROBE_FENCES == 256
        if (t == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RTS_MARK(1); }
#endif
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (NSPLIT == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(la[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], lb[kk][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
                }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(part + (((t * MT + i) * NT + j) * 64 + lane) * 4) = acc[i][j];
    }
    if (wave >= ktiles) epilogue_loads();          // (a wave without a K tile still owes its epilogue requests)
    RTS_MARK(2);
    __syncthreads();
    if (wave >= MT * NT) return;
    // the K sum of this wave's MFMA tile, in tile order
    f32x4 acc = *reinterpret_cast<const f32x4*>(part + (wave * 64 + lane) * 4);
    for (int t = 1; t < ktiles; ++t) acc += *reinterpret_cast<const f32x4*>(part + ((t * MT * NT + wave) * 64 + lane) * 4);
    RTS_MARK(3);
    if (!live) return;

    // ---- epilogue in the accumulator layout
    unsigned lot[4] = {0u, 0u, 0u, 0u};
    if (g.drop_p > 0.f) {
        if (g.drop_head_dim == 0) {
            const uint4 bits = dropout_bits8(dkey, (unsigned)gm0 >> 2, drop_cc((unsigned)gn));
            const int h = drop_half((unsigned)gn);
#pragma unroll
            for (int r = 0; r < 4; ++r) lot[r] = pick_lot(bits, h, r);
        } else {                                  // one keep / drop decision per (row, head), see slnlp.h
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned rh = (unsigned)(gm0 + r) * (unsigned)(N / g.drop_head_dim) + (unsigned)(gn / g.drop_head_dim);
                lot[r] = pick_lot(dropout_bits8(dkey, rh >> 2, 0u), 0, (int)(rh & 3u));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gm = gm0 + r;
        if (gm >= M) break;
        float v = acc[r] + bias;
        if (g.relu == 1) v = fmaxf(v, 0.f);
        else if (g.relu == 2) v = tanhf(v);
        if (g.gate) v = g.gate_mode == 1 ? v * (1.f - gt[r] * gt[r]) : (gt[r] > 0.f ? v * g.gate_scale : 0.f);
        if (g.drop_p > 0.f) v = (lot[r] >= p.drop_thr) ? v * p.drop_scale : 0.f;
        if (g.resid) v += rs[r];
        if (g.C) g.C[(long)gm * g.ldc + gn] = v;
        if (g.C_hi) {
            unsigned short h, l;
            split_bf16(v, h, l);
            g.C_hi[(long)gm * g.ldc_p + gn] = h;
            if (g.C_lo) g.C_lo[(long)gm * g.ldc_p + gn] = l;
        }
    }
}

// geometries: MFMA tiles per workgroup
struct RowsGeo { int mt, nt; };
constexpr int RT_NGEO = 3;
constexpr RowsGeo RT_GEO[RT_NGEO] = {{1, 1}, {4, 1}, {4, 2}};
#define SLNLP_ROWS_KERNEL(NS, G)                                                                                          \
    __device__ __forceinline__ void gemm_rows_body_##NS##_##G(RowsParams p) { gemm_rows_body<NS, RT_GEO[G].mt, RT_GEO[G].nt>(p); } \
    SLNLP_ZKERNEL(gemm_rows_kernel_##NS##_##G, RT_THREADS, gemm_rows_body_##NS##_##G)
SLNLP_ROWS_KERNEL(3, 0)
SLNLP_ROWS_KERNEL(3, 1)
SLNLP_ROWS_KERNEL(3, 2)
SLNLP_ROWS_KERNEL(1, 0)
SLNLP_ROWS_KERNEL(1, 1)
SLNLP_ROWS_KERNEL(1, 2)

using RowsKernel = void (*)(Pack<RowsParams>, const Pack<RowsParams>*);
static RowsKernel rows_kernel(int precision, int geo) {
    if (precision == 3) return geo == 0 ? gemm_rows_kernel_3_0 : geo == 1 ? gemm_rows_kernel_3_1 : gemm_rows_kernel_3_2;
    return geo == 0 ? gemm_rows_kernel_1_0 : geo == 1 ? gemm_rows_kernel_1_1 : gemm_rows_kernel_1_2;
}
static size_t rows_lds(int geo, int K) { return (size_t)ceil_div(K, 64) * RT_GEO[geo].mt * RT_GEO[geo].nt * 64 * 4 * sizeof(float); }
static dim3 rows_grid(int geo, int M, int N) { return dim3(ceil_div(N, 16 * RT_GEO[geo].nt), ceil_div(M, 16 * RT_GEO[geo].mt)); }

// -1 = automatic; 0 .. RT_NGEO-1 forced (slnlp_set_rows_tile: tests, tuning)
static std::atomic<int> g_rows_geo{[] { const char* e = getenv("SLNLP_ROWS_TILE"); const int v = e ? atoi(e) : -1; return v >= 0 && v < RT_NGEO ? v : -1; }()};
// Which tile a launch of `fits` products [M x N x K] takes: rounds of workgroups over the 256 compute units x the panel bytes one
// workgroup loads (a compute unit's load rate is the limit, see gemm_rows_body) -- the small tile while the launch fits the chip
// about once (one fit: 128 workgroups of 64 KB), the wide ones when K fits in lockstep fill it many times over.
static int rows_geo(int M, int N, int K, int fits) {
    const int forced = g_rows_geo.load(std::memory_order_relaxed);
    if (forced >= 0) return forced;
    int best = 0;
    long best_cost = -1;
    for (int geo = 0; geo < RT_NGEO; ++geo) {
        const dim3 gr = rows_grid(geo, M, N);
        const long units = (long)gr.x * gr.y * fits, rounds = (units + 255) / 256;
        const long cost = rounds * (16L * (RT_GEO[geo].mt + RT_GEO[geo].nt) * K * 4 + 16384);     // (+ a fixed cost per round)
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = geo; }
    }
    return best;
}

static int rows_init() {
    static DeviceOnce once;
    return once.run([]() -> int {
        bool ok = true;
        for (int prec = 1; prec <= 3; prec += 2)
            for (int geo = 0; geo < RT_NGEO; ++geo)
                ok = ok && hipFuncSetAttribute((const void*)rows_kernel(prec, geo), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)rows_lds(geo, 64 * RT_MAX_KTILES)) == hipSuccess;
        if (!ok) {
            set_error("gemm_rows: cannot raise dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
        return 0;
    });
}

// A merged (lockstep) launch of K fits' recorded gemm_rows call sites picks its tile again for K products (lockstep.hip): returns the
// kernel to launch and its grid / LDS, or nullptr when `fn` is not a gemm_rows kernel.  Results do not depend on the tile.
const void* gemm_rows_for_fits(const void* fn, const void* recorded_args, int fits, dim3* grid, size_t* lds) {
    int prec = 0;
    for (int geo = 0; geo < RT_NGEO; ++geo) {
        if (fn == (const void*)rows_kernel(3, geo)) prec = 3;
        if (fn == (const void*)rows_kernel(1, geo)) prec = 1;
    }
    if (!prec) return nullptr;
    const slnlp_gemm_args& a = reinterpret_cast<const RowsParams*>(recorded_args)->a;
    const int geo = rows_geo(a.M, a.N, a.K, fits);
    *grid = rows_grid(geo, a.M, a.N);
    *lds = rows_lds(geo, a.K);
    return (const void*)rows_kernel(prec, geo);
}

int gemm_rows(const slnlp_gemm_args& a, hipStream_t st) {
    SLNLP_CHECK_ARG(a.A_hi && a.B_hi, "gemm_rows: operand planes required");
    SLNLP_CHECK_ARG(a.a_kmajor && a.b_kmajor, "gemm_rows: both operands k-major (the forward products y = x W^T)");
    SLNLP_CHECK_ARG(a.C || a.C_hi, "gemm_rows: no output");
    SLNLP_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "gemm_rows: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    SLNLP_CHECK_ARG(a.precision == 1 || (a.precision == 3 && a.A_lo && a.B_lo), "gemm_rows: precision 1, or 3 with both lo planes");
    SLNLP_CHECK_ARG(a.lda_p % 64 == 0 && a.ldb_p % 64 == 0 && a.lda_p >= a.K && a.ldb_p >= a.K,
                    "gemm_rows: plane row strides must be multiples of 64 and cover K (zero-padded)");
    SLNLP_CHECK_ARG((((uintptr_t)a.A_hi | (uintptr_t)a.B_hi | (uintptr_t)a.A_lo | (uintptr_t)a.B_lo) & 15) == 0,
                    "gemm_rows: planes must be 16-byte aligned");
    SLNLP_CHECK_ARG(!a.C || a.ldc >= a.N, "gemm_rows: ldc < N");
    SLNLP_CHECK_ARG(!a.C_hi || a.ldc_p >= a.N, "gemm_rows: ldc_p < N");
    SLNLP_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f && (a.drop_p == 0.f || a.rng), "gemm_rows: bad dropout args");
    SLNLP_CHECK_ARG(!a.gate || a.ldg >= a.N, "gemm_rows: ldg too small");
    SLNLP_CHECK_ARG(!a.resid || a.ldr >= a.N, "gemm_rows: ldr too small");
    SLNLP_CHECK_ARG(!a.rowsum_a && a.batch <= 1, "gemm_rows: no row sums, no batched jobs");
    SLNLP_CHECK_ARG(a.drop_head_dim >= 0 && (a.drop_head_dim == 0 || a.N % a.drop_head_dim == 0),
                    "gemm_rows: drop_head_dim %d does not divide N %d", a.drop_head_dim, a.N);
    RowsParams p;
    p.a = a;
    p.drop_thr = dropout_threshold(a.drop_p);
    p.drop_scale = 1.f / (1.f - a.drop_p);
    const int ktiles = ceil_div(a.K, 64);
    SLNLP_CHECK_ARG(ktiles <= RT_MAX_KTILES, "gemm_rows: K = %d > %d (the workgroup's partial tiles live in LDS)", a.K, 64 * RT_MAX_KTILES);
    SLNLP_TRY(rows_init());
    const int geo = rows_geo(a.M, a.N, a.K, 1);
    return zlaunch(rows_kernel(a.precision, geo), rows_grid(geo, a.M, a.N), RT_THREADS, rows_lds(geo, a.K), st, "gemm_rows", p);
}

}  // namespace slnlp

#if SLNLP_PROBE_FENCES == 256
// probe build only: copy the recorded workgroup timelines to the host and reset the recorder; returns the number recorded
extern "C" int slnlp_probe_rows_ts(unsigned long long* dst, int max_entries) {
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(slnlp::g_rts_n), sizeof(n)) != hipSuccess) return -1;
    if (n > (unsigned)slnlp::RTS_MAX) n = slnlp::RTS_MAX;
    if ((int)n > max_entries) n = max_entries;
    if (n && hipMemcpyFromSymbol(dst, HIP_SYMBOL(slnlp::g_rts), (size_t)n * slnlp::RTS_W * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(slnlp::g_rts_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}
#endif

extern "C" int slnlp_set_rows_tile(int tile) {
    if (tile < -1 || tile >= slnlp::RT_NGEO) {
        slnlp::set_error("set_rows_tile: %d (-1 = automatic, 0 = 16 x 16, 1 = 64 x 16, 2 = 64 x 32 outputs per workgroup)", tile);
        return SLNLP_ERR_INVALID_ARG;
    }
    slnlp::g_rows_geo.store(tile, std::memory_order_relaxed);
    return 0;
}

extern "C" int slnlp_gemm_rows(const slnlp_gemm_args* args, void* stream) {
    if (!args) {
        slnlp::set_error("gemm_rows: null args");
        return SLNLP_ERR_INVALID_ARG;
    }
    return slnlp::gemm_rows(*args, (hipStream_t)stream);
}
