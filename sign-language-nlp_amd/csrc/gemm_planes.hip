// gemm_planes.hip -- split-bf16 MFMA GEMM over PRE-SPLIT operands, staged by LDS-DMA.
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ),  A = Ahi + Alo, B = Bhi + Blo  (bf16 planes)
//
// gemm.hip converts fp32 -> bf16 hi/lo inside its K-loop; measured on MI355X that loop is ~90 %
// address / convert / ds_write / barrier work and ~10 % MFMA, and every element is converted again
// by each tile that reads it.  Here every operand arrives as two bf16 planes written ONCE by its
// producer (GEMM / LayerNorm / attention epilogues, the weight splitter), zero-padded to multiples
// of 64 in both dimensions, so a K-step is: 4 x global_load_lds_dwordx4 per wave (global -> LDS
// DMA, no VGPRs, no VALU, no ds_write), one counted s_waitcnt, two barriers, fragment reads, MFMA.
//
//  * tile 64x64x64, 8 waves (4x2, each 16x32), two LDS stages of {Ahi, Alo, Bhi, Blo} x 8 KiB;
//  * LDS-DMA writes lane-linearly, so the bank-conflict swizzle is applied to the SOURCE address
//    and undone with the same XOR on the fragment read (both are involutions):
//      k-major image [row][64 k]  : 16-B slot ^= (row & 7)            -> conflict-free ds_read_b128
//      m-major image [k][64 rows] : 16-B slot ^= 2*(k>>1 & 1 | k>>3 & 1 << 1) -> ds_read_b64_tr_b16
//  * padding makes every tile interior: no bounds logic anywhere in the loop.
#include "common.hpp"

namespace slnlp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;

constexpr int PT = 64;                 // tile edge (M, N and K)
constexpr int IMG = PT * PT;           // bf16 elements per plane image (8 KiB)
constexpr int PTHREADS = 512;
constexpr int NSTAGE = 2;              // LDS ring depth (NSTAGE-1 tiles in flight)

struct PlaneGemmParams {
    slnlp_gemm_args a;
    unsigned drop_thr;
    float drop_scale;
};

__device__ __forceinline__ int mswz(int k) { return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1; }

// element offset of logical (row, k) inside a plane image
template <bool KMAJOR>
__device__ __forceinline__ int img_off(int row, int k) {
    if (KMAJOR) return row * PT + ((((k >> 3) ^ (row & 7)) << 3) | (k & 7));
    return k * PT + ((((row >> 3) ^ mswz(k)) << 3) | (row & 7));
}

// One wave copies 8 image lines (1 KiB) of one plane: lane -> (line 8*wave + lane/8, physical slot lane%8),
// source = the logical slot that the swizzle maps there.
template <bool KMAJOR>
__device__ __forceinline__ void dma_plane(const unsigned short* __restrict__ plane, long ld, int row0, int k0,
                                          unsigned short* img, int wave, int lane) {
    const int line = 8 * wave + (lane >> 3), ps = lane & 7;
    const unsigned short* src;
    if (KMAJOR) src = plane + (long)(row0 + line) * ld + k0 + ((ps ^ (line & 7)) << 3);
    else src = plane + (long)(k0 + line) * ld + row0 + ((ps ^ mswz(line)) << 3);
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(img + wave * 512), 16, 0, 0);
}

template <bool KMAJOR>
__device__ __forceinline__ bf16x8 pfrag(const unsigned short* __restrict__ img, int r0, int kk, int lane) {
    if (KMAJOR) {
        return *reinterpret_cast<const bf16x8*>(img + img_off<true>(r0 + (lane & 15), kk * 32 + ((lane >> 4) << 3)));
    } else {
        const int i = lane & 15, kb = kk * 32 + ((lane >> 4) << 3) + (i >> 2), col = r0 + ((i & 3) << 2);
        typedef __attribute__((address_space(3))) s16x4* lds_p;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + img_off<false>(col, kb)));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + img_off<false>(col, kb + 4)));
        const s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

__device__ __forceinline__ float pbf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

template <int NSPLIT, bool AK, bool BK>
__global__ __launch_bounds__(PTHREADS) void gemm_planes_kernel(const PlaneGemmParams p) {
    constexpr int NP = NSPLIT == 3 ? 2 : 1;
    constexpr int STAGE = 2 * NP * IMG;     // A planes then B planes
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];   // NSTAGE stages (+ row-sum scratch after the loop)
    const slnlp_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 16, wn0 = (wave & 1) * 32;
    int bx, by;
    {   // XCD-aware tile order (see gemm.hip)
        const int nwg = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        by = t / gridDim.x;
        bx = t - by * gridDim.x;
    }
    const int bm0 = by * PT, bn0 = bx * PT;
    const int M = g.M, N = g.N, K = g.K;
    const int ktiles = (K + PT - 1) / PT;

    auto issue = [&](int kt, int stage) {
        const int k0 = (kt < ktiles ? kt : 0) * PT;       // past-the-end prefetch re-reads tile 0 (never consumed)
        unsigned short* s = smem + stage * STAGE;
        dma_plane<AK>(g.A_hi, g.lda_p, bm0, k0, s, wave, lane);
        if (NSPLIT == 3) dma_plane<AK>(g.A_lo, g.lda_p, bm0, k0, s + IMG, wave, lane);
        dma_plane<BK>(g.B_hi, g.ldb_p, bn0, k0, s + NP * IMG, wave, lane);
        if (NSPLIT == 3) dma_plane<BK>(g.B_lo, g.ldb_p, bn0, k0, s + NP * IMG + IMG, wave, lane);
    };

    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bool do_rowsum = (g.rowsum_a != nullptr) && (bx == 0);
    float rowsum = 0.f;

    // NSTAGE-deep LDS ring: tiles kt+1 .. kt+NSTAGE-1 are in flight while tile kt is consumed (a K-step's
    // MFMA work is ~0.2 us, one DMA round trip ~1 us).  ONE barrier per step: the stage refilled at step kt
    // was consumed at step kt-1, which every wave has finished once it passes this step's barrier.
    for (int t = 0; t < NSTAGE - 1; ++t) issue(t, t);
    for (int kt = 0; kt < ktiles; ++kt) {
        // this wave's DMA of tile kt has landed once only the (NSTAGE-2) newer tiles' pieces are outstanding
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"((NSTAGE - 2) * 2 * NP) : "memory");
        __builtin_amdgcn_s_barrier();                      // ... and so has every other wave's part
        asm volatile("" ::: "memory");
        issue(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
        const unsigned short* s = smem + (kt % NSTAGE) * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah, al, bh[2], bl[2];
            ah = pfrag<AK>(s, wm0, kk, lane);
            if (NSPLIT == 3) al = pfrag<AK>(s + IMG, wm0, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = pfrag<BK>(s + NP * IMG, wn0 + 16 * j, kk, lane);
                if (NSPLIT == 3) bl[j] = pfrag<BK>(s + NP * IMG + IMG, wn0 + 16 * j, kk, lane);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (NSPLIT == 3) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[j], 0, 0, 0);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[j], 0, 0, 0);
            }
        }
        if (do_rowsum) {   // thread owns row (tid & 63), k-octet (tid >> 6)
            const int row = tid & 63, kq = (tid >> 6) * 8;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int off = img_off<AK>(row, kq + k);
                t += pbf2f(s[off]);
                if (NSPLIT == 3) t += pbf2f(s[IMG + off]);
            }
            rowsum += t;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the dummy prefetches before LDS is reused / freed
    __builtin_amdgcn_s_barrier();
    if (do_rowsum) {
        float* rs = reinterpret_cast<float*>(smem);
        rs[(tid >> 6) * PT + (tid & 63)] = rowsum;
        __syncthreads();
        if (tid < PT && bm0 + tid < M) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < PTHREADS / 64; ++w) t += rs[w * PT + tid];
            g.rowsum_a[bm0 + tid] = t;
        }
    }

    // ---- epilogue: +bias -> activation -> gate -> dropout -> +resid ; fp32 store (+ optional bf16 planes)
    const int crow = (lane >> 4) << 2, ccol = lane & 15;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gm0 = bm0 + wm0 + crow;
        const int gn = bn0 + wn0 + j * 16 + ccol;
        if (gn >= N || gm0 >= M) continue;
        const float bias = g.bias ? g.bias[gn] : 0.f;
        uint4 bits = make_uint4(0, 0, 0, 0);
        if (g.drop_p > 0.f) bits = dropout_bits4(g.rng, g.drop_site, (unsigned)gm0 >> 2, (unsigned)gn);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = gm0 + r;
            if (gm >= M) break;
            float v = acc[j][r] + bias;
            if (g.relu == 1) v = fmaxf(v, 0.f);
            else if (g.relu == 2) v = tanhf(v);
            if (g.gate) {
                const float gt = g.gate[(long)gm * g.ldg + gn];
                v = g.gate_mode == 1 ? v * (1.f - gt * gt) : (gt > 0.f ? v * g.gate_scale : 0.f);
            }
            if (g.drop_p > 0.f) v = (pick_word(bits, r) >= p.drop_thr) ? v * p.drop_scale : 0.f;
            if (g.resid) v += g.resid[(long)gm * g.ldr + gn];
            if (g.C) g.C[(long)gm * g.ldc + gn] = v;
            if (g.C_hi) {
                const unsigned u = __float_as_uint(v);
                g.C_hi[(long)gm * g.ldc_p + gn] = (unsigned short)(u >> 16);
                if (g.C_lo) {
                    __bf16 lo = (__bf16)(v - __uint_as_float(u & 0xFFFF0000u));
                    g.C_lo[(long)gm * g.ldc_p + gn] = __builtin_bit_cast(unsigned short, lo);
                }
            }
        }
    }
}

constexpr size_t PLANE_LDS = (size_t)NSTAGE * 2 * 2 * IMG * sizeof(unsigned short);   // NSTAGE x (A,B) x (hi,lo) x 8 KiB = 128 KiB

template <int NSPLIT, bool AK, bool BK>
static int launch_planes(const PlaneGemmParams& p, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)gemm_planes_kernel<NSPLIT, AK, BK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)PLANE_LDS) != hipSuccess) {
            set_error("gemm_planes: cannot raise dynamic LDS limit");
            return SLNLP_ERR_LAUNCH;
        }
        attr = true;
    }
    dim3 grid(ceil_div(p.a.N, PT), ceil_div(p.a.M, PT));
    hipLaunchKernelGGL((gemm_planes_kernel<NSPLIT, AK, BK>), grid, dim3(PTHREADS), PLANE_LDS, s, p);
    return 0;
}

template <int NSPLIT, bool AK, bool BK>
static bool set_lds_attr() {
    return hipFuncSetAttribute((const void*)gemm_planes_kernel<NSPLIT, AK, BK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)PLANE_LDS) == hipSuccess;
}

// set every instantiation's LDS attribute up front (plan creation) so none lands inside a graph capture
int gemm_planes_init() {
    const bool ok = set_lds_attr<3, true, true>() && set_lds_attr<3, true, false>() && set_lds_attr<3, false, false>() &&
                    set_lds_attr<1, true, true>() && set_lds_attr<1, true, false>() && set_lds_attr<1, false, false>();
    if (!ok) {
        set_error("gemm_planes_init: cannot raise dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    return 0;
}

int gemm_planes(const slnlp_gemm_args& a, hipStream_t s) {
    SLNLP_CHECK_ARG(a.A_hi && a.B_hi, "gemm_planes: operand planes required");
    SLNLP_CHECK_ARG(a.C || a.C_hi, "gemm_planes: no output");
    SLNLP_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "gemm_planes: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    SLNLP_CHECK_ARG(a.precision == 1 || (a.precision == 3 && a.A_lo && a.B_lo), "gemm_planes: precision 3 needs lo planes");
    SLNLP_CHECK_ARG(a.lda_p % 64 == 0 && a.ldb_p % 64 == 0, "gemm_planes: plane row strides must be multiples of 64");
    SLNLP_CHECK_ARG((((uintptr_t)a.A_hi | (uintptr_t)a.B_hi | (uintptr_t)a.A_lo | (uintptr_t)a.B_lo) & 15) == 0,
                    "gemm_planes: planes must be 16-byte aligned");
    SLNLP_CHECK_ARG(!a.C || a.ldc >= a.N, "gemm_planes: ldc < N");
    SLNLP_CHECK_ARG(!a.C_hi || a.ldc_p >= a.N, "gemm_planes: ldc_p < N");
    SLNLP_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f && (a.drop_p == 0.f || a.rng), "gemm_planes: bad dropout args");
    SLNLP_CHECK_ARG(!(a.a_kmajor == 0 && a.b_kmajor != 0), "gemm_planes: layout (A m-major, B k-major) not built");
    PlaneGemmParams p;
    p.a = a;
    p.drop_thr = dropout_threshold(a.drop_p);
    p.drop_scale = 1.f / (1.f - a.drop_p);
    const bool ak = a.a_kmajor != 0, bk = a.b_kmajor != 0;
    int rc;
    if (a.precision == 3) rc = (ak && bk) ? launch_planes<3, true, true>(p, s) : ak ? launch_planes<3, true, false>(p, s) : launch_planes<3, false, false>(p, s);
    else rc = (ak && bk) ? launch_planes<1, true, true>(p, s) : ak ? launch_planes<1, true, false>(p, s) : launch_planes<1, false, false>(p, s);
    if (rc) return rc;
    SLNLP_CHECK_LAUNCH("gemm_planes");
    return 0;
}

// fp32 [R, C] (row stride ld) -> bf16 hi / lo planes with row stride ldp (valid region only; the padding of the
// planes stays zero from their one-time memset).  Used for the weights once per step and by tests.
__global__ void split_planes_kernel(const float* __restrict__ x, long ld, int R, int C, unsigned short* __restrict__ hi,
                                    unsigned short* __restrict__ lo, long ldp) {
    const long n4 = (long)R * (C >> 2);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int r = (int)(i / (C >> 2)), c = (int)(i % (C >> 2)) << 2;
        const float4 v = *reinterpret_cast<const float4*>(x + (long)r * ld + c);
        const float f[4] = {v.x, v.y, v.z, v.w};
        unsigned short h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned u = __float_as_uint(f[e]);
            h[e] = (unsigned short)(u >> 16);
            __bf16 b = (__bf16)(f[e] - __uint_as_float(u & 0xFFFF0000u));
            l[e] = __builtin_bit_cast(unsigned short, b);
        }
        uint2 w;
        w.x = h[0] | ((unsigned)h[1] << 16); w.y = h[2] | ((unsigned)h[3] << 16);
        *reinterpret_cast<uint2*>(hi + (long)r * ldp + c) = w;
        if (lo) {
            w.x = l[0] | ((unsigned)l[1] << 16); w.y = l[2] | ((unsigned)l[3] << 16);
            *reinterpret_cast<uint2*>(lo + (long)r * ldp + c) = w;
        }
    }
}

int split_planes(const float* x, int64_t ld, int R, int C, unsigned short* hi, unsigned short* lo, int64_t ldp, hipStream_t st) {
    SLNLP_CHECK_ARG(x && hi && R > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ldp % 4 == 0 && ldp >= C,
                    "split_planes: bad args (C, ld, ldp must be multiples of 4)");
    int grid = ceil_div((long)R * (C / 4), 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, st, x, (long)ld, R, C, hi, lo, (long)ldp);
    SLNLP_CHECK_LAUNCH("split_planes");
    return 0;
}

}  // namespace slnlp

extern "C" int slnlp_split_planes(const float* x, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, int64_t ldp,
                                  void* stream) {
    return slnlp::split_planes(x, ld, R, C, hi, lo, ldp, (hipStream_t)stream);
}
