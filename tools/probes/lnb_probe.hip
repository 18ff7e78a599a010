// lnb_probe.hip -- an instrumented copy of the LayerNorm-backward arithmetic (no dropout, 4 rows per wave, E = 512) for the
// multi-queue investigation.  Besides dx it records, per row, XOR checksums of the bits of the dy / x
// values the wave actually LOADED, the two row sums it computed, and an XOR checksum of what it read back from dx after its own
// stores: a wrong dx row with right checksums = the arithmetic or the store went wrong; wrong checksums = the load returned
// other bytes than memory holds.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/lnb_probe.hip -o tools/probes/liblnb.so
#include <hip/hip_runtime.h>

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int wave_xor(int v) {
    v ^= dpp_i<0xB1>(v); v ^= dpp_i<0x4E>(v); v ^= dpp_i<0x141>(v); v ^= dpp_i<0x140>(v);
    return __builtin_amdgcn_readlane(v, 0) ^ __builtin_amdgcn_readlane(v, 16) ^ __builtin_amdgcn_readlane(v, 32) ^ __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ float wave_sum(float x) {
    int v = 0;
    float f = x;
    f += __builtin_bit_cast(float, dpp_i<0xB1>(__builtin_bit_cast(int, f)));
    f += __builtin_bit_cast(float, dpp_i<0x4E>(__builtin_bit_cast(int, f)));
    f += __builtin_bit_cast(float, dpp_i<0x141>(__builtin_bit_cast(int, f)));
    f += __builtin_bit_cast(float, dpp_i<0x140>(__builtin_bit_cast(int, f)));
    v = __builtin_bit_cast(int, f);
    auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, l)); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}
__device__ __forceinline__ int bits4(float4 a) { return __float_as_int(a.x) ^ (__float_as_int(a.y) * 3) ^ (__float_as_int(a.z) * 5) ^ (__float_as_int(a.w) * 7); }

// diag[row] = {xor(dy bits), xor(x bits), s1 bits, s2 bits, xor(dx read back), 0, 0, 0}
__global__ __launch_bounds__(256) void lnb_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ stats, int rows, float* __restrict__ dx, int* __restrict__ diag) {
    constexpr int E = 512, U = 2, GS = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * GS;
    if (row0 >= rows) return;
    float4 g[U], d[GS][U], v[GS][U];
    float mean[GS], rstd[GS];
    for (int u = 0; u < U; ++u) g[u] = *reinterpret_cast<const float4*>(gamma + lane * 4 + u * 256);
#pragma unroll
    for (int i = 0; i < GS; ++i) {
        const int row = row0 + i < rows ? row0 + i : rows - 1;
        mean[i] = stats[2 * row]; rstd[i] = stats[2 * row + 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[i][u] = *reinterpret_cast<const float4*>(dy + (long)row * E + lane * 4 + u * 256);
            v[i][u] = *reinterpret_cast<const float4*>(x + (long)row * E + lane * 4 + u * 256);
        }
    }
    const float invE = 1.f / (float)E;
#pragma unroll
    for (int i = 0; i < GS; ++i) {
        const int row = row0 + i;
        if (row >= rows) break;
        float a1 = 0.f, a2 = 0.f;
        int cd = 0, cv = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 dd = d[i][u], vv = v[i][u];
            cd ^= bits4(dd) * (u + 1); cv ^= bits4(vv) * (u + 1);
            const float4 xh = make_float4((vv.x - mean[i]) * rstd[i], (vv.y - mean[i]) * rstd[i], (vv.z - mean[i]) * rstd[i], (vv.w - mean[i]) * rstd[i]);
            const float4 gv = make_float4(dd.x * g[u].x, dd.y * g[u].y, dd.z * g[u].z, dd.w * g[u].w);
            a1 += gv.x + gv.y + gv.z + gv.w;
            a2 += gv.x * xh.x + gv.y * xh.y + gv.z * xh.z + gv.w * xh.w;
        }
        // lane-position-dependent checksums: a piece that moved between lanes changes them too
        cd = wave_xor(cd * (2 * lane + 1)); cv = wave_xor(cv * (2 * lane + 1));
        const float s1 = wave_sum(a1) * invE, s2 = wave_sum(a2) * invE;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 dd = d[i][u], vv = v[i][u];
            const float4 xh = make_float4((vv.x - mean[i]) * rstd[i], (vv.y - mean[i]) * rstd[i], (vv.z - mean[i]) * rstd[i], (vv.w - mean[i]) * rstd[i]);
            const float4 gv = make_float4(dd.x * g[u].x, dd.y * g[u].y, dd.z * g[u].z, dd.w * g[u].w);
            float4 o;
            o.x = rstd[i] * (gv.x - s1 - xh.x * s2); o.y = rstd[i] * (gv.y - s1 - xh.y * s2);
            o.z = rstd[i] * (gv.z - s1 - xh.z * s2); o.w = rstd[i] * (gv.w - s1 - xh.w * s2);
            *reinterpret_cast<float4*>(dx + (long)row * E + lane * 4 + u * 256) = o;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int cb = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const volatile float* bp = dx + (long)row * E + lane * 4 + u * 256;
            const float4 back = make_float4(bp[0], bp[1], bp[2], bp[3]);
            cb ^= bits4(back) * (u + 1);
        }
        cb = wave_xor(cb * (2 * lane + 1));
        if (lane == 0) {
            int* q = diag + (long)row * 8;
            q[0] = cd; q[1] = cv; q[2] = __float_as_int(s1); q[3] = __float_as_int(s2); q[4] = cb;
        }
    }
}

extern "C" int lnb_launch(void* stream, const float* dy, const float* x, const float* gamma, const float* stats, int rows, float* dx, int* diag) {
    hipLaunchKernelGGL(lnb_kernel, dim3((rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, stats, rows, dx, diag);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
