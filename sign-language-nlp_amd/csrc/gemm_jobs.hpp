// gemm_jobs.hpp -- job descriptors of the grouped GEMM kernels (gemm.hip, gemm_planes.hip), shared with the lockstep
// driver (lockstep.hip), which concatenates the job lists of K fits into one device-resident table per launch.
#pragma once
#include <vector>
#include "common.hpp"
#include "launch.hpp"

namespace slnlp {

// ---- fp32-operand GEMM (gemm.hip)
struct GemmParams {
    slnlp_gemm_args a;
    unsigned drop_thr;
    float drop_scale;
    int a_vec, b_vec;  // 16-B vector loads legal for this operand
};
constexpr int GEMM_GROUP_MAX = 8;
struct GemmGroupParams {
    GemmParams job[GEMM_GROUP_MAX];
    int variant[GEMM_GROUP_MAX];      // (a_kmajor, b_kmajor, narrow) -> 0..5
    int gx[GEMM_GROUP_MAX], gy[GEMM_GROUP_MAX], block_begin[GEMM_GROUP_MAX];
    int njobs;
};
// one entry of a merged (lockstep) launch's job table; blocks [block_begin, block_begin + gx * gy) run it
struct GemmJob {
    GemmParams p;
    int variant, gx, gy, block_begin;
};

// ---- pre-split plane GEMM (gemm_planes.hip)
// One GEMM of a grouped launch.  A launch runs up to MAX_JOBS independent GEMMs (e.g. the data-gradient and
// the weight-gradient of one dY): blocks [block_begin, block_begin + tiles_x * tiles_y * nks) belong to the job.
// nks > 1 splits the K loop over nks blocks per output tile; partial tiles go to scratch in the accumulator
// layout and the LAST block to arrive (per-tile counter) adds them in split order -- deterministic, no float atomics.
struct PlaneJob {
    slnlp_gemm_args a;
    unsigned drop_thr;
    float drop_scale;
    int variant;        // 0: A,B k-major   1: A k-major, B m-major   2: A,B m-major
    int tiles_x, tiles_y, nks, block_begin;
    int vec_out;        // epilogue may use 16-byte row pieces (N % 4 == 0, aligned C / residual / planes)
    int hand_off;       // split-K meeting point: 0 = sc1 accesses only; probes (SLNLP_SPLITK_MODE): 2 = + reader acquire, 3 = + writer release
    float* part;        // [tile][nks][bm x bn]   (accumulator layout)
    float* part_rs;     // [tile_y][nks][bm]      (row sums of A)
    int* counters;      // [tiles], zero outside a launch
};
constexpr int MAX_JOBS = 4;
struct PlaneGroupParams {
    PlaneJob job[MAX_JOBS];
    int njobs;
};

// every pointer of a job read from a device table is generic to the compiler: make them global again (launch.hpp)
__device__ __forceinline__ void launder(slnlp_gemm_args& a) {
    a.A = as_global(a.A); a.B = as_global(a.B); a.C = as_global(a.C);
    a.bias = as_global(a.bias); a.gate = as_global(a.gate); a.rng = as_global(a.rng);
    a.resid = as_global(a.resid); a.rowsum_a = as_global(a.rowsum_a);
    a.A_hi = as_global(a.A_hi); a.A_lo = as_global(a.A_lo); a.B_hi = as_global(a.B_hi); a.B_lo = as_global(a.B_lo);
    a.C_hi = as_global(a.C_hi); a.C_lo = as_global(a.C_lo);
    a.col_scale = as_global(a.col_scale); a.C_q8 = as_global(a.C_q8);
}

// kernels a merged launch replays (type-erased by the recorder; declared here so lockstep.hip can name them)
const void* gemm_group_kernel_ptr(int precision, int ks = 1);
const void* rnn_step_fwd_for_blocks(const void* fn, const void* recorded_args, int fits, int* threads, dim3* grid);   // the forward-step kernel (tile, thread groups, grid) of a merged launch (gemm.hip)
const void* gemm_rows_for_fits(const void* fn, const void* recorded_args, int fits, dim3* grid, size_t* lds);   // the B-row plane kernel of a merged launch (gemm_rows.hip)
int gemm_group_ks(int blocks, int longest_ktiles);   // thread groups per workgroup a launch of this size takes (results do not depend on it)
// tile geometries of the plane GEMM (gemm_planes.hip, GEO[]): 0 = 64 x 64, 1 = 128 x 128 (64-k stages), 2 = 128 x 128 (32-k stages), 3 = 256 x 256 (32-k stages)
const void* gemm_planes_kernel_ptr(int precision, int geo);
int plane_geo_for(const slnlp_gemm_args* jobs, const int* split_k, int njobs);   // the geometry a launch of these jobs takes
void plane_job_retile(PlaneJob& j, int geo);
void plane_merge_geometry(const void* recorded_fn, PlaneJob* jobs, int njobs, const void** fn, size_t* lds);
// the block map of a merged plane-GEMM launch: units of all jobs placed on the XCDs (1; 0: fp8 launch, keep the per-job layout;
// -1: more jobs / units than a map entry can name)
int plane_merge_place(const void* merged_fn, const PlaneJob* jobs, int njobs, std::vector<int>& map);

}  // namespace slnlp
