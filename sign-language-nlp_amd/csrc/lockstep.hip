// lockstep.hip -- K fits of one shape (Transformer or LSTM / GRU encoder-decoder) advancing through ONE launch sequence.
//
// What it replaces: the reference runs its (candidate x fold) fits as independent dask tasks, one after another per
// worker (/root/reference/main.py:70-78, helper.py:490-526).  A batch-50 fit cannot fill 256 CUs (its decoder stages
// are 50-row kernels), so here the K fits of a work unit (same shapes, own weights / lr / dropout / seed / data) step
// together: every call site of the step is launched ONCE for all K fits (launch.hpp).
//
//  record   for each fit, run the ordinary plan code (forward + criterion + backward + clip + SGD, or an eval
//           forward) with a Recorder installed: nothing is launched, the call sites come back as a list of
//           {kernel, grid, argument pack};
//  merge    call site i of every fit must be the same kernel with the same geometry (same shapes => same code
//           path); the K packs become one device table and grid.z = K; grouped-GEMM call sites concatenate their job
//           lists into one device job table + a block -> job map (jobs start on multiples of 8 blocks so the
//           XCD-aware tile order inside a job still sees blocks b and b + 8 on one XCD);
//  replay   one hipLaunchKernel per merged call site.  Programs are cached per (data slot, batch size, train).
//
// Per-fit arithmetic is untouched -- block (x, y, z) does for fit z exactly what block (x, y) does in that fit's own
// launch -- so every fit's weights, losses and log-probs are bit-identical to a solo run of the same fit.
//
// Data: each fit has its own device-resident dataset per slot (train / valid / test: X int64 [rows, S], y int64
// [rows]); one gather launch copies batch [row0, row0 + B) of every fit into the staging buffers the recorded
// programs read, and publishes {row0, batch index} as device scalars: lsm_nll writes the batch's log-probs at row
// row0 of the fit's epoch-long output buffer and the loss into the fit's per-batch loss history, so a whole epoch
// needs no host synchronisation and no per-step device-to-device copies.
#include <map>
#include <tuple>

#include <algorithm>

#include "gemm_jobs.hpp"
#include "tf_plan.hpp"

namespace slnlp {

constexpr int LS_MAX_FITS = 64;
constexpr int LS_SLOTS = 4;

// hooks of the RNN plan (its struct is private to rnn_plan.hip)
int rnn_ls_prepare(slnlp_rnn_plan* pl, int B, hipStream_t st);
int rnn_ls_record(slnlp_rnn_plan* pl, const int64_t* X, const int64_t* y, const int64_t* len, int B, int train, float momentum,
                  float max_norm, const LsAdam* adam, float* exp_avg_sq, hipStream_t st);
void rnn_ls_outputs(slnlp_rnn_plan* pl, float* logp, float* loss, const int* dyn);
void rnn_ls_replayed(slnlp_rnn_plan* pl, int B, int train);
const slnlp_rnn_config* rnn_ls_cfg(slnlp_rnn_plan* pl);

struct GatherArgs {
    const int64_t* const* X;     // [K] dataset pointers (device table)
    const int64_t* const* y;
    const int64_t* const* len;   // RNN fits: lengths per sequence (nullptr for the Transformer, which reads the pad ids)
    int64_t* const* Xst;         // [K] staging pointers (device table)
    int64_t* const* yst;
    int64_t* const* Lst;
    int* dyn;                    // {row0, batch index}
    int S, row0, B, step;
};

__global__ __launch_bounds__(256) void ls_gather_kernel(const GatherArgs a) {
    const int f = blockIdx.z;
    const int64_t* X = a.X[f] + (long)a.row0 * a.S;
    int64_t* Xs = a.Xst[f];
    const int n = a.B * a.S;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) Xs[i] = X[i];
    if (blockIdx.x == 0) {
        const int64_t* y = a.y[f] + a.row0;
        int64_t* ys = a.yst[f];
        for (int i = threadIdx.x; i < a.B; i += 256) ys[i] = y[i];
        if (a.len) {
            const int64_t* l = a.len[f] + a.row0;
            int64_t* ls = a.Lst[f];
            for (int i = threadIdx.x; i < a.B; i += 256) ls[i] = l[i];
        }
        if (f == 0 && threadIdx.x == 0) {
            a.dyn[0] = a.row0;
            a.dyn[1] = a.step;
        }
    }
}

struct MergedOp {
    const void* fn;
    dim3 grid, block;
    size_t lds;
    int kind;
    std::vector<char> arg0;      // by-value first kernel argument (any fit's; unused when the table is set)
    void* tab;                   // device: K packs, or the concatenated job table
    int* blockmap;               // device: block -> job (group kinds)
    const char* what;
};
struct Program {
    std::vector<MergedOp> ops;
};

}  // namespace slnlp

using namespace slnlp;

// What the driver needs from one fit, whatever its plan type.
struct LsFit {
    void* plan;
    int (*prepare)(void* plan, int B, hipStream_t st);      // work that must stay outside a recorded program (memsets, re-splits)
    int (*record)(void* plan, const int64_t* X, const int64_t* y, const int64_t* len, int B, int train, float momentum, float max_norm,
                  const LsAdam* adam, float* exp_avg_sq, hipStream_t st);   // the ordinary step code, run under a Recorder
    void (*outputs)(void* plan, float* logp, float* loss, const int* dyn);
    void (*replayed)(void* plan, int B, int train);         // host bookkeeping after the step's launches were issued
};

struct LockstepGroup {
    std::vector<LsFit> fits;
    bool has_len = false;
    int K = 0, S = 0, maxB = 0;
    int destroy_sync = 1;                           // slnlp_*_lockstep_set_destroy_sync (launch.hpp)
    char* ws = nullptr;
    size_t ws_bytes = 0, ws_used = 0, ws_mark = 0;  // ws_mark: bump position after ls_init (staging + pointer tables)
    std::vector<int64_t*> Xst, yst, Lst;            // per-fit staging (device)
    int64_t** d_Xst = nullptr;                      // device tables of the above
    int64_t** d_yst = nullptr;
    int64_t** d_Lst = nullptr;
    int* dyn = nullptr;
    struct Slot {
        bool set = false;
        int64_t rows = 0;
        const int64_t** d_X = nullptr;              // device tables [K]
        const int64_t** d_y = nullptr;
        const int64_t** d_len = nullptr;
        std::vector<const int64_t*> hX, hy, hlen;   // host copies of the tables (re-uploaded when the table space is reclaimed)
        std::vector<float*> logp, loss;             // per-fit output buffers (device, caller-owned)
    } slot[LS_SLOTS];
    std::map<std::tuple<int, int, int>, Program> programs;   // (slot, B, train)
    // the optimizer constants a train program baked into its recorded update launches: every later call must pass the same
    bool have_opt = false;
    float opt_momentum = 0.f, opt_max_norm = 0.f;
    // slnlp_*_lockstep_set_adam: the group's train programs end in the fused Adam update instead of SGD-momentum
    bool use_adam = false;
    LsAdam adam{};
    std::vector<float*> v2;                         // per-fit exp_avg_sq arenas (device, caller-owned)

    void* take(size_t bytes) {
        ws_used = (ws_used + 255) & ~(size_t)255;
        if (ws_used + bytes > ws_bytes) return nullptr;
        void* p = ws + ws_used;
        ws_used += bytes;
        return p;
    }
};
struct slnlp_tf_lockstep : LockstepGroup {};
struct slnlp_rnn_lockstep : LockstepGroup {};

static int upload(LockstepGroup* ls, const void* host, size_t bytes, void** dev, hipStream_t st) {
    void* d = ls->take(bytes);
    SLNLP_CHECK_ARG(d, "lockstep: workspace exhausted (%zu of %zu bytes used, %zu more needed)", ls->ws_used, ls->ws_bytes, bytes);
    // synchronous with respect to the host buffer (pageable memory); ordered on `st` before the launches that read it
    if (hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        set_error("lockstep: table upload failed: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    *dev = d;
    return 0;
}

// Tables of one program are staged in ONE host blob and uploaded with one copy; ops hold offsets until then.
struct Blob {
    std::vector<char> bytes;
    size_t add(const void* p, size_t n) {
        const size_t at = (bytes.size() + 255) & ~(size_t)255;
        bytes.resize(at + n);
        memcpy(bytes.data() + at, p, n);
        return at;
    }
};

// merged fp32-operand GEMM launches re-tile their 16-column jobs to 64 columns from this many workgroups on
// (env SLNLP_LS_RETILE_MIN: tuning / tests; 0 = never)
static long gemm_group_retile_min() {
    static const long v = [] { const char* e = getenv("SLNLP_LS_RETILE_MIN"); const long x = e ? atol(e) : 512; return x > 0 ? x : (1L << 40); }();
    return v;
}

static int merge(LockstepGroup* ls, std::vector<Recorder>& recs, Program& prog, hipStream_t st) {
    const int K = (int)recs.size();
    Blob blob;
    const size_t nops = recs[0].ops.size();
    for (int f = 1; f < K; ++f)
        SLNLP_CHECK_ARG(recs[f].ops.size() == nops, "lockstep: fit %d recorded %zu launches, fit 0 %zu -- not the same shape", f,
                        recs[f].ops.size(), nops);
    for (size_t i = 0; i < nops; ++i) {
        const RecOp& o0 = recs[0].ops[i];
        for (int f = 1; f < K; ++f) {
            const RecOp& o = recs[f].ops[i];
            SLNLP_CHECK_ARG(o.fn == o0.fn && o.kind == o0.kind && o.block.x == o0.block.x && o.lds == o0.lds &&
                                o.args.size() == o0.args.size() && (o.kind != REC_Z || (o.grid.x == o0.grid.x && o.grid.y == o0.grid.y)),
                            "lockstep: call site %zu (%s) differs between fit 0 and fit %d", i, o0.what, f);
        }
        MergedOp m;
        m.fn = o0.fn; m.block = o0.block; m.lds = o0.lds; m.kind = o0.kind; m.arg0 = o0.args; m.what = o0.what;
        m.tab = nullptr; m.blockmap = nullptr;
        if (o0.kind == REC_Z) {
            SLNLP_CHECK_ARG(o0.grid.z == 1, "lockstep: call site %zu (%s) already uses grid.z", i, o0.what);
            std::vector<char> tab(o0.args.size() * K);
            for (int f = 0; f < K; ++f) memcpy(tab.data() + (size_t)f * o0.args.size(), recs[f].ops[i].args.data(), o0.args.size());
            m.tab = (void*)(blob.add(tab.data(), tab.size()) + 1);          // offset + 1 until the blob is uploaded
            m.grid = dim3(o0.grid.x, o0.grid.y, K);
            {   // the recurrent forward step splits its K loop over two thread groups while a launch is a latency chain; K fits side
                // by side fill the chip and take the one-group twin (same bits: gemm.hip)
                int threads = 0;
                dim3 gr = o0.grid;
                if (const void* twin = rnn_step_fwd_for_blocks(o0.fn, o0.args.data(), K, &threads, &gr)) {
                    m.fn = twin;
                    m.block = dim3(threads);
                    m.grid = dim3(gr.x, gr.y, K);
                }
            }
            {   // the decoder's B-row products pick their tile for K fits' worth of workgroups (same bits whatever the tile: gemm_rows.hip)
                dim3 gr;
                size_t lds = 0;
                if (const void* twin = gemm_rows_for_fits(o0.fn, o0.args.data(), K, &gr, &lds)) {
                    m.fn = twin;
                    m.grid = dim3(gr.x, gr.y, K);
                    m.lds = lds;
                }
            }
        } else if (o0.kind == REC_PLANE_GROUP) {
            std::vector<PlaneJob> jobs;
            std::vector<int> map;
            for (int f = 0; f < K; ++f) {
                const PlaneGroupParams* P = reinterpret_cast<const PlaneGroupParams*>(recs[f].ops[i].args.data());
                for (int j = 0; j < P->njobs; ++j) jobs.push_back(P->job[j]);
            }
            // The merged launch has K times the tiles of a solo one: once it holds enough large tiles to fill the chip a few
            // times it takes them (fewer operand bytes per FLOP through L2 -> LDS, gemm_planes.hip).  The K partition of every
            // job stays, so each fit's results keep the bits of its solo launch whatever the geometry.
            plane_merge_geometry(o0.fn, jobs.data(), (int)jobs.size(), &m.fn, &m.lds);
            const int placed = plane_merge_place(m.fn, jobs.data(), (int)jobs.size(), map);
            SLNLP_CHECK_ARG(placed >= 0, "lockstep: call site %zu (%s) merges %zu plane-GEMM jobs -- more than a block-map entry can name", i, o0.what,
                            jobs.size());
            if (!placed) {
                for (size_t j = 0; j < jobs.size(); ++j) {          // fp8 launches: a job's blocks in a row, jobs on multiples of 8
                    PlaneJob& job = jobs[j];
                    const int blocks = job.tiles_x * job.tiles_y * job.nks, padded = (blocks + 7) & ~7;
                    job.block_begin = (int)map.size();
                    map.insert(map.end(), padded, (int)j);
                }
            }
            m.tab = (void*)(blob.add(jobs.data(), jobs.size() * sizeof(PlaneJob)) + 1);
            m.blockmap = (int*)(blob.add(map.data(), map.size() * sizeof(int)) + 1);
            m.grid = dim3((unsigned)map.size());
        } else {
            std::vector<GemmJob> jobs;
            std::vector<int> map;
            long blocks = 0;
            for (int f = 0; f < K; ++f) {
                const GemmGroupParams* P = reinterpret_cast<const GemmGroupParams*>(recs[f].ops[i].args.data());
                for (int j = 0; j < P->njobs; ++j) {
                    GemmJob job;
                    job.p = P->job[j]; job.variant = P->variant[j]; job.gx = P->gx[j]; job.gy = P->gy[j];
                    blocks += (long)job.gx * job.gy * (job.p.a.batch > 1 ? job.p.a.batch : 1);
                    jobs.push_back(job);
                }
            }
            // A solo fit runs its 50-row products on 16-column tiles (4x the workgroups: the launch is a latency chain on a
            // few CUs).  The merged launch of many fits is throughput-bound instead, and every 16-column workgroup converts
            // the job's whole A tile again: once the launch fills the chip anyway, the narrow jobs take 64-column tiles -- a
            // quarter of the A conversions and LDS writes per output.  Same K order per output element (the K loop and its
            // KS split do not depend on the tile width): each fit keeps the bits of its solo launch.
            if (blocks >= gemm_group_retile_min())
                for (GemmJob& job : jobs)
                    if ((job.variant & 1) && job.p.a.N > 16) { job.variant -= 1; job.gx = (job.p.a.N + 63) / 64; }
            // jobs with the longest K loops first (the fits' 8-step data gradients before their one-step weight gradients): the launch
            // ends when the last long workgroup does, so none of them should start late.  (Job order has no bearing on results.)
            std::stable_sort(jobs.begin(), jobs.end(), [](const GemmJob& a, const GemmJob& b) { return a.p.a.K > b.p.a.K; });
            for (size_t j = 0; j < jobs.size(); ++j) {
                GemmJob& job = jobs[j];
                job.block_begin = (int)map.size();
                map.insert(map.end(), job.gx * job.gy * (job.p.a.batch > 1 ? job.p.a.batch : 1), (int)j);
            }
            {   // thread groups per workgroup for the MERGED size: a fit's own 96-workgroup launch is a latency chain and splits its
                // K loop over two groups, fifteen of them fill the chip and run one group (gemm.hip gemm_tile: same bits either way)
                int longest = 0;
                for (const GemmJob& job : jobs) longest = std::max(longest, (job.p.a.K + 63) / 64);
                const int prec = (o0.fn == gemm_group_kernel_ptr(3, 1) || o0.fn == gemm_group_kernel_ptr(3, 2)) ? 3 : 1;
                const int ks = gemm_group_ks((int)map.size(), longest);
                m.fn = gemm_group_kernel_ptr(prec, ks);
                m.block = dim3(256 * ks);
            }
            m.tab = (void*)(blob.add(jobs.data(), jobs.size() * sizeof(GemmJob)) + 1);
            m.blockmap = (int*)(blob.add(map.data(), map.size() * sizeof(int)) + 1);
            m.grid = dim3((unsigned)map.size());
        }
        prog.ops.push_back(std::move(m));
    }
    void* base = nullptr;
    SLNLP_TRY(upload(ls, blob.bytes.data(), blob.bytes.size(), &base, st));
    for (MergedOp& m : prog.ops) {
        if (m.tab) m.tab = (char*)base + ((size_t)m.tab - 1);
        if (m.blockmap) m.blockmap = (int*)((char*)base + ((size_t)m.blockmap - 1));
    }
    return 0;
}

static int replay(const Program& prog, hipStream_t st) {
    for (const MergedOp& m : prog.ops) {
        void* tab = m.tab;
        void* bm = m.blockmap;
        void* args[3] = {(void*)m.arg0.data(), &tab, &bm};
        if (hipLaunchKernel(m.fn, m.grid, m.block, args, m.lds, st) != hipSuccess) {
            set_error("lockstep: launch of %s failed: %s", m.what, hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ generic driver ----
static int ls_init(LockstepGroup* ls, int B, int S, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    const int K = (int)ls->fits.size();
    ls->K = K; ls->S = S; ls->maxB = B;
    ls->ws = (char*)workspace; ls->ws_bytes = (size_t)workspace_bytes;
    for (int f = 0; f < K; ++f) {
        int64_t* x = (int64_t*)ls->take((size_t)B * S * sizeof(int64_t));
        int64_t* y = (int64_t*)ls->take((size_t)B * sizeof(int64_t));
        int64_t* l = ls->has_len ? (int64_t*)ls->take((size_t)B * sizeof(int64_t)) : nullptr;
        SLNLP_CHECK_ARG(x && y && (l || !ls->has_len), "lockstep_create: workspace too small");
        ls->Xst.push_back(x);
        ls->yst.push_back(y);
        ls->Lst.push_back(l);
    }
    ls->dyn = (int*)ls->take(64);
    SLNLP_CHECK_ARG(ls->dyn, "lockstep_create: workspace too small");
    SLNLP_TRY(upload(ls, ls->Xst.data(), K * sizeof(void*), (void**)&ls->d_Xst, st));
    SLNLP_TRY(upload(ls, ls->yst.data(), K * sizeof(void*), (void**)&ls->d_yst, st));
    if (ls->has_len) SLNLP_TRY(upload(ls, ls->Lst.data(), K * sizeof(void*), (void**)&ls->d_Lst, st));
    ls->ws_mark = ls->ws_used;
    return 0;
}

static int upload_slot_tables(LockstepGroup* ls, LockstepGroup::Slot& s, hipStream_t st) {
    SLNLP_TRY(upload(ls, s.hX.data(), ls->K * sizeof(void*), (void**)&s.d_X, st));
    SLNLP_TRY(upload(ls, s.hy.data(), ls->K * sizeof(void*), (void**)&s.d_y, st));
    if (ls->has_len) SLNLP_TRY(upload(ls, s.hlen.data(), ls->K * sizeof(void*), (void**)&s.d_len, st));
    return 0;
}

static void ls_destroy(LockstepGroup* ls) {
    destroy_sync(ls->destroy_sync);    // tables live in caller memory that may be freed next
    for (LsFit& f : ls->fits) f.outputs(f.plan, nullptr, nullptr, nullptr);
}

static int ls_set_data(LockstepGroup* ls, int slot, const int64_t* const* X, const int64_t* const* y, const int64_t* const* len,
                       int64_t rows, float* const* logp, float* const* loss, hipStream_t st) {
    SLNLP_CHECK_ARG(ls && slot >= 0 && slot < LS_SLOTS && X && y && logp && loss && rows > 0 && (len || !ls->has_len),
                    "lockstep_set_data: bad arguments");
    LockstepGroup::Slot& s = ls->slot[slot];
    for (auto it = ls->programs.begin(); it != ls->programs.end();)     // programs of this slot baked the old output pointers
        it = std::get<0>(it->first) == slot ? ls->programs.erase(it) : std::next(it);
    s.rows = rows;
    s.logp.assign(logp, logp + ls->K);
    s.loss.assign(loss, loss + ls->K);
    s.hX.assign(X, X + ls->K);
    s.hy.assign(y, y + ls->K);
    if (ls->has_len) s.hlen.assign(len, len + ls->K);
    s.set = true;
    // the baked optimizer constants belong to the TRAIN programs: once none is left a new momentum / max_norm is welcome again,
    // whatever eval programs of other slots are still cached
    bool any_train = false;
    for (const auto& kv : ls->programs) any_train = any_train || std::get<2>(kv.first) != 0;
    if (!any_train) ls->have_opt = false;
    if (ls->programs.empty()) {
        // no recorded program refers to the table space any more: hand it all back (a long-lived group that keeps getting new
        // data would otherwise run the bump allocator dry) and put the slots' pointer tables at its start again.  Ordered on
        // `st` behind every launch that read the old tables.
        ls->ws_used = ls->ws_mark;
        ls->have_opt = false;
        for (int k = 0; k < LS_SLOTS; ++k)
            if (ls->slot[k].set) SLNLP_TRY(upload_slot_tables(ls, ls->slot[k], st));
        return 0;
    }
    return upload_slot_tables(ls, s, st);
}

// One lockstep step of every fit on rows [row0, row0 + B) of slot `slot`: train != 0 -> forward + criterion + backward +
// clip + SGD (what slnlp_{tf,rnn}_train_step does for one fit), else an eval-mode forward + criterion.
static int ls_step(LockstepGroup* ls, int slot, int64_t row0, int B, int step_index, int train, float momentum, float max_norm,
                   hipStream_t st) {
    SLNLP_CHECK_ARG(ls && slot >= 0 && slot < LS_SLOTS && ls->slot[slot].set, "lockstep_step: slot %d has no data", slot);
    LockstepGroup::Slot& s = ls->slot[slot];
    SLNLP_CHECK_ARG(B > 0 && B <= ls->maxB && row0 >= 0 && row0 + B <= s.rows, "lockstep_step: rows [%ld, %ld) outside 0..%ld or batch > %d",
                    (long)row0, (long)(row0 + B), (long)s.rows, ls->maxB);
    StepScope scope(st);               // one kernel sequence per device (launch.hpp)
    SLNLP_TRY(scope.rc);
    if (train) {
        // momentum / max_norm are baked by value into the recorded update launches: a replay cannot change them
        SLNLP_CHECK_ARG(!ls->have_opt || (ls->opt_momentum == momentum && ls->opt_max_norm == max_norm),
                        "lockstep_step: momentum %g / max_norm %g differ from the values the group's train programs were recorded with "
                        "(%g / %g); set new data (which drops the programs) or use another group",
                        momentum, max_norm, ls->opt_momentum, ls->opt_max_norm);
    }
    for (LsFit& f : ls->fits) SLNLP_TRY(f.prepare(f.plan, B, st));
    const auto key = std::make_tuple(slot, B, train ? 1 : 0);
    auto it = ls->programs.find(key);
    if (it == ls->programs.end()) {
        std::vector<Recorder> recs(ls->K);
        int rc = 0;
        for (int f = 0; f < ls->K && !rc; ++f) {
            LsFit& fit = ls->fits[f];
            fit.outputs(fit.plan, s.logp[f], s.loss[f], ls->dyn);
            set_recorder(&recs[f]);
            rc = fit.record(fit.plan, ls->Xst[f], ls->yst[f], ls->Lst[f], B, train, momentum, max_norm, ls->use_adam ? &ls->adam : nullptr,
                            ls->use_adam ? ls->v2[f] : nullptr, st);
            set_recorder(nullptr);
        }
        if (rc) return rc;
        Program prog;
        SLNLP_TRY(merge(ls, recs, prog, st));
        it = ls->programs.emplace(key, std::move(prog)).first;
        if (train) { ls->have_opt = true; ls->opt_momentum = momentum; ls->opt_max_norm = max_norm; }
    }
    GatherArgs g;
    g.X = s.d_X; g.y = s.d_y; g.len = ls->has_len ? s.d_len : nullptr;
    g.Xst = ls->d_Xst; g.yst = ls->d_yst; g.Lst = ls->d_Lst; g.dyn = ls->dyn;
    g.S = ls->S; g.row0 = (int)row0; g.B = B; g.step = step_index;
    int gx = (B * ls->S + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(ls_gather_kernel, dim3(gx, 1, ls->K), dim3(256), 0, st, g);
    SLNLP_CHECK_LAUNCH("lockstep gather");
    SLNLP_TRY(replay(it->second, st));
    for (LsFit& f : ls->fits) f.replayed(f.plan, B, train);
    return 0;
}

// One pass over slot `slot` in dataset order, batches of `batch` rows (the last one may be shorter).
static int ls_epoch(LockstepGroup* ls, int slot, int batch, int train, float momentum, float max_norm, hipStream_t st) {
    SLNLP_CHECK_ARG(ls && slot >= 0 && slot < LS_SLOTS && ls->slot[slot].set && batch > 0, "lockstep_epoch: bad arguments");
    const int64_t rows = ls->slot[slot].rows;
    int step = 0;
    for (int64_t r = 0; r < rows; r += batch, ++step) {
        const int B = (int)(rows - r < batch ? rows - r : batch);
        SLNLP_TRY(ls_step(ls, slot, r, B, step, train, momentum, max_norm, st));
    }
    return 0;
}

// Train with clip_grad_norm_ + Adam from now on (exp_avg = each plan's momentum arena, exp_avg_sq[f] = an arena-shaped buffer of
// the caller's, zero before the first step; the step count is each plan's scalars[2]).  Recorded train programs are dropped.
static int ls_set_adam(LockstepGroup* ls, float* const* exp_avg_sq, float beta1, float beta2, float eps, float weight_decay) {
    SLNLP_CHECK_ARG(ls && exp_avg_sq, "lockstep_set_adam: null argument");
    for (int f = 0; f < ls->K; ++f) SLNLP_CHECK_ARG(exp_avg_sq[f], "lockstep_set_adam: null exp_avg_sq for fit %d", f);
    for (auto it = ls->programs.begin(); it != ls->programs.end();)
        it = std::get<2>(it->first) ? ls->programs.erase(it) : std::next(it);
    ls->v2.assign(exp_avg_sq, exp_avg_sq + ls->K);
    ls->adam = LsAdam{beta1, beta2, eps, weight_decay};
    ls->use_adam = true;
    ls->have_opt = false;
    // (the dropped train programs' tables stay in the bump-allocated workspace until every program is gone -- set_data on the
    //  last slot hands the space back; set_adam is called once per group, before its first train epoch, when there is nothing to drop)
    return 0;
}

static int ls_num_launches(LockstepGroup* ls, int slot, int B, int train) {
    if (!ls) return -1;
    auto it = ls->programs.find(std::make_tuple(slot, B, train ? 1 : 0));
    return it == ls->programs.end() ? -1 : (int)it->second.ops.size() + 1;
}

// ---------------------------------------------------------------------------------------------- Transformer hooks ----
static int tf_prepare(void* p, int B, hipStream_t st) {
    slnlp_tf_plan* pl = (slnlp_tf_plan*)p;
    SLNLP_TRY(pl->prepare_planes(B, st));      // re-zero plane padding when B changes
    SLNLP_TRY(pl->ensure_wplanes(st));         // never part of a recorded program: the update kernel keeps the planes current
    return pl->ensure_wq(st);                  // precision 8: re-quantised weights, likewise outside the program
}
static int tf_record(void* p, const int64_t* X, const int64_t* y, const int64_t*, int B, int train, float momentum, float max_norm,
                     const LsAdam* adam, float* exp_avg_sq, hipStream_t st) {
    slnlp_tf_plan* pl = (slnlp_tf_plan*)p;
    SLNLP_TRY(pl->forward_impl(X, y, B, train, nullptr, st));
    if (!train) return 0;
    SLNLP_TRY(slnlp_tf_backward(pl, st));
    if (adam) return slnlp_tf_optim_adam(pl, exp_avg_sq, adam->beta1, adam->beta2, adam->eps, adam->weight_decay, max_norm, st);
    return slnlp_tf_optim(pl, momentum, max_norm, st);
}
static void tf_outputs(void* p, float* logp, float* loss, const int* dyn) {
    slnlp_tf_plan* pl = (slnlp_tf_plan*)p;
    pl->ls_logp = logp; pl->ls_loss = loss; pl->ls_dyn = dyn;
}
static void tf_replayed(void* p, int B, int train) {
    slnlp_tf_plan* pl = (slnlp_tf_plan*)p;
    pl->last_B = B;
    pl->last_p = train ? pl->cfg.dropout : 0.f;
    if (train) pl->params_stepped();
}

// ------------------------------------------------------------------------------------------------------ RNN hooks ----
static int rnn_prepare(void* p, int B, hipStream_t st) { return rnn_ls_prepare((slnlp_rnn_plan*)p, B, st); }
static int rnn_record(void* p, const int64_t* X, const int64_t* y, const int64_t* len, int B, int train, float momentum, float max_norm,
                      const LsAdam* adam, float* exp_avg_sq, hipStream_t st) {
    return rnn_ls_record((slnlp_rnn_plan*)p, X, y, len, B, train, momentum, max_norm, adam, exp_avg_sq, st);
}
static void rnn_outputs(void* p, float* logp, float* loss, const int* dyn) { rnn_ls_outputs((slnlp_rnn_plan*)p, logp, loss, dyn); }
static void rnn_replayed(void* p, int B, int train) { rnn_ls_replayed((slnlp_rnn_plan*)p, B, train); }

extern "C" {

int64_t slnlp_tf_lockstep_workspace_bytes(const slnlp_tf_config* cfg, int K) {
    if (!cfg || K < 1 || K > LS_MAX_FITS) return -1;
    // staging + pointer tables + argument / job tables of the cached programs (<= 8: train / eval x full / tail batch x
    // data slots).  Per fit a program holds, for each of its ~45 + 55 N call sites, either a <= 256-byte argument pack, or
    // -- the 8 N plane-GEMM sites -- up to 2 jobs and a block map of 4 B per workgroup (the dgrad's tiles plus the wgrad's
    // tiles x split-K, jobs padded to multiples of 8 blocks), or -- ~16 N + 4 fp32-operand group sites -- up to 4 jobs and
    // a block map over the decoder's B-row tiles.  x 1.5 for alignment (256 B per table) and slack.
    const size_t staging = (size_t)K * ((size_t)cfg->B * cfg->S + cfg->B + 64) * sizeof(int64_t);
    const size_t N = (size_t)cfg->N, E = (size_t)cfg->E, F = (size_t)cfg->F, M = (size_t)cfg->B * cfg->S;
    auto cd = [](size_t a, size_t b) { return (a + b - 1) / b; };
    const size_t wide = 3 * E > F ? 3 * E : F, inner = E > F ? E : F;
    const size_t plane_blocks = cd(M, 64) * cd(wide, 64) + cd(wide, 64) * cd(inner, 64) * MAX_SPLITK + 16;
    const size_t brow = cd((size_t)cfg->B, 64) * cd(wide > (size_t)cfg->Vt ? wide : (size_t)cfg->Vt, 16) * (size_t)cfg->H;
    const size_t per_fit = (45 + 55 * N) * 256 + 8 * N * (2 * sizeof(PlaneJob) + 4 * plane_blocks) +
                           (16 * N + 4) * (4 * sizeof(GemmJob) + 4 * 4 * brow);
    return (int64_t)(staging + 65536 + 8 * ((45 + 55 * N) * 512 + (size_t)K * per_fit * 3 / 2));
}

int slnlp_tf_lockstep_create(slnlp_tf_plan** plans, int K, void* workspace, int64_t workspace_bytes, void* stream,
                             slnlp_tf_lockstep** out) {
    SLNLP_CHECK_ARG(plans && out && K >= 1 && K <= LS_MAX_FITS, "lockstep_create: 1..%d plans", LS_MAX_FITS);
    SLNLP_CHECK_ARG(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "lockstep_create: workspace must be 256-byte aligned");
    const slnlp_tf_config& c0 = plans[0]->cfg;
    for (int f = 0; f < K; ++f) {
        SLNLP_CHECK_ARG(plans[f], "lockstep_create: null plan %d", f);
        const slnlp_tf_config& c = plans[f]->cfg;
        SLNLP_CHECK_ARG(c.E == c0.E && c.H == c0.H && c.N == c0.N && c.F == c0.F && c.Vs == c0.Vs && c.Vt == c0.Vt && c.B == c0.B &&
                            c.S == c0.S && c.precision == c0.precision && (c.dropout > 0.f) == (c0.dropout > 0.f),
                        "lockstep_create: plan %d does not have the shape of plan 0 (lr, dropout rate and seed may differ; "
                        "dropout on/off may not)", f);
        for (int g = 0; g < f; ++g) SLNLP_CHECK_ARG(plans[g] != plans[f], "lockstep_create: plan %d listed twice", f);
    }
    slnlp_tf_lockstep* ls = new slnlp_tf_lockstep();
    for (int f = 0; f < K; ++f) ls->fits.push_back(LsFit{plans[f], tf_prepare, tf_record, tf_outputs, tf_replayed});
    const int rc = ls_init(ls, c0.B, c0.S, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc) { delete ls; return rc; }
    *out = ls;
    return 0;
}

void slnlp_tf_lockstep_destroy(slnlp_tf_lockstep* ls) {
    if (!ls) return;
    ls_destroy(ls);
    delete ls;
}

// Slot `slot` of every fit: dataset X[f] int64 [rows, S] / y[f] int64 [rows] and where its outputs go: logp[f] float
// [rows, Vt] (log-probs of every batch of a pass) and loss[f] float [ceil(rows / batch)] (loss per batch).
int slnlp_tf_lockstep_set_data(slnlp_tf_lockstep* ls, int slot, const int64_t* const* X, const int64_t* const* y, int64_t rows,
                               float* const* logp, float* const* loss, void* stream) {
    return ls_set_data(ls, slot, X, y, nullptr, rows, logp, loss, (hipStream_t)stream);
}
int slnlp_tf_lockstep_step(slnlp_tf_lockstep* ls, int slot, int64_t row0, int B, int step_index, int train, float momentum,
                           float max_norm, void* stream) {
    return ls_step(ls, slot, row0, B, step_index, train, momentum, max_norm, (hipStream_t)stream);
}
int slnlp_tf_lockstep_epoch(slnlp_tf_lockstep* ls, int slot, int batch, int train, float momentum, float max_norm, void* stream) {
    return ls_epoch(ls, slot, batch, train, momentum, max_norm, (hipStream_t)stream);
}
int slnlp_tf_lockstep_num_launches(slnlp_tf_lockstep* ls, int slot, int B, int train) { return ls_num_launches(ls, slot, B, train); }
int slnlp_tf_lockstep_set_adam(slnlp_tf_lockstep* ls, float* const* exp_avg_sq, float beta1, float beta2, float eps, float weight_decay) {
    return ls_set_adam(ls, exp_avg_sq, beta1, beta2, eps, weight_decay);
}
int slnlp_tf_lockstep_set_destroy_sync(slnlp_tf_lockstep* ls, int on) {
    SLNLP_CHECK_ARG(ls, "lockstep_set_destroy_sync: null group");
    ls->destroy_sync = on ? 1 : 0;
    return 0;
}

// ---- the same for K EncoderDecoder{LSTM,GRU}Attn fits (rnn_plan.hip); a slot also carries the sequence lengths
int64_t slnlp_rnn_lockstep_workspace_bytes(const slnlp_rnn_config* cfg, int K) {
    if (!cfg || K < 1 || K > LS_MAX_FITS) return -1;
    // call sites: one fused step per timestep forward, a cell + a grouped recurrent dgrad per timestep backward, ~24 more per
    // layer and ~64 around them.  Table bytes per site and fit: a z-pack is <= 320 B, the recurrent dgrad group up to 8 jobs
    // of ~400 B + a block map: budget 6 KiB on average, for 6 programs (train / eval x full / tail batch, test).
    const size_t staging = (size_t)K * ((size_t)cfg->B * cfg->S + 2 * cfg->B + 64) * sizeof(int64_t);
    const size_t sites = 64 + (size_t)cfg->N * (3 * (size_t)cfg->S + 24);
    return (int64_t)(staging + 65536 + 6 * sites * (size_t)K * 6144);
}

int slnlp_rnn_lockstep_create(slnlp_rnn_plan** plans, int K, void* workspace, int64_t workspace_bytes, void* stream,
                              slnlp_rnn_lockstep** out) {
    SLNLP_CHECK_ARG(plans && out && K >= 1 && K <= LS_MAX_FITS, "rnn_lockstep_create: 1..%d plans", LS_MAX_FITS);
    SLNLP_CHECK_ARG(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "rnn_lockstep_create: workspace must be 256-byte aligned");
    for (int f = 0; f < K; ++f) SLNLP_CHECK_ARG(plans[f], "rnn_lockstep_create: null plan %d", f);
    const slnlp_rnn_config& c0 = *rnn_ls_cfg(plans[0]);
    for (int f = 0; f < K; ++f) {
        const slnlp_rnn_config& c = *rnn_ls_cfg(plans[f]);
        SLNLP_CHECK_ARG(c.lstm == c0.lstm && c.E == c0.E && c.Hd == c0.Hd && c.N == c0.N && c.Vs == c0.Vs && c.Vt == c0.Vt && c.B == c0.B &&
                            c.S == c0.S && c.precision == c0.precision && (c.dropout > 0.f) == (c0.dropout > 0.f),
                        "rnn_lockstep_create: plan %d does not have the shape of plan 0 (lr, dropout rate and seed may differ; "
                        "dropout on/off may not)", f);
        for (int g = 0; g < f; ++g) SLNLP_CHECK_ARG(plans[g] != plans[f], "rnn_lockstep_create: plan %d listed twice", f);
    }
    slnlp_rnn_lockstep* ls = new slnlp_rnn_lockstep();
    ls->has_len = true;
    for (int f = 0; f < K; ++f) ls->fits.push_back(LsFit{plans[f], rnn_prepare, rnn_record, rnn_outputs, rnn_replayed});
    const int rc = ls_init(ls, c0.B, c0.S, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc) { delete ls; return rc; }
    *out = ls;
    return 0;
}

void slnlp_rnn_lockstep_destroy(slnlp_rnn_lockstep* ls) {
    if (!ls) return;
    ls_destroy(ls);
    delete ls;
}

int slnlp_rnn_lockstep_set_data(slnlp_rnn_lockstep* ls, int slot, const int64_t* const* X, const int64_t* const* y,
                                const int64_t* const* lengths, int64_t rows, float* const* logp, float* const* loss, void* stream) {
    SLNLP_CHECK_ARG(lengths, "rnn_lockstep_set_data: lengths are required");
    return ls_set_data(ls, slot, X, y, lengths, rows, logp, loss, (hipStream_t)stream);
}
int slnlp_rnn_lockstep_step(slnlp_rnn_lockstep* ls, int slot, int64_t row0, int B, int step_index, int train, float momentum,
                            float max_norm, void* stream) {
    return ls_step(ls, slot, row0, B, step_index, train, momentum, max_norm, (hipStream_t)stream);
}
int slnlp_rnn_lockstep_epoch(slnlp_rnn_lockstep* ls, int slot, int batch, int train, float momentum, float max_norm, void* stream) {
    return ls_epoch(ls, slot, batch, train, momentum, max_norm, (hipStream_t)stream);
}
int slnlp_rnn_lockstep_num_launches(slnlp_rnn_lockstep* ls, int slot, int B, int train) { return ls_num_launches(ls, slot, B, train); }
int slnlp_rnn_lockstep_set_adam(slnlp_rnn_lockstep* ls, float* const* exp_avg_sq, float beta1, float beta2, float eps, float weight_decay) {
    return ls_set_adam(ls, exp_avg_sq, beta1, beta2, eps, weight_decay);
}
int slnlp_rnn_lockstep_set_destroy_sync(slnlp_rnn_lockstep* ls, int on) {
    SLNLP_CHECK_ARG(ls, "rnn_lockstep_set_destroy_sync: null group");
    ls->destroy_sync = on ? 1 : 0;
    return 0;
}

}  // extern "C"
