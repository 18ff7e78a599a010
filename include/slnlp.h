/* slnlp.h -- C ABI of libslnlp.so, the MI355X (gfx950) hot path for
 * amorim-cleison/sign-language-nlp.
 *
 * The reference has no native code and no FFI: its hot path is the Python
 * call `module(**{"X","lengths","y"})` + criterion + backward + clip + SGD that
 * skorch issues per batch.  The entry points below are
 * what a binding for that path attaches to; each cites the reference interface
 * it replaces as /root/reference/<file>:<line>.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer
 *    into memory the caller owns (the library never allocates or frees
 *    persistent memory); `stream` is a hipStream_t passed as void*;
 *  - all floating-point tensors are fp32 row-major; token ids / labels /
 *    lengths are int64 exactly as `collate_data` builds them (helper.py:293-304);
 *  - activations are sequence-first like the reference: token row m = s*B + b;
 *  - return 0 on success, an SLNLP_ERR_* code otherwise (never aborts);
 *    `slnlp_last_error()` returns a thread-local message;
 *  - every launch is asynchronous on `stream` and hipGraph-capturable (no
 *    allocation, no synchronisation inside).
 */
#ifndef SLNLP_H
#define SLNLP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLNLP_OK 0
#define SLNLP_ERR_INVALID_ARG 1
#define SLNLP_ERR_LAUNCH 2
#define SLNLP_ERR_UNSUPPORTED 3

#define SLNLP_ABI_VERSION 1

const char* slnlp_last_error(void);
int slnlp_abi_version(void);

/* ------------------------------------------------------------------ GEMM --
 * C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ) on bf16 MFMA with fp32
 * accumulate.  Replaces every torch.nn.Linear / in_proj / out_proj matmul the
 * reference reaches through nn.Transformer (model/transformer.py:40-48,82-88)
 * and nn.LSTM/GRU/Linear (model/base/encoder_decoder_attn_bkp.py:95-100,
 * 186-200,297-299), and their autograd backward (dgrad / wgrad).
 *
 * Operand (i,k) lives at ptr + i*ld + k when *_kmajor, else at ptr + k*ld + i.
 *   forward   y = x W^T      : A=x   kmajor, B=W  kmajor
 *   dgrad     dx = dy W      : A=dy  kmajor, B=W  NOT kmajor (k = W row)
 *   wgrad     dW = dy^T x    : A=dy  NOT kmajor, B=x NOT kmajor (k = token)
 * precision: 1 = single bf16 pass (~5e-3 rel); 3 = split-bf16 hi/lo, three MFMA
 * passes (~2e-5 rel, the parity-grade default).
 * Epilogue order: +bias[n] -> activation (relu: 1 = ReLU, 2 = tanh) -> *gate -> dropout -> +resid.
 * gate_mode 0: C *= (gate > 0 ? gate_scale : 0)   (ReLU + inverted-dropout backward, gate = saved output)
 * gate_mode 1: C *= (1 - gate^2)                   (tanh backward, gate = saved tanh output)
 */
typedef struct slnlp_gemm_args {
    const float* A; int64_t lda; int32_t a_kmajor;
    const float* B; int64_t ldb; int32_t b_kmajor;
    float* C; int64_t ldc;
    int32_t M, N, K;
    const float* bias;            /* [N] or NULL */
    int32_t relu;                 /* max(x,0) after bias */
    const float* gate; int64_t ldg; float gate_scale; /* C *= gate>0 ? gate_scale : 0 (ReLU+dropout backward) */
    float drop_p; int32_t drop_site; const unsigned long long* rng; /* inverted dropout, mask regenerated from rng */
    const float* resid; int64_t ldr; /* added last; may alias C */
    float* rowsum_a;              /* [M] or NULL: sum_k A(m,k) (bias grad fused into wgrad) */
    int32_t precision;            /* 1 or 3 */
    int32_t gate_mode;            /* see above */
    /* Optional PRE-SPLIT operands (bf16 hi / lo planes, row-major, rows and columns zero-padded to
     * multiples of 64, row stride ld*_p elements).  When A_hi and B_hi are set the GEMM stages them by
     * LDS-DMA and does no conversion (A, B, lda, ldb are then ignored; *_lo required for precision 3).
     * C_hi / C_lo (optional): also emit the result as planes for the next GEMM. C may then be NULL. */
    const uint16_t* A_hi; const uint16_t* A_lo; int64_t lda_p;
    const uint16_t* B_hi; const uint16_t* B_lo; int64_t ldb_p;
    uint16_t* C_hi; uint16_t* C_lo; int64_t ldc_p;
    /* > 0: the dropout of this GEMM is per (row, head) instead of per element -- element (m, n) keeps or drops
     * with site element (m * (N / drop_head_dim) + n / drop_head_dim, 0).  That is nn.MultiheadAttention's
     * attention-weight dropout when there is a single key (decoder self-attention, tgt length 1): the softmax
     * weight is the scalar 1 per (row, head).  fp32-operand GEMMs only. */
    int32_t drop_head_dim;
    /* precision 8 -- "fp8 MFMA weights" (BASELINE.json configs[4]), forward products only, both operands k-major:
     * A_hi / B_hi are then OCP e4m3 BYTE planes (row strides lda_p / ldb_p in bytes; rows zero-padded to multiples of 64, K
     * to multiples of 128), contracted on the fp8 MFMA; col_scale[n] (optional) multiplies column n of the product -- the
     * per-row scale of a quantised weight matrix.  C_q8 (optional): also emit the result as an e4m3 plane, row stride ldc_p. */
    const float* col_scale;
    uint8_t* C_q8;
    /* batch > 1 (fp32-operand jobs of slnlp_gemm_group only): the job is `batch` GEMMs of this shape; GEMM z reads
     * A + z * batch_stride_a, B + z * batch_stride_b and writes C + z * batch_stride_c (resid, when set, moves with C) --
     * e.g. the per-head products of the decoder's cross-attention (attention_mem.hip).  0 or 1: a single GEMM. */
    int32_t batch;
    int64_t batch_stride_a, batch_stride_b, batch_stride_c;
} slnlp_gemm_args;

int slnlp_gemm(const slnlp_gemm_args* args, void* stream);
/* Grouped launch: up to 4 independent GEMMs (all with pre-split plane operands, or all with fp32 operands -- then
 * split_k / scratch are ignored) in ONE kernel launch -- e.g. the data gradient
 * and the weight gradient of one dY, which replace autograd's separate mm calls for nn.Linear
 * (transformer.py:40-48 -> torch).  split_k[i] > 1 (or NULL = all 1) divides job i's K loop over that many
 * workgroups per output tile; the partial tiles meet in `scratch` and are added in split order by the last
 * workgroup to arrive, so the result is deterministic (no float atomics).  scratch: at least
 * slnlp_gemm_group_scratch_bytes(...) bytes, 16-byte aligned, its first 16 KiB zero before the first use (the
 * library leaves them zero); one scratch buffer must not serve two launches that may run concurrently. */
int64_t slnlp_gemm_group_scratch_bytes(const slnlp_gemm_args* jobs, const int32_t* split_k, int njobs);
/* Output tile of the plane-GEMM launches: 0 = automatic (128 x 128 once a launch holds >= 200 of them -- merged lockstep
 * launches, the configs[4] shapes, cfg2's in_proj gradients -- 256 x 256 for forward launches with K >= 1024 that fill the
 * chip's rounds, else 64 x 64); forced: 64, 128 (128 x 128, 64-k stages), 12832 (128 x 128, 32-k stages), 256 (256 x 256,
 * 32-k stages).  A tuning / test knob: results do not depend on it (the K partition, hence every element's accumulation
 * order, is the same for every geometry). */
int slnlp_set_plane_tile(int tile);
/* Thread groups per workgroup of the fp32-operand GEMM launches (slnlp_gemm, fp32 jobs of slnlp_gemm_group, the plans' 50-row
 * products): 0 = automatic (two -- each walks one half of the K tiles -- for launches of <= 128 workgroups with at least four K
 * tiles; one for everything larger, merged lockstep launches included), 1 or 2 forced.  A tuning / test knob: the K sum is
 * defined as (first half of the tiles) + (second half) whichever way it is scheduled, so results do not depend on it. */
int slnlp_set_gemm_ks(int ks);
/* Tile of the fused recurrent forward timestep (slnlp_rnn_step_fwd) when ONE fit launches it: 1 (default) = 16 batch rows x 16
 * hidden units x all gates per workgroup -- 256 workgroups at B = 50, Hd = 512, two directions, 160 KB of operands each -- 0 = the
 * 64-row tile (64 workgroups of 256 KB) that merged lockstep launches keep.  A tuning / test knob: same K order, same cell
 * arithmetic per element, same bits.  Env: SLNLP_RNN_STEP_RT. */
int slnlp_set_rnn_step_tile(int rows16);
/* the same knob for precision-8 launches: 0 = automatic (128 x 128 once the launch holds >= 512 of them), 64 or 128 */
int slnlp_set_fp8_tile(int tile);
int slnlp_gemm_group(const slnlp_gemm_args* jobs, const int32_t* split_k, int njobs, void* scratch,
                     int64_t scratch_bytes, void* stream);
/* The gradient pair of one dY over plane operands -- wgrad: dW = dY^T x (A, B not k-major, rowsum_a = db), dgrad: dX = dY W (A k-major,
 * B not) -- launched the way the training plans launch it: the library picks the weight gradient's K-split and whether the two
 * share ONE grouped launch (the weight gradient's workgroups fill the CUs the data gradient leaves idle) or, when both are large,
 * take a launch each with the tile that suits each.  scratch as for slnlp_gemm_group with split factors up to 8.
 * slnlp_gemm_wd_plan reports the choice (geometry codes: 0 = 64 x 64, 1 = 128 x 128 / 64-k, 2 = 128 x 128 / 32-k, 3 = 256 x 256). */
int slnlp_gemm_wd(const slnlp_gemm_args* wgrad, const slnlp_gemm_args* dgrad, void* scratch, int64_t scratch_bytes, void* stream);
int slnlp_gemm_wd_plan(const slnlp_gemm_args* wgrad, const slnlp_gemm_args* dgrad, int32_t* split, int32_t* separate, int32_t* geo_wgrad,
                       int32_t* geo_dgrad);
/* The decoder's products: the reference decodes ONE target position (transformer.py:82-87), so every nn.Linear of its
 * decoder (and the generator, transformer.py:46-48,88) is y[B rows, N] = x[B rows, K] W[N, K]^T on a dependent chain.
 * x as k-major bf16 hi / lo planes (A_hi / A_lo / lda_p; rows zero-padded to multiples of 64, stride a multiple of 64 covering
 * K), the weight W as fp32 (B / ldb, k-major, K a multiple of 64): the kernel splits it in registers -- hi = bf16(w), lo =
 * bf16(w - hi), the bits a plane of W would hold -- so nobody maintains planes of these weights.  Each wave loads its MFMA
 * fragments straight from memory: no LDS staging, no barrier in the K loop.  Same epilogue fields as slnlp_gemm (bias, relu,
 * gate, dropout incl. drop_head_dim, resid, C and / or C_hi / C_lo); the K sum is per 64-k tile: partial products from zero, added
 * in tile order.  Meant for up to 64 rows (one block of rows; more work, the plans use the plane GEMM there). */
int slnlp_gemm_rows(const slnlp_gemm_args* args, void* stream);
/* The backward pair of such a product in ONE launch (autograd's two mm calls for nn.Linear at batch rows):
 *   dgrad: dX[B rows, Kin] = dY[B rows, Nout] W[Nout, Kin] (+ the slnlp_gemm epilogue: gate, dropout, residual, planes out) --
 *          A = dY planes k-major, B = W as fp32 (B / ldb), NOT k-major (m-major: k = W's row; Nout a multiple of 64, Kin of 4);
 *   wgrad: dW[Nout, Kin] = dY^T x, rowsum_a = db[Nout] = column sums of dY (optional; computed as dY^T 1 on the MFMA) -- A = the same
 *          dY planes, B = x planes, both NOT k-major, K = the batch rows (plane rows beyond them must be zero), C = dW fp32, no epilogue.
 * Same precision for both; K sums per 64-k tile in tile order, as slnlp_gemm_rows. */
int slnlp_gemm_rows_bwd(const slnlp_gemm_args* dgrad, const slnlp_gemm_args* wgrad, void* stream);
/* Output tile of slnlp_gemm_rows launches: -1 = automatic (16 x 16 for one fit's launch -- it is bound by what ONE compute unit can
 * load, so the panels are spread over as many as possible -- 64 x 16 or 64 x 32 for the merged launches of fits in lockstep, which
 * pay for total bytes instead); 0 / 1 / 2 force 16 x 16, 64 x 16, 64 x 32.  A tuning / test knob: the K sum is defined per 64-k
 * tile (partial products added in tile order), so results do not depend on it. */
int slnlp_set_rows_tile(int tile);
/* fp32 [R,K] rows (row stride ld) -> OCP e4m3 rows with one fp32 scale per row: scale[r] = max|x[r,:]| / 448 (1 for an
 * all-zero row), q[r,k] = e4m3(x[r,k] / scale[r]); row stride of q = ldq bytes.  The weight operand of precision 8. */
int slnlp_quant_rows_fp8(const float* x, int64_t ld, int R, int K, uint8_t* q, int64_t ldq, float* scale, void* stream);
/* fp32 [R,C] (row stride ld) -> bf16 hi/lo planes with row stride ldp (lo may be NULL); writes the valid
 * region only -- the planes' zero padding comes from their allocation. */
int slnlp_split_planes(const float* x, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, int64_t ldp, void* stream);

/* ------------------------------------------------------------- embedding --
 * x[s*B+b, :] = table[ids[b,s], :] * sqrt(E) + pe[s, :], then dropout.
 * model/transformer.py:106-109 forward_embedding; positional_encoding.py:48-49.
 * ids is batch-first int64 [B,S] with row stride ld_ids (for y: S=1). */
int slnlp_embed_fwd(const int64_t* ids, int64_t ld_ids, int B, int S, int E, int V,
                    const float* table, const float* pe /* NULL: no positional term */, float* out,
                    float scale /* sqrt(E) for the Transformer, 1 for the RNN models (bkp.py:49) */,
                    float drop_p, int drop_site, const unsigned long long* rng,
                    int64_t nan_idx /* id whose rows become NaN (decoder <pad> target), or -1 */, void* stream);
/* dtable[v,:] = sqrt(E) * sum_{tokens with id v} dropout_bwd(dx[token,:]);
 * rows with no token are zeroed.  Deterministic (fixed summation tree, no
 * float atomics).  scratch: slnlp_embed_bwd_scratch_bytes(B,S,E) bytes. */
int64_t slnlp_embed_bwd_scratch_bytes(int B, int S, int E);
int slnlp_embed_bwd(const int64_t* ids, int64_t ld_ids, int B, int S, int E, int V,
                    const float* dx, float* dtable, float scale,
                    int64_t zero_row /* nn.Embedding(padding_idx): this row gets no gradient; -1 = none */,
                    float drop_p, int drop_site, const unsigned long long* rng,
                    void* scratch, void* stream);

/* -------------------------------------------------------- self attention --
 * Encoder self-attention core for one layer, all (b,h) pairs: scores =
 * q k^T / sqrt(dh), blocked where key j > query i (causal, transformer.py:68 /
 * util.py:11-42) or ids[b,j] == pad (util.py:45-61), softmax, dropout, @ v.
 * qkv [S*B, 3E] is the in_proj output (q | k | v); ctx [S*B, E]; probs
 * [B,H,S,S] keeps the pre-dropout softmax for backward.  Requires S <= 64. */
int slnlp_attn_self_fwd(const float* qkv, const int64_t* ids, int64_t ld_ids, int64_t pad_idx,
                        int causal, int B, int S, int H, int dh,
                        float* ctx, float* probs,
                        float drop_p, int drop_site, const unsigned long long* rng, void* stream);
int slnlp_attn_self_bwd(const float* qkv, const float* probs, const float* dctx,
                        int B, int S, int H, int dh, float* dqkv,
                        float drop_p, int drop_site, const unsigned long long* rng, void* stream);

/* Decoder cross-attention with ONE query per sequence (tgt length 1) over S
 * memory positions, no masks (transformer.py:82-87 passes neither memory_mask
 * nor memory_key_padding_mask).  q [B,E]; kv rows m = s*B+b with row stride
 * ld_kv, k at column 0 and v at column E; probs [B,H,S]. */
/* Sequences longer than 64 (any S up to the reference's 5000-row positional table) take wave-per-row kernels instead of
 * the one-tile MFMA kernels -- same arguments and semantics; only the self-attention backward needs more: a scratch buffer
 * of slnlp_attn_long_scratch_bytes(B, S, H) bytes (the dS tensor between its row pass and its column pass). */
int64_t slnlp_attn_long_scratch_bytes(int B, int S, int H);
int slnlp_attn_self_bwd_long(const float* qkv, const float* probs, const float* dctx, int B, int S, int H, int dh,
                             float* dqkv, float* scratch, float drop_p, int drop_site, const unsigned long long* rng,
                             void* stream);
int slnlp_attn_cross_fwd(const float* q, const float* kv, int64_t ld_kv, int B, int S, int H, int dh,
                         float* ctx, float* probs,
                         float drop_p, int drop_site, const unsigned long long* rng, void* stream);
int slnlp_attn_cross_bwd(const float* q, const float* kv, int64_t ld_kv, const float* probs,
                         const float* dctx, int B, int S, int H, int dh,
                         float* dq, float* dkv, int64_t ld_dkv,
                         float drop_p, int drop_site, const unsigned long long* rng, void* stream);

/* -------------------------------------------------------------- layernorm --
 * y = (x - mean) * rstd * gamma + beta, eps added to the biased variance
 * (torch.nn.LayerNorm as used by nn.Transformer, eps 1e-5).  stats [rows,2] =
 * (mean, rstd) kept for backward.  The residual add is fused into the GEMM
 * that produced x. */
int slnlp_layernorm_fwd(const float* x, const float* gamma, const float* beta, int rows, int E,
                        float eps, float* y, float* stats, void* stream);
/* dx (+ optional add_to_dx [rows,E]) ; dx_drop (optional) = dropout_bwd(dx) at
 * drop_site for the sub-layer branch; partial [nblk,2,E] per-block partial
 * (dgamma, dbeta) sums, reduced later by slnlp_ln_param_reduce. nblk is
 * returned through *nblk_out (<= SLNLP_LN_MAX_PARTIALS).  Blocks are 16 rows; for rows >= 1024 the row kernel
 * sums them itself (one pass over dy and x), below a separate column-sum launch does. */
#define SLNLP_LN_MAX_PARTIALS 1024
int slnlp_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stats,
                        int rows, int E, const float* add_to_dx, float* dx, float* dx_drop,
                        float drop_p, int drop_site, const unsigned long long* rng,
                        float* partial, int* nblk_out, void* stream);
/* table: n entries of {partial ptr, nblk, dgamma ptr, dbeta ptr} in DEVICE memory */
typedef struct slnlp_ln_reduce_entry {
    const float* partial; float* dgamma; float* dbeta; int32_t nblk; int32_t E;
} slnlp_ln_reduce_entry;
int slnlp_ln_param_reduce(const slnlp_ln_reduce_entry* table_dev, int n, int max_E, void* stream);

/* ------------------------------------------------------------------ loss --
 * logp = log_softmax(logits) (transformer.py:88-89 / bkp.py:75-76) and
 * CrossEntropyLoss(ignore_index) ON THE LOG-PROBS as skorch applies it
 * (helper.py:61-70): loss = -mean_{y!=ignore} log_softmax(logp)[y].
 * Writes logp [B,V] (ld = V), loss[0], and (if dlogits) d loss / d logits. */
int slnlp_lsm_nll(const float* logits, int64_t ld_logits, const int64_t* y, int B, int V,
                  int64_t ignore_index, float* logp, float* loss, float* dlogits, int64_t ld_dlogits,
                  float* row_scratch /* [B] */, void* stream);
/* backward of log_softmax alone, for callers that own the criterion (torch
 * autograd): dlogits = dlogp - exp(logp) * rowsum(dlogp). */
int slnlp_lsm_bwd(const float* logp, const float* dlogp, int B, int V, float* dlogits,
                  int64_t ld_dlogits, void* stream);

/* -------------------------------------------------------------- optimizer --
 * clip_grad_norm_(max_norm, 2) + SGD(momentum, dampening 0, nesterov False,
 * no weight decay) over ONE flat parameter arena (helper.py:227-229,
 * config-transformer.yaml:19-20,40-43).  lr is read from device memory so a
 * captured graph follows ReduceLROnPlateau.  norm_out[0] receives the pre-clip
 * total norm; rng[1] (the dropout step counter) is incremented if rng != NULL. */
int slnlp_clip_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n,
                        const float* lr_dev, float momentum, float max_norm,
                        float* partials /* [1024] scratch */, float* norm_out,
                        unsigned long long* rng, void* stream);

/* clip_grad_norm_ + torch.optim.Adam (amsgrad False; torch/optim/adam.py _single_tensor_adam) over one flat arena:
 * step_count[0] (float, device) holds the number of steps taken so far and is advanced by one. */
int slnlp_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                         const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                         float* partials /* [1024] scratch */, float* norm_out, float* step_count, void* stream);

/* debug / test helper: materialise the keep mask (1.0 / 0.0) of a dropout site */
int slnlp_dropout_mask(float* out, int R, int C, float p, int site,
                       const unsigned long long* rng, void* stream);

/* ---------------------------------------------------------- LSTM / GRU cell --
 * Point-wise cell of torch.nn.LSTM (gates i,f,g,o) / torch.nn.GRU (r,z,n) as the
 * reference instantiates them (bkp.py:95-100,186-190), one timestep of up to two
 * directions per launch.  xproj = x W_ih^T + b_ih and hproj = h_{t-1} W_hh^T + b_hh
 * come from slnlp_gemm.  Sequence b advances only while t < lengths[b]
 * (pack_padded_sequence, bkp.py:110-114); otherwise the state is carried and the
 * layer output is `fill` (pad_packed_sequence(padding_value=pad_idx), :120-123).
 * lengths == NULL: every row is valid (decoder step). */
typedef struct slnlp_rnn_cell_dir {
    const float* xproj;      /* [B, G*Hd] of this timestep */
    const float* hproj;      /* [B, G*Hd] */
    float* h;                /* [B, Hd] running state, updated in place */
    float* c;                /* LSTM: [B, Hd] running cell state */
    float* hprev_save;       /* [B, Hd] h before the update (kept for backward) */
    float* cprev_save;       /* LSTM */
    float* acts;             /* [B, G*Hd] gate activations (kept for backward) */
    float* hn_save;          /* GRU: [B, Hd] hidden part of the n gate */
    float* out;              /* layer output rows of this timestep (row stride ld_out), or NULL */
    int32_t t, out_row0, out_col0; /* timestep; (row, col) origin of `out` inside its dropout site */
} slnlp_rnn_cell_dir;
int slnlp_rnn_cell_fwd(int lstm, const slnlp_rnn_cell_dir* dirs, int ndir, int B, int Hd,
                       const int64_t* lengths, float fill, int64_t ld_out,
                       float drop_p, int drop_site, const unsigned long long* rng, void* stream);
/* Fused forward timestep: the recurrent GEMM h_{t-1} W_hh^T (+ b_hh) and the cell above in ONE launch (results are
 * bit-identical to slnlp_gemm + slnlp_rnn_cell_fwd).  The new state is written to h_out, which must not alias h_in
 * (workgroups of the same launch still read h_{t-1}): callers chain the per-timestep `hprev` slots, so h_in doubles as
 * the saved h_{t-1} of this step. */
typedef struct slnlp_rnn_step_dir {
    const float* h_in;       /* [B, Hd] state before the step (= what backward needs as h_{t-1}) */
    float* h_out;            /* [B, Hd] state after the step */
    const float* w_hh;       /* [G*Hd, Hd] */
    const float* b_hh;       /* [G*Hd] or NULL */
    const float* xproj;      /* [B, G*Hd] of this timestep (x W_ih^T + b_ih) */
    float* c;                /* LSTM: [B, Hd] running cell state, updated in place */
    float* cprev_save;       /* LSTM */
    float* acts;             /* [B, G*Hd] */
    float* hn_save;          /* GRU */
    float* out;              /* layer output rows of this timestep (row stride ld_out), or NULL */
    int32_t t, out_row0, out_col0;
} slnlp_rnn_step_dir;
int slnlp_rnn_step_fwd(int lstm, const slnlp_rnn_step_dir* dirs, int ndir, int B, int Hd,
                       const int64_t* lengths, float fill, int64_t ld_out,
                       float drop_p, int drop_site, const unsigned long long* rng, int precision, void* stream);
/* Persistent forward of ALL S timesteps of one (bi)directional layer in ONE launch: each workgroup keeps its slice
 * of W_hh in LDS for the whole sequence and the Hd/16 x ndir co-resident workgroups meet at a device-wide barrier
 * between timesteps (results bit-identical to S calls of slnlp_rnn_step_fwd).  Covered shapes: B <= 64,
 * Hd % 64 == 0, G * Hd * 64 bytes + 16 KiB of LDS <= 156 KiB (LSTM: Hd <= 512); otherwise *launched = 0 and nothing
 * was done -- use the per-timestep entry point.  The per-timestep arrays are indexed by time t; `hprev` is the state
 * chain: slot t holds the state BEFORE time t is processed (slot of the first processed timestep: zeros, set by the
 * caller), the last state goes to h_final.  sync: 3 device words {barrier count, barrier generation, error flag},
 * zero before the first use; a non-zero error flag afterwards means a workgroup gave up waiting (bounded spin) and
 * the results are invalid.  Do not run more such launches concurrently than fit the GPU (one workgroup per CU). */
typedef struct slnlp_rnn_layer_dir {
    float* hprev;            /* [S][B, Hd] state chain (input slot zeroed by the caller; the rest is written) */
    float* h_final;          /* [B, Hd] */
    const float* w_hh;       /* [G*Hd, Hd] */
    const float* b_hh;       /* [G*Hd] or NULL */
    const float* xproj;      /* [S][B, G*Hd] */
    float* c;                /* LSTM: [B, Hd] running cell state (zeroed by the caller), updated in place */
    float* cprev;            /* LSTM: [S][B, Hd] */
    float* acts;             /* [S][B, G*Hd] */
    float* hn;               /* GRU: [S][B, Hd] */
    float* out;              /* layer output [S*B, ld_out] at column out_col0, or NULL */
    int32_t out_col0;
    int32_t reverse;         /* 0: t = 0..S-1, 1: t = S-1..0 */
} slnlp_rnn_layer_dir;
int slnlp_rnn_layer_fwd(int lstm, const slnlp_rnn_layer_dir* dirs, int ndir, int B, int Hd, int S,
                        const int64_t* lengths, float fill, int64_t ld_out,
                        float drop_p, int drop_site, const unsigned long long* rng, int precision,
                        uint32_t* sync, int* launched, void* stream);
/* Backward of one timestep: consumes the running d(state) and d(out), emits the gate
 * gradients dgx (w.r.t. xproj) / dgh (w.r.t. hproj; LSTM: same buffer as dgx) and
 * `carry`, the part of dh that bypasses the recurrent matmul; the caller forms
 * dh_state(t-1) = dgh W_hh + carry with slnlp_gemm. */
typedef struct slnlp_rnn_cell_bwd_dir {
    float* dh_state; float* dc_state;
    const float* dout;       /* d(layer output) rows of this timestep (row stride ld_dout), or NULL */
    const float* acts; const float* cprev_save; const float* hprev_save; const float* hn_save;
    float* dgx; float* dgh; float* carry;
    int32_t t, out_row0, out_col0;
    /* optional: dh_state arrives as a sum of partial products (the caller split the K loop of dgh W_hh over
     * several GEMM jobs of one launch): dh = dh_state + sum_{e < n_extra} dh_extra[e * extra_stride + i] */
    const float* dh_extra; int64_t extra_stride; int32_t n_extra;
} slnlp_rnn_cell_bwd_dir;
int slnlp_rnn_cell_bwd(int lstm, const slnlp_rnn_cell_bwd_dir* dirs, int ndir, int B, int Hd,
                       const int64_t* lengths, int64_t ld_dout,
                       float drop_p, int drop_site, const unsigned long long* rng, void* stream);

/* One backward timestep in ONE launch: the recurrent data gradient of the step processed just before and this step's cell
 * backward (what autograd does for nn.LSTM / nn.GRU behind /root/reference/model/base/encoder_decoder_attn_bkp.py:95-132):
 *   dh = dgh_next W_hh + carry (+ dout)  ->  slnlp_rnn_cell_bwd's arithmetic  ->  dgx, dgh, dc_state, carry of this step.
 * dgh_next == NULL (both directions): the first step of a layer, dh = cell.dh_state.  cell.dh_extra / n_extra are ignored; the
 * product's partial sums are added in gate order, exactly as the K-sliced slnlp_gemm_group + slnlp_rnn_cell_bwd pair does.
 * Covered: Hd % 64 == 0 (any B); dgh_next / w_hh 16-byte aligned. */
typedef struct slnlp_rnn_step_bwd_dir {
    slnlp_rnn_cell_bwd_dir cell;
    const float* dgh_next;   /* [B, G*Hd] dgh written by the previous launch of this chain, or NULL */
    const float* w_hh;       /* [G*Hd, Hd] */
} slnlp_rnn_step_bwd_dir;
int slnlp_rnn_step_bwd(int lstm, const slnlp_rnn_step_bwd_dir* dirs, int ndir, int B, int Hd,
                       const int64_t* lengths, int64_t ld_dout,
                       float drop_p, int drop_site, const unsigned long long* rng, int precision, void* stream);

/* Bahdanau (MLP) attention, one query per sequence (bkp.py:304-327 with max_len 1):
 * scores[s] = w_e . tanh(q[b] + proj_key[s,b]); masked where ids[b,s] == pad; softmax;
 * ctx = alphas . value.  proj_key [S*B,Hd] / value [S*B,2Hd] rows are time-major. */
int slnlp_bahdanau_fwd(const float* q, const float* proj_key, const float* value, const float* w_energy,
                       const int64_t* ids, int64_t ld_ids, int64_t pad_idx, int B, int S, int Hd,
                       float* alphas, float* ctx, void* stream);
int slnlp_bahdanau_bwd(const float* q, const float* proj_key, const float* value, const float* w_energy,
                       const float* alphas, const float* dctx, int B, int S, int Hd,
                       float* dq, float* dproj_key, float* dvalue, float* dwe_partial /* [B,Hd] scratch */,
                       float* dw_energy, void* stream);

/* ------------------------------------------------------ Transformer plan --
 * Whole-model drop-in for model.Transformer (model/transformer.py:10-109):
 * the library owns the parameter-arena LAYOUT (names follow the reference
 * state_dict), the activation workspace layout and the launch sequence. */
typedef struct slnlp_tf_config {
    int32_t E, H, N, F;          /* embedding_size, num_heads, num_layers, hidden_size */
    int32_t Vs, Vt;              /* len(src_vocab), len(tgt_vocab) */
    int32_t B, S;                /* max batch, sequence length (S <= 64) */
    int32_t pad_src, pad_tgt;    /* vocab.stoi['<pad>'] (util.py:5-6) */
    float dropout;
    int32_t precision;           /* 1 | 3 */
} slnlp_tf_config;

int slnlp_tf_num_params(const slnlp_tf_config* cfg);
/* i-th parameter in reference state_dict order: name (<=127 chars), shape, offset (floats) */
int slnlp_tf_param_info(const slnlp_tf_config* cfg, int i, char* name, int64_t shape[2], int* ndim,
                        int64_t* offset);
int64_t slnlp_tf_arena_floats(const slnlp_tf_config* cfg);
int64_t slnlp_tf_workspace_bytes(const slnlp_tf_config* cfg);

typedef struct slnlp_tf_buffers {
    float* params;               /* arena, slnlp_tf_arena_floats */
    float* grads;                /* arena-shaped */
    float* momentum;             /* arena-shaped */
    const float* pe;             /* [>=S, E] positional table (positional_encoding.py:27-35) */
    void* workspace;             /* slnlp_tf_workspace_bytes */
    unsigned long long* rng;     /* [2] = {seed, step} */
    float* lr;                   /* [1] */
    float* scalars;              /* [4] = {loss, grad_norm, -, -} */
} slnlp_tf_buffers;

typedef struct slnlp_tf_plan slnlp_tf_plan;
int slnlp_tf_create(const slnlp_tf_config* cfg, const slnlp_tf_buffers* buf, slnlp_tf_plan** out);
void slnlp_tf_destroy(slnlp_tf_plan* plan);
/* forward: X int64 [B,S] (row stride S), y int64 [B] -> logp [B,Vt] (may be
 * NULL).  Always evaluates the criterion too (scalars[0] = loss); train != 0
 * applies dropout, keeps activations and seeds backward with d loss/d logits. */
int slnlp_tf_forward(slnlp_tf_plan* plan, const int64_t* X, const int64_t* y, int B, int train,
                     float* logp, void* stream);
/* re-seed backward from an external d loss / d logp (torch autograd owns the criterion) */
int slnlp_tf_seed_dlogp(slnlp_tf_plan* plan, const float* dlogp, void* stream);
/* backward through the whole model into buf.grads (every element written) */
int slnlp_tf_backward(slnlp_tf_plan* plan, void* stream);
/* clip + SGD-momentum on the arena; scalars[1] = pre-clip grad norm */
int slnlp_tf_optim(slnlp_tf_plan* plan, float momentum, float max_norm, void* stream);
/* clip + torch.optim.Adam (amsgrad False) on the arena -- north_star's "fused SGD-momentum/Adam update"; the reference
 * reaches any torch.optim class through pydoc.locate (helper.py:91-104).  exp_avg = buf.momentum, exp_avg_sq = an
 * arena-shaped buffer of the caller's (zero before the first step), step count = scalars[2] (device-side). */
int slnlp_tf_optim_adam(slnlp_tf_plan* plan, float* exp_avg_sq, float beta1, float beta2, float eps, float weight_decay,
                        float max_norm, void* stream);
/* forward(train) + loss + backward + optim in one call */
int slnlp_tf_train_step(slnlp_tf_plan* plan, const int64_t* X, const int64_t* y, int B,
                        float momentum, float max_norm, float* logp, void* stream);
/* Capture one train step over FIXED device buffers (X, y, logp) for batch size
 * B into a hipGraph kept inside the plan (one per distinct B);
 * slnlp_tf_graph_launch(B) replays it.  lr, the dropout step counter and the
 * data are read from device memory, so one captured graph serves every step
 * of a fit.  `stream` must not be the null stream. */
int slnlp_tf_graph_capture_train(slnlp_tf_plan* plan, const int64_t* X, const int64_t* y, int B,
                                 float momentum, float max_norm, float* logp, void* stream);
int slnlp_tf_graph_launch(slnlp_tf_plan* plan, int B, void* stream);
/* test helper: copy a named activation tap ("enc0", "memory", "dec1", "logits", ...) */
int slnlp_tf_tap(slnlp_tf_plan* plan, const char* name, float* out, int64_t max_floats,
                 int64_t* n_out, void* stream);
/* The parameter arena was written from outside the library (load_state_dict, a torch optimizer, an in-place edit):
 * data derived from it (the bf16 weight planes the fused update keeps current) is rebuilt by the next forward. */
int slnlp_tf_params_changed(slnlp_tf_plan* plan);
/* test / debug helper: "name byte_offset" lines of the workspace's activation and gradient buffers, in layout order */
int slnlp_tf_debug_layout(const slnlp_tf_config* cfg, char* out, int64_t out_bytes);

/* slnlp_*_destroy wait for the device (hipDeviceSynchronize) before they return: the plan's buffers are the caller's and
 * may be freed next.  on = 0 drops that wait FOR THIS PLAN (never process-wide) -- only for a caller whose buffers come from
 * a stream-ordered allocator on the stream the plan ran on (torch's caching allocator), where the device-wide wait stalls
 * every other host thread's queued work each time a fit ends. */
int slnlp_tf_set_destroy_sync(slnlp_tf_plan* plan, int on);

/* One kernel sequence per device (default).  The step entry points (slnlp_{tf,rnn}_{forward,backward,optim*,train_step,
 * graph_launch}, slnlp_*_lockstep_{step,epoch}) serialise per device: host threads enqueue whole steps in turn, and a step issued
 * on another stream than the device's previous step waits (event) for that stream's tail, so the library's kernels never overlap
 * across streams inside one process.  Why: MI355X / ROCm 7.2 compute packed fp32 VALU instructions (v_pk_{add,mul,fma}_f32 with
 * op_sel) wrongly in lanes 48-63 while another kernel's waves run MFMA on the same CU (reproducer
 * tools/probes/packed_fp32_repro.hip).  The shipped library is built WITHOUT those instructions (and tests/ disassemble it), so
 * its kernels are bit-stable on overlapping queues; the ordering is kept as the default because it costs nothing with one
 * stream and protects a caller of a library rebuilt with other flags.
 *   slnlp_set_thread_stream_policy(0): the CALLING host thread's steps skip the ordering (1: take part; -1: follow the
 *     process-wide policy again).  For a thread that owns a stream and wants its steps to overlap other threads' -- the grid
 *     search's worker threads (slnlp/net.py) -- never process-wide.
 *   slnlp_set_stream_policy(0): the process-wide switch (probes). */
int slnlp_set_thread_stream_policy(int serialise);
int slnlp_set_stream_policy(int serialise);

/* Split-bf16 passes of the gradient products that run on the plane GEMM (the [S*B]-row dgrad / wgrad of every encoder-side
 * Linear; what autograd computes for nn.Linear behind /root/reference/model/transformer.py:40-45,82-87): the DEFAULT FOR PLANS
 * CREATED AFTERWARDS -- a plan copies both counts at creation and keeps them for its life, so its eager steps, its captured
 * graph, its lockstep program and every host thread that steps it issue the same products; other plans are not touched.  3: the full split, A_lo B_hi + A_hi B_lo + A_hi B_hi (fp32-grade
 * products).  2: dY enters with its bf16 head only (rounded to nearest: unbiased), A_hi (B_hi + B_lo) -- a third less MFMA work,
 * a quarter less operand staging, a three-stage ring in the same LDS.  Default since round 4: wgrad 2, dgrad 2.  Measured against
 * the reference's golden training trajectories (tools/backward_pass_errors.py, profiles/r04_backward_pass_errors.jsonl; cfg2, five
 * steps): worst per-tensor gradient-norm error 2.3e-3 at (2, 2) against 2.6e-3 at (3, 3), loss 6e-5 against 3e-5 (bar 1e-3),
 * pre-clip gradient norm 4.3e-3 against 4.2e-3 (bar 5e-3), weights after five steps 2e-7 either way -- the gradient's distance to
 * the fp32 reference is set by ReLU gates that sit within rounding of zero, not by the 2^-9 rounding of dY.  The forward products
 * (logits, loss: the 1e-3 / bit-exact-argmax bar) always take 3 passes at precision 3.  Env: SLNLP_WGRAD_PASSES / SLNLP_DGRAD_PASSES. */
int slnlp_set_backward_passes(int wgrad, int dgrad);
int slnlp_get_backward_passes(int* wgrad, int* dgrad);

/* Measurement hook (bench.py's roofline): between _start and _stop every plane-GEMM group launch that goes out as a plain launch
 * (not under graph capture) is bracketed by two HIP events on ITS stream; _stop waits for them and returns up to max_out records --
 * the launch's workgroup count, job count, tile geometry (0: 64 x 64, 1: 128 x 128 / 64-k, 2: 128 x 128 / 32-k) and the time
 * between the events in microseconds: the dominant kernel timed inside the train step it belongs to.  Process-wide; not for
 * production steps (two event records per launch). */
typedef struct slnlp_timed_launch { int32_t blocks, njobs, geometry; float us; } slnlp_timed_launch;
int slnlp_launch_timer_start(int max_records);
int slnlp_launch_timer_stop(slnlp_timed_launch* out, int max_out);

/* ---------------------------------------------------------------- lockstep --
 * K Transformer fits of ONE shape (own weights, lr, dropout rate, seed and data) advancing through one launch
 * sequence: every call site of the step is launched once for all K fits, so a 50-row decoder stage becomes a
 * K x 50-row stage at the same latency.  Replaces the reference's one-fit-at-a-time dask tasks
 * (/root/reference/main.py:70-78, helper.py:490-526) for the fits of a work unit; each fit's results are
 * bit-identical to running it alone through slnlp_tf_train_step / slnlp_tf_forward.
 * The plans must outlive the group and must not be stepped on their own while it exists.
 * workspace: slnlp_tf_lockstep_workspace_bytes(cfg, K) bytes, 256-byte aligned, caller-owned (staging + tables). */
typedef struct slnlp_tf_lockstep slnlp_tf_lockstep;
int64_t slnlp_tf_lockstep_workspace_bytes(const slnlp_tf_config* cfg, int K);
int slnlp_tf_lockstep_create(slnlp_tf_plan** plans, int K, void* workspace, int64_t workspace_bytes, void* stream,
                             slnlp_tf_lockstep** out);
void slnlp_tf_lockstep_destroy(slnlp_tf_lockstep* group);
/* data slot (0..3, e.g. train / valid / test) of every fit: X[f] int64 [rows, S], y[f] int64 [rows] on the device,
 * and where a pass over it leaves its results: logp[f] float [rows, Vt], loss[f] float [ceil(rows / batch)] */
int slnlp_tf_lockstep_set_data(slnlp_tf_lockstep* group, int slot, const int64_t* const* X, const int64_t* const* y,
                               int64_t rows, float* const* logp, float* const* loss, void* stream);
/* rows [row0, row0 + B) of the slot, every fit: train != 0 -> forward + criterion + backward + clip + SGD-momentum,
 * else eval-mode forward + criterion.  Log-probs go to logp[f][row0 ..], the batch loss to loss[f][step_index]. */
int slnlp_tf_lockstep_step(slnlp_tf_lockstep* group, int slot, int64_t row0, int B, int step_index, int train,
                           float momentum, float max_norm, void* stream);
/* one pass over the slot in dataset order, batches of `batch` rows (the last may be shorter): no host sync inside */
int slnlp_tf_lockstep_epoch(slnlp_tf_lockstep* group, int slot, int batch, int train, float momentum, float max_norm,
                            void* stream);
/* kernel launches per step of the cached program for (slot, B, train), or -1 if none has been recorded yet */
int slnlp_tf_lockstep_num_launches(slnlp_tf_lockstep* group, int slot, int B, int train);
/* train with clip_grad_norm_ + Adam instead of SGD-momentum from now on (what slnlp_tf_optim_adam does for one fit):
 * exp_avg = each plan's momentum arena, exp_avg_sq[f] = an arena-shaped buffer of the caller's per fit (zero before the first
 * step), step count = each plan's scalars[2]; `momentum` of the step calls is then ignored.  Drops recorded train programs. */
int slnlp_tf_lockstep_set_adam(slnlp_tf_lockstep* group, float* const* exp_avg_sq, float beta1, float beta2, float eps,
                               float weight_decay);
int slnlp_tf_lockstep_set_destroy_sync(slnlp_tf_lockstep* group, int on);   /* as slnlp_tf_set_destroy_sync, for the group's tables */


/* ------------------------------------------------- enc-dec RNN (+attn) plan --
 * Whole-model drop-in for model.EncoderDecoder{LSTM,GRU}Attn
 * (model/base/encoder_decoder_attn_bkp.py:330-413): packed bidirectional encoder
 * (:102-132), bridge (:268-280), ONE Bahdanau-attention decoder step fed <bos>
 * (:202-266 with MAX_OUTPUT_LEN = 1, :332), generator on the decoder state (:40-46,69-76).
 * Same conventions as the Transformer plan; buffers are a slnlp_tf_buffers (pe unused). */
typedef struct slnlp_rnn_config {
    int32_t lstm;                /* 1 = LSTM, 0 = GRU */
    int32_t E, Hd, N;            /* embedding_size, hidden_size, num_layers */
    int32_t Vs, Vt, B, S;
    int32_t pad_src, pad_tgt, bos_idx;   /* bos_idx = tgt_vocab.stoi['<bos>'] (0 on a torchtext-0.6 vocab) */
    float dropout;
    int32_t precision;
} slnlp_rnn_config;
int slnlp_rnn_num_params(const slnlp_rnn_config* cfg);
int slnlp_rnn_param_info(const slnlp_rnn_config* cfg, int i, char* name, int64_t shape[2], int* ndim, int64_t* offset);
int64_t slnlp_rnn_arena_floats(const slnlp_rnn_config* cfg);
int64_t slnlp_rnn_workspace_bytes(const slnlp_rnn_config* cfg);
typedef struct slnlp_rnn_plan slnlp_rnn_plan;
int slnlp_rnn_create(const slnlp_rnn_config* cfg, const slnlp_tf_buffers* buf, slnlp_rnn_plan** out);
void slnlp_rnn_destroy(slnlp_rnn_plan* plan);
/* X int64 [B,S], y int64 [B] (only the criterion reads it: the decoder consumes <bos>), lengths int64 [B] */
int slnlp_rnn_forward(slnlp_rnn_plan* plan, const int64_t* X, const int64_t* y, const int64_t* lengths, int B,
                      int train, float* logp, void* stream);
/* on = 1: run each encoder layer's S timesteps as ONE persistent launch (slnlp_rnn_layer_fwd) instead of one launch
 * per timestep.  Default 0: measured no faster in round 1, and its Hd/16 x 2 workgroups must all be resident at
 * once, so never enable it when several fits share the GPU. */
int slnlp_rnn_set_persistent(slnlp_rnn_plan* plan, int on);
/* on = 1: the encoder's backward through time issues ONE launch per timestep (slnlp_rnn_step_bwd: recurrent data gradient +
 * cell backward; Hd % 64 == 0); on = 0 (default): the cell kernel + K-sliced grouped GEMM pair.  The fused launch halves the
 * launches of a backward pass but measured slower for one fit (cfg3 LSTM 9.11 vs 8.09 ms) and faster only
 * for many GRU fits in lockstep, so it is opt-in (env SLNLP_RNN_FUSED_BWD=1 sets it for new plans).  Same results to fp32
 * rounding.  Takes effect for launches issued (or recorded by a lockstep group) after the call. */
int slnlp_rnn_set_fused_backward(slnlp_rnn_plan* plan, int on);
/* *status = 0 when every device-wide barrier of the plan's persistent kernels completed, 1 if a workgroup timed out
 * (bounded spin; that step's results are invalid).  Synchronises the device. */
int slnlp_rnn_health(slnlp_rnn_plan* plan, int* status);
int slnlp_rnn_seed_dlogp(slnlp_rnn_plan* plan, const float* dlogp, void* stream);
int slnlp_rnn_backward(slnlp_rnn_plan* plan, void* stream);
int slnlp_rnn_optim(slnlp_rnn_plan* plan, float momentum, float max_norm, void* stream);
/* clip + torch.optim.Adam on the arena, as slnlp_tf_optim_adam (exp_avg = buf.momentum, step count = scalars[2]) */
int slnlp_rnn_optim_adam(slnlp_rnn_plan* plan, float* exp_avg_sq, float beta1, float beta2, float eps, float weight_decay,
                         float max_norm, void* stream);
int slnlp_rnn_set_destroy_sync(slnlp_rnn_plan* plan, int on);   /* as slnlp_tf_set_destroy_sync */
int slnlp_rnn_train_step(slnlp_rnn_plan* plan, const int64_t* X, const int64_t* y, const int64_t* lengths, int B,
                         float momentum, float max_norm, float* logp, void* stream);
int slnlp_rnn_graph_capture_train(slnlp_rnn_plan* plan, const int64_t* X, const int64_t* y, const int64_t* lengths,
                                  int B, float momentum, float max_norm, float* logp, void* stream);
int slnlp_rnn_graph_launch(slnlp_rnn_plan* plan, int B, void* stream);
/* taps: "enc_out" [S*B,2Hd] (time-major), "enc_final" [N*B,2Hd], "alphas" [B,S], "context" [B,2Hd], "dec_out" [B,Hd], "logits" */
int slnlp_rnn_tap(slnlp_rnn_plan* plan, const char* name, float* out, int64_t max_floats, int64_t* n_out, void* stream);

/* Lockstep for K EncoderDecoder{LSTM,GRU}Attn fits of one shape -- the same contract as slnlp_tf_lockstep_* above
 * (one launch per call site for all K fits, each fit bit-identical to its solo run; replaces the one-fit-at-a-time
 * tasks of /root/reference/main.py:70-78).  The recurrence makes an RNN fit a chain of ~600 dependent small launches
 * per step, so K fits cost little more than one.  A data slot also carries lengths[f] int64 [rows]
 * (pack_padded_sequence semantics, bkp.py:110-114).  The persistent layer kernel must be off (it is by default). */
typedef struct slnlp_rnn_lockstep slnlp_rnn_lockstep;
int64_t slnlp_rnn_lockstep_workspace_bytes(const slnlp_rnn_config* cfg, int K);
int slnlp_rnn_lockstep_create(slnlp_rnn_plan** plans, int K, void* workspace, int64_t workspace_bytes, void* stream,
                              slnlp_rnn_lockstep** out);
void slnlp_rnn_lockstep_destroy(slnlp_rnn_lockstep* group);
int slnlp_rnn_lockstep_set_data(slnlp_rnn_lockstep* group, int slot, const int64_t* const* X, const int64_t* const* y,
                                const int64_t* const* lengths, int64_t rows, float* const* logp, float* const* loss,
                                void* stream);
int slnlp_rnn_lockstep_step(slnlp_rnn_lockstep* group, int slot, int64_t row0, int B, int step_index, int train,
                            float momentum, float max_norm, void* stream);
int slnlp_rnn_lockstep_epoch(slnlp_rnn_lockstep* group, int slot, int batch, int train, float momentum, float max_norm,
                             void* stream);
int slnlp_rnn_lockstep_num_launches(slnlp_rnn_lockstep* group, int slot, int B, int train);
int slnlp_rnn_lockstep_set_adam(slnlp_rnn_lockstep* group, float* const* exp_avg_sq, float beta1, float beta2, float eps,
                                float weight_decay);
int slnlp_rnn_lockstep_set_destroy_sync(slnlp_rnn_lockstep* group, int on);

#ifdef __cplusplus
}
#endif
#endif /* SLNLP_H */
