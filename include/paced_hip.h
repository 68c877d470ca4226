/*
 * paced_hip.h -- C ABI of libpaced_hip.so: the gfx950 (MI355X) kernels of the PacedNegatives
 * lceT5 training step and of the paced negative-pool selection.
 *
 * The reference (Parry-Parry/PacedNegatives) is pure Python and has no FFI of its own; every
 * arithmetic step of its hot path is a call into torch / Hugging Face transformers.  Each entry
 * point below therefore cites the reference call site (file:line under the reference checkout)
 * or the third-party routine that call site reaches (HF:line = transformers 5.15.0
 * models/t5/modeling_t5.py) whose device work it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed or owned by torch's allocator) unless it is
 *     named host_*; the library never allocates, frees or synchronises;
 *   - bf16 tensors are passed as void* (16-bit brain float, row-major, row stride in elements);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - return value: 0 on success, negative on error; paced_last_error() gives the message of the
 *     calling thread's most recent failure;
 *   - dropout: p in [0,1); a kept element is scaled by 1/(1-p); the decision is a pure function
 *     of (seed, element index), so forward and backward calls that pass the same (p, seed) agree
 *     without a stored mask.  p = 0 disables it.
 */
#ifndef PACED_HIP_H
#define PACED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int paced_abi_version(void);
const char* paced_last_error(void);

/* ---- dense contraction ------------------------------------------------------------------
 * C = alpha * op(A) * op(B) with a fused epilogue; fp32 accumulation on the MFMA units.
 *   trans_a = 0: A is [M][K] (lda >= K)       trans_a = 1: A is [K][M] (lda >= M)
 *   trans_b = 0: B is [N][K] (ldb >= K)       trans_b = 1: B is [K][N] (ldb >= N)
 * (0,0) = nn.Linear forward  y = x W^T   -- HF:304,325,326,367 (q,k,v,o), HF:84-93 (wi,wo), HF:1047 (lm_head)
 * (0,1) = input gradient     dx = dy W    -- autograd of the above, reached from lceT5.py:155
 * (1,1) = weight gradient    dW = dy^T x  -- idem; contraction over tokens, `splits` workgroups per
 *                                            output tile, fp32 atomic accumulation into C
 * epilogue:
 *   0  C(bf16) = acc
 *   1  C(bf16) = aux + dropout(acc)                  residual add, HF:400-401, HF:140
 *   2  C(bf16) = dropout(relu(acc))                  HF:85-86; when aux is not NULL it is an output here: a byte
 *                                                    matrix, ldaux BYTES per row, bit (n%8) of byte n/8 of row m = (C[m][n] > 0)
 *   3  C(bf16) = aux > 0 ? acc * mask_scale : 0      backward of (2) given its saved output
 *   4  C(f32)  = acc
 *   5  C(f32) += acc   (atomic; required when splits > 1)
 *   6  C(bf16) = bit(aux, m, n) ? acc * mask_scale : 0   backward of (2) given the byte matrix it wrote (N % 8 == 0,
 *                                                    ldaux >= N/8 bytes); reads 1/16 of what (3) reads
 * Requirements: K % 64 == 0 for a non-transposed operand; N % 4 == 0 (N % 8 == 0 for a transposed B); lda, ldb % 8 == 0;
 * 16-byte aligned bases.  M, N and a transposed contraction length may be ragged; a transposed A with M % 8 != 0
 * needs lda >= M rounded up to 8 with every row of lda elements allocated (the padding is read, never used). */
int paced_gemm_bf16(const void* A, const void* B, void* C, const void* aux,
                    int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                    int trans_a, int trans_b, int epilogue, float alpha, float mask_scale,
                    float drop_p, uint32_t drop_seed, int splits, void* stream);

/* paced_gemm_bf16 with HEAD-MAJOR operands.  A head-major operand (its stride argument > 0, its leading dimension 64)
 * stores the feature columns [64h, 64h + 64) as the contiguous [rows, 64] matrix at base + h * stride (elements) -- the
 * layout HF's T5Attention gives its projected states before the attention product (modeling_t5.py `shape()`:
 * view(batch, -1, n_heads, d_kv).transpose(1, 2)) -- so that one head of one sequence is one contiguous block for the
 * attention kernels below instead of 128-byte pieces at a row stride.  A: features = K (trans_a = 0) or M (trans_a = 1);
 * B: transposed only, features = N; C: plain bf16 epilogue only, features = N.  Feature counts are multiples of 64, the
 * stride is at least rows * 64.  Of a transposed A and B at most one may be head-major; when its row count (the
 * contraction length) is not a multiple of 64, the last step reads up to 63 rows past it -- the padding or the next
 * head's rows, zeros past the last head -- which must hold finite values: they meet the other operand's zeros.
 * Same bits as paced_gemm_bf16 on the same values. */
int paced_gemm_bf16_heads(const void* A, const void* B, void* C, const void* aux,
                          int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                          int trans_a, int trans_b, int epilogue, float alpha, float mask_scale,
                          float drop_p, uint32_t drop_seed, int splits,
                          long long a_heads, long long b_heads, long long c_heads, void* stream);

/* `batch` independent products of the same shape in one launch: member b uses A + b*strideA, B + b*strideB,
 * C + b*strideC (strides in elements, multiples of 8).  Epilogues 0 (bf16), 4 (fp32) and 5 (fp32 atomic) only.
 * Serves the per-head products of the decoder's cross-attention (HF modeling_t5.py:459-498 computes them as one
 * projection followed by a per-head view; here each head's 64 x d_model weight slice is applied on its own). */
int paced_gemm_bf16_batched(const void* A, const void* B, void* C,
                            int M, int N, int K, int lda, int ldb, int ldc,
                            int trans_a, int trans_b, int epilogue, float alpha, int splits,
                            int batch, long long strideA, long long strideB, long long strideC, void* stream);

/* One entry point for every form above (plain: head strides 0, batch 1; head-major: a/b/c_heads; batched: batch and
 * strides) with PER-CALL launch options.  The same products as paced_gemm_bf16 (same reference call sites); what the
 * options choose never changes a result bit except through the kernel that is taken (tile edge, ring variant).
 *   grid_limit  < 0: the process default;  0: one workgroup per compute unit;  n > 0: at most n workgroups for a 256x256
 *               phased launch.  A phased workgroup keeps its compute unit for the whole launch, so the units a smaller
 *               grid leaves out are where kernels of other streams can start meanwhile: LCEModel.training_step passes a
 *               limit for the no-grad forwards (lceT5.py:131-133) that run beside the two-token decoder of the pass that
 *               keeps state.
 *   tile        < 0: the process default;  0: heuristic;  64 / 128 / 256: force that tile configuration.
 *   ring        < 0: the process default;  0 / 1: ring variant of the small-tile kernels off / on.
 *   kernel      OUT: the kernel this call launched, as paced_gemm_last_kernel() encodes it (0 if the call failed).
 * opts may be NULL (all process defaults).  The options are read on the calling thread for this call only: nothing another
 * thread launches meanwhile sees them. */
typedef struct paced_gemm_opts {
    int grid_limit;
    int tile;
    int ring;
    int kernel;
} paced_gemm_opts;
int paced_gemm_bf16_ex(const void* A, const void* B, void* C, const void* aux,
                       int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                       int trans_a, int trans_b, int epilogue, float alpha, float mask_scale,
                       float drop_p, uint32_t drop_seed, int splits,
                       long long a_heads, long long b_heads, long long c_heads,
                       int batch, long long strideA, long long strideB, long long strideC,
                       paced_gemm_opts* opts, void* stream);

/* ---- process-wide defaults of the options above: for benchmarks and tests only (the training step passes its options per
 * call).  Tile selection: 0 = heuristic (default), 64 = 64x64 tiles, 128 = 128x128 tiles, 256 = 256x256 phased tiles (one
 * workgroup per CU, LDS ring). */
int paced_gemm_set_tile(int tile);

/* Ring variant of the 64x64 / 128x128 kernels (a workgroup keeps several K steps in flight in its own LDS; taken when
 * the output has at most a few tiles per CU, e.g. the decoder's two-token rows): 1 = on (default; the environment
 * variable PACED_GEMM_RING=0 turns it off at load), 0 = off.  Same results bit for bit for the non-atomic epilogues.
 * Process-wide; meant for benchmarking and tests. */
int paced_gemm_set_ring(int on);

/* Process default of the upper limit on the workgroups of a 256x256 phased launch: 0 = one per compute unit (default;
 * PACED_GEMM_GRID=n sets a limit at load).  Read at launch time by calls that pass no limit of their own.
 * paced_gemm_get_grid returns the current default. */
int paced_gemm_set_grid(int workgroups);
int paced_gemm_get_grid(void);

/* Size in bytes of a TILE-ORDERED bit matrix for an M x N product, or 0 when a product of that shape would not take the
 * 256x256 phased kernel (then only the row-major form, ldaux >= N/8, is accepted).  The sign-bit matrix that epilogue 2
 * writes and epilogue 6 reads (the ReLU mask between T5DenseActDense's wi and wo, HF modeling_t5.py:288-300, kept for
 * the backward pass reached from lceT5.py:155) is only ever passed from one launch to the other, so the two may agree on
 * any order: with ldaux = -1 the aux buffer (16-byte aligned, at least this many bytes) holds 16 bytes per thread and
 * 256x256 tile, in the order the kernel's threads meet the elements.  The writer then stores one 16-byte value per
 * thread and tile instead of 16 single bytes, and the reader asks for its 16 bytes before the tile's contraction loop,
 * so that the epilogue does not wait on memory.  Writer and reader must have the same M and N. */
long long paced_gemm_tile_bits_bytes(int M, int N);

/* Which kernel the calling THREAD's most recent paced_gemm_bf16[_batched|_heads|_ex] launch used: the tile edge (64, 128 or 256;
 * 256 is the phased kernel), plus 1 when the ring variant took it; 0 before the first launch.  For profilers that
 * attribute launch times to kernels (bench.py's roofline leg). */
int paced_gemm_last_kernel(void);

/* ---- T5LayerNorm (RMS norm), HF:59-72; backward reached from lceT5.py:155 ------------------
 * fwd: y = bf16(x * rsqrt(mean(x^2)+eps) * w * out_scale) then dropout; rstd[M] saved (may be NULL).
 *      out_scale = d_model^-0.5 folds the tied-head rescale HF:1044-1045 into the final decoder norm.
 * bwd: dx = norm_backward(dropout_mask(dy)) + dres (dres may be NULL); dw[d] += ... (fp32 atomics,
 *      may be NULL); dx2 (may be NULL) = dropout2_mask(dx), the masked copy the preceding
 *      sub-layer's residual branch consumes. */
int paced_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int d, float eps,
                      float out_scale, float drop_p, uint32_t drop_seed, void* stream);
int paced_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres,
                      void* dx, void* dx2, float* dw, int M, int d, float out_scale,
                      float drop_p, uint32_t drop_seed, float drop2_p, uint32_t drop2_seed, void* stream);

/* ---- token embedding, HF:678,725 (gather + dropout) and its scatter-add gradient ------------ */
int paced_embedding_fwd(const long long* ids, const void* table, void* out, int M, int d, int vocab,
                        float drop_p, uint32_t drop_seed, void* stream);
int paced_embedding_bwd(const long long* ids, const void* dout, float* dtable, int M, int d, int vocab,
                        float drop_p, uint32_t drop_seed, void* stream);

/* ---- dropout helpers (nn.Dropout, HF:86,140,400,725,745) --------------------------------- */
int paced_dropout_apply(const void* x, void* y, size_t n, float drop_p, uint32_t drop_seed, void* stream);
int paced_dropout_mask(float* out, size_t n, float drop_p, uint32_t drop_seed, void* stream);   /* factors, for tests */
int paced_cast_f32_bf16(const float* x, void* y, size_t n, void* stream);

/* ---- relative position bias, HF:264-279 ------------------------------------------------------
 * tbl[h][i] = rel_bias[bucket[i]][h], i = key - query + (L-1)/2 for the encoder (L = 2S-1); the
 * bucket of every offset (HF:217-262) is computed on the host.  bwd sums `replicas` copies of dtbl
 * ([replicas, H, L], see paced_enc_attn_bwd) and scatters the sum back (atomics). */
int paced_relbias_table_fwd(const void* rel_bias, const int* bucket, float* tbl, int H, int L, void* stream);
int paced_relbias_table_bwd(const float* dtbl, int replicas, const int* bucket, float* drel_bias, int H, int L, void* stream);

/* ---- encoder self-attention, HF:144-173 with scaling = 1.0 (HF:196-197), no mask (lceT5.py:53) ----
 * Every matrix (q, k, v | ctx | dctx | dq, dk, dv) is addressed by a row stride and a head stride: row (n*S+s) of head h
 * starts at element (n*S+s)*ld + h*hs; hs = 64 for heads side by side in rows of ld elements, or ld = 64 and
 * hs >= rows * 64 for head-major matrices (paced_gemm_bf16_heads writes and reads those).
 * lse [N,H,S] fp32 (log-sum-exp of the biased scores); bias_tbl [H, 2S-1] fp32 or NULL.
 * bwd writes dq, dk, dv (same addressing, row stride ldd) and adds into dbias_tbl (may be NULL):
 * [dbias_replicas, H, 2S-1]; workgroups spread their adds over the replicas so that same-address
 * atomics do not serialise; the caller sums the replicas (paced_relbias_table_bwd does).
 * Every row stride and head stride is a multiple of 8 elements, and ctx, dq, dk, dv are 16-byte aligned: the outputs are
 * stored as whole 128-byte rows, 16 bytes per lane. */
int paced_enc_attn_fwd(const void* q, const void* k, const void* v, int ld, long long hs, void* ctx, int ldo, long long hso, float* lse,
                       const float* bias_tbl, int N, int S, int H, int dk, float drop_p, uint32_t drop_seed, void* stream);
int paced_enc_attn_bwd(const void* q, const void* k, const void* v, int ld, long long hs, const void* ctx, int ldo, long long hso,
                       const void* dctx, int lddo, long long hsdo, const float* lse, const float* bias_tbl,
                       void* dq, void* dk_, void* dv, int ldd, long long hsd, float* dbias_tbl, int dbias_replicas,
                       int N, int S, int H, int dk, float drop_p, uint32_t drop_seed, void* stream);

/* ---- decoder attention for the 2-token target, HF:459-498 ---------------------------------------
 * self: qkv [N*2, ld] = [q | k | v]; causal; bias2[h*2+0] = bias at offset 0, bias2[h*2+1] at -1.
 * cross: 2 queries per sequence (qc [N*2, ldq]) over S encoder positions; kv [N*S, ldkv] holds k at
 *        column h*64 and v at column voff + h*64; zero position bias (HF:337-342); lse [N,H,2]. */
int paced_dec_self_attn_fwd(const void* qkv, int ld, void* ctx, int ldo, const float* bias2, int N, int H, int dk,
                            float drop_p, uint32_t drop_seed, void* stream);
int paced_dec_self_attn_bwd(const void* qkv, int ld, const void* dctx, int lddo, void* dqkv, int ldd,
                            const float* bias2, float* dbias2, int N, int H, int dk,
                            float drop_p, uint32_t drop_seed, void* stream);
int paced_cross_attn_fwd(const void* qc, int ldq, const void* kv, int ldkv, int voff, void* ctx, int ldo, float* lse,
                         int N, int S, int H, int dk, float drop_p, uint32_t drop_seed, void* stream);
int paced_cross_attn_bwd(const void* qc, int ldq, const void* kv, int ldkv, int voff, const void* ctx, int ldo,
                         const void* dctx, int lddo, const float* lse, void* dqc, int lddq, void* dkv, int lddkv,
                         int N, int S, int H, int dk, float drop_p, uint32_t drop_seed, void* stream);

/* Decoder cross-attention WITHOUT key/value projections of the encoder states (same function as HF
 * modeling_t5.py:459-498 for EncDecAttention with 2 target positions, re-associated): with C = 2H projected
 * queries qk[n][c] = q[n][t][h] Wk_h (c = t*H + h, d wide; a batched paced_gemm_bf16_batched product),
 *   s[c][k] = qk[c] . enc[n][k]     p = softmax_k(s)     z[n][c] = sum_k dropout(p)[c][k] enc[n][k]
 * and the context of head h is z[c] Wv_h^T (another batched product).  enc [N*S, d] is read once; lse [N][C].
 * Dropout counter: (n*C + c)*S + k.  Requires 2H <= 32 and d a multiple of 64 up to 1024. */
int paced_cross2_fwd(const void* enc, int ld, const void* qk, void* z, float* lse,
                     int N, int S, int H, int d, float drop_p, uint32_t drop_seed, void* stream);
/* Backward of paced_cross2_fwd given dz [N][C][d]: dqk [N][C][d], and the factors of the encoder-state gradient
 *   d enc[n][k] = sum_c ds[c][k] qk[c] + dropout(p)[c][k] dz[c]
 * written for a later token-contraction product over all decoder layers: wslot + n*wstride holds ds then
 * dropout(p), [2C][wld] bf16 (wld >= S, a multiple of 4; columns S..wld-1 are not written); uslot + n*ustride
 * holds qk then dz, [2C][d] bf16 (strides in elements). */
int paced_cross2_bwd(const void* enc, int ld, const void* qk, const void* dz, const void* z, const float* lse,
                     void* dqk, void* wslot, long long wstride, int wld, void* uslot, long long ustride,
                     int N, int S, int H, int d, float drop_p, uint32_t drop_seed, void* stream);

/* ---- loss: nn.CrossEntropyLoss(ignore_index, reduction='none'), lceT5.py:110,117-118 ----------
 * loss[r] = logsumexp(logits[r,:]) - logits[r,label[r]] (0 for ignored rows);
 * dlogits (bf16, may be NULL) = (softmax - onehot) * gscale. */
int paced_cross_entropy(const float* logits, const long long* labels, float* loss, void* dlogits,
                        int rows, int V, int ldl, long long ignore_index, float gscale, void* stream);
/* lceT5.py:119-121 after the caller's view at :134,151: ce[r] = pce[r] + sum|mean nce[r*n..(r+1)*n) */
int paced_lce_group(const float* pce, const float* nce, float* ce, int R, int n, int use_mean, void* stream);
/* LCEWeights.forward, lceT5.py:91-99, vectorised, plus the reductions training_step needs (:138-147):
 * w[i] = loss[i] <= eta ? loss[i]/eta : 0;
 * stats = { mean(w), mean(w*loss), d mean(w*loss) / d eta, mean(loss) }.  eta is read on the device. */
int paced_lce_weights(const float* loss, const float* eta, float* w, float* stats, int R, void* stream);
/* Threshold weights of pacednegatives/weights.py, vectorised (the reference loops over the batch on the host), with
 * their slope in eta (dw may be NULL):
 *   form 0 = LCEWeights (weights.py:47-78):  w = l <= eta ? l/eta     : l*mask*eta
 *   form 1 = EtaWeights (weights.py:10-45):  w = l <= eta ? 1 - l/eta : l*mask*eta */
int paced_eta_weights(const float* loss, const float* eta, float mask, int form, float* w, float* dw, int R, void* stream);
/* Meta objective of the eta-bound pace schedule, old/eta_bound.py:73-78: ce = (pce+nce)/2, v = EtaWeights(ce),
 * J = mean(pce*v) + mean(nce*v) - sum(v).  v[R] receives the weights;
 * out = { J, dJ/d eta, mean(v), mean(ce < eta) (the success rate of old/eta_bound.py:57-58) }. */
int paced_eta_bound_meta(const float* pce, const float* nce, const float* eta, float mask, float* v, float* out, int R, void* stream);

/* ---- transformers.AdamW as built at lceT5.py:171-172 (betas .9/.999, eps 1e-6, wd 0) ------------
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= step_size * m/(sqrt(v)+eps); p -= decay*p;
 * g is multiplied by grad_scale first (1/world_size after a summing all-reduce).  step_size and
 * decay are host scalars (lr * sqrt(1-b2^t)/(1-b1^t) and lr*wd).  p_bf16 (may be NULL) receives the
 * rounded parameters the kernels read.  clear_grad != 0: g is set to zero once read (optimizer.zero_grad(), lceT5.py:153,
 * folded into the update: the next backward pass accumulates into g). */
int paced_adamw(float* p, float* g, float* m, float* v, void* p_bf16, size_t n,
                float beta1, float beta2, float eps, float step_size, float decay, float grad_scale, int clear_grad,
                void* stream);

/* ---- paced negative-pool selection ---------------------------------------------------------------
 * Replaces the retriever call of pacednegatives/utilities/compute_all_bm25.py:26-27,36 by a dense
 * scorer and keeps that file's contract (:37-44): per query the top-k by score.  scores = Q . Pool^T
 * (bf16 inputs, fp32 accumulate), ordered by (score descending, pool index ascending).
 * queries [Q, D] bf16, pool [Np, D] bf16, D % 64 == 0; out_scores f32 [Q, k], out_idx int64 [Q, k]
 * (best first; index_base is added to every index so that pool shards can be merged).
 * workspace: paced_pool_topk_workspace_bytes(Q, k) bytes of device memory. */
size_t paced_pool_topk_workspace_bytes(int Q, int k);
int paced_pool_topk(const void* queries, const void* pool, int Q, long long Np, int D, int k,
                    long long index_base, float* out_scores, long long* out_idx,
                    void* workspace, size_t workspace_bytes, void* stream);
/* k-way merge of per-shard results (shards concatenated along the last axis: [Q, W*k]) into the
 * global top-k under the same order. */
int paced_topk_merge(const float* scores, const long long* idx, int Q, int total, int k,
                     float* out_scores, long long* out_idx, void* stream);

/* ---- paced_pool_topk for callers that cannot allocate -------------------------------------------
 * No counterpart in the reference.  For callers that may not call hipMalloc or take a workspace from an
 * allocator: code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same
 * arguments, results and order as paced_pool_topk, without the workspace: the candidate lists live in an
 * array inside the library's device image (paced_pool_topk_static_bytes() bytes, about 17 MiB, held by every
 * GPU the library runs on until the process ends; its per-device address is looked up on the first call on
 * that device).  Limits: Q <= paced_pool_topk_static_max_queries() (64), k <= paced_pool_topk_static_max_k()
 * (1024); larger requests fail with an error naming the limit.  The call allocates and frees nothing.  All
 * calls on one GPU share that GPU's array: calls that overlap there (two streams, two threads) race; calls
 * on one stream, or otherwise ordered, are safe, and a call needs nothing cleared by its predecessor. */
size_t paced_pool_topk_static_bytes(void);
int paced_pool_topk_static_max_queries(void);
int paced_pool_topk_static_max_k(void);
int paced_pool_topk_static(const void* queries, const void* pool, int Q, long long Np, int D, int k,
                           long long index_base, float* out_scores, long long* out_idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PACED_HIP_H */
