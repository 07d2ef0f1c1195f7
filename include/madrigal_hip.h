/*
 * madrigal_hip.h -- C ABI of libmadrigal_hip.so: MI355X (gfx950) kernels for Madrigal's
 * encode -> fuse -> pairwise-score path.
 *
 * This is the drop-in boundary below the Python classes in madrigal_amd/ (which mirror
 * madrigal.models of the reference).  The reference owns no native code: every entry point
 * replaces an implicit device-kernel call site inside a third-party wheel (ATen/cuBLAS,
 * torch_scatter, torchdrug, PyG); the call site is cited as <reference file>:<line>.
 *
 * Conventions (all entry points)
 *   - plain pointers and sizes, no torch types; every pointer is a DEVICE pointer unless the
 *     name ends in _host; tensors are dense row-major fp32 unless stated;
 *   - returns 0 on success, a negative MDG_E* code on failure; mdg_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - never allocates or frees device memory: scratch is passed in as `workspace`, sized by the
 *     matching *_workspace_bytes() query; never synchronises: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream) and the call returns;
 *   - re-entrant, no internal threads, safe under hipGraph capture; no global mutable state apart
 *     from the two diagnostics hooks at the end of this header (tuning switches, clock stamps).
 */
#ifndef MADRIGAL_HIP_H
#define MADRIGAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDG_OK 0
#define MDG_EINVAL (-1)   /* bad argument (shape, alignment, enum) */
#define MDG_EWORKSPACE (-2) /* workspace too small / missing */
#define MDG_ELAUNCH (-3)  /* HIP launch failed */
#define MDG_EUNSUPPORTED (-4)

/* Arithmetic of the matrix products.  Inputs and outputs stay fp32. */
enum mdg_precision {
  MDG_PREC_F32 = 0,    /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate        */
  MDG_PREC_BF16X3 = 1, /* each fp32 operand split hi+lo bf16; hi*hi + hi*lo + lo*hi on the
                          bf16 MFMA, fp32 accumulate: ~2^-17 relative per product             */
  MDG_PREC_BF16 = 2,   /* operands rounded to bf16 once, fp32 accumulate (cfg "bf16")          */
  MDG_PREC_F16 = 3     /* operands rounded to IEEE half once (v_mfma_f32_32x32x16_f16), fp32
                          accumulate: BASELINE configs[4] "fp16 bilinear head".  Accepted by
                          mdg_bilinear_allpairs only; inputs must stay inside the half range.   */
};

/* What the all-pairs head does with each score tile. */
enum mdg_bilinear_epilogue {
  MDG_EPI_STORE = 0,         /* out[l,i,j] = S            (raw logits, as the reference returns) */
  MDG_EPI_STORE_SIGMOID = 1, /* out[l,i,j] = sigmoid(S)   (train_ddi_batch.py:285)               */
  MDG_EPI_ROWSTATS = 2,      /* nothing is materialised: stats[l,i,0] = sum_j S, stats[l,i,1] =
                                max_j S (roofline stress runs whose [L,N,N] cannot exist)        */
  MDG_EPI_TRIKEYS = 3        /* for callers whose product is ranks (notebooks/normalize_scores.py:36-85 reads
                                the strict lower triangle only): out[l,i,j] for j < i = the order-preserving
                                uint32 key of S[l,i,j] (the bits mdg_rank_normalize sorts), the same value the
                                STORE epilogue puts there; entries with j >= i are unspecified (never written
                                outside the 256 x 256 blocks on the diagonal): half the store stream.
                                Symmetric sweep only (z_head == z_tail, symmetric W, row pitch >= n rounded
                                up to 4); consumed by mdg_rank_normalize_keys_ld.                          */
};

/* Activation fused into mdg_linear's epilogue (madrigal/models/models.py:31, actn2actfunc). */
enum mdg_activation {
  MDG_ACT_NONE = 0, MDG_ACT_RELU = 1, MDG_ACT_GELU = 2 /* exact erf form */, MDG_ACT_SIGMOID = 3, MDG_ACT_TANH = 4,
  MDG_ACT_LEAKYRELU = 5, MDG_ACT_SOFTPLUS = 6, MDG_ACT_SELU = 7
};

const char* mdg_last_error(void);
/* "gfx950" build tag, and the ABI version (bumped on any signature change). */
const char* mdg_build_arch(void);
int mdg_abi_version(void);

/* Diagnostics.  The MDG_* environment variables are experiment switches that select between kernel schedules
 * with identical results; each is read once per process, at its first use.  mdg_tuning_reload() makes every
 * switch re-read the environment at its next use (tests and timing scripts flip one between two launches).
 * mdg_debug_bilinear_stamps hands mdg_bilinear_allpairs a device buffer of `entries` uint64 that receives
 * {shader cycles, 100 MHz ticks} per workgroup of each following launch whose grid fits; NULL switches it off.
 * Neither is meant to be called concurrently with launches from other threads. */
void mdg_tuning_reload(void);
void mdg_debug_bilinear_stamps(void* buffer, int64_t entries);

/* ------------------------------------------------------------------ bilinear DDI head ---- */

/* W_sym[l] = triu(W[l]) + triu(W[l],1)^T for l in [0,L): the `Symmetric` parametrisation the
 * reference recomputes on every forward.  Replaces madrigal/models/models.py:522-524.
 * w_original, w_sym: [L,D,D].  In place (w_sym == w_original) is allowed. */
int mdg_symmetrize(const float* w_original, float* w_sym, int64_t L, int64_t D, void* stream);

/* Scratch for mdg_bilinear_allpairs (hi/lo bf16 images of z_tail and W_sym; 0 for F32). */
size_t mdg_bilinear_allpairs_workspace_bytes(int64_t n_head, int64_t n_tail, int64_t n_labels, int64_t D,
                                             int precision);

/* All-pairs bilinear scores  S[l,i,j] = z_head[i]^T W_sym[l] z_tail[j]  for every outcome l,
 * head drug i and tail drug j, with the association order of the reference ((z W) z^T, the
 * inner product rounded to fp32).  Replaces BilinearDDIScorer.bilinear/forward,
 * madrigal/models/models.py:537-547 (two batched cuBLAS GEMMs there), and one chunk of the
 * scoring loop madrigal/evaluate/predict.py:420-429 when called on a label range (pass
 * w_sym + lo*D*D and n_labels = hi-lo).
 *   z_head [n_head,D], z_tail [n_tail,D], w_sym [n_labels,D,D] (already symmetrised), D == 128.
 *   epilogue STORE / STORE_SIGMOID: out [n_labels,n_head,n_tail] fp32, outcome-major.
 *   epilogue ROWSTATS: out [n_labels,n_head,2] fp32.
 * No alignment requirement beyond 4 bytes on out; z_* and w_sym must be 16-byte aligned. */
int mdg_bilinear_allpairs(const float* z_head, const float* z_tail, const float* w_sym, float* out,
                          int64_t n_head, int64_t n_tail, int64_t n_labels, int64_t D, int precision,
                          int epilogue, void* workspace, size_t workspace_bytes, void* stream);
/* The same with a row pitch: out[(l * n_head + i) * ldo + j], ldo >= n_tail floats.  The MI355X-native layout of a score tensor
 * whose n_tail is not a multiple of 32: with ldo = n_tail rounded up to 32 every row starts on a 128-byte line and the head's
 * 16-byte stores stay whole-line (a contiguous [L,N,N] with such an N is written correctly but with cache-line-straddling
 * stores, 2-3x slower).  The padding columns [n_tail, ldo) may be overwritten with unspecified values. */
int mdg_bilinear_allpairs_ld(const float* z_head, const float* z_tail, const float* w_sym, float* out, int64_t ldo, int64_t n_head,
                             int64_t n_tail, int64_t n_labels, int64_t D, int precision, int epilogue, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------ dense blocks ---- */

/* Y = alpha * act( (X W^T + bias) * scale + shift ) + beta * R      X [M,K] ldx, W [N,K] ldw (nn.Linear
 * layout), Y [M,N] ldy, R [M,N] ldr; bias/scale/shift [N]; any of bias, scale+shift, R may be NULL.
 * Replaces every nn.Linear (+ eval BatchNorm1d folded into scale/shift, + activation, + residual) on
 * the path: MLPEncoder/MLPAdaptor madrigal/models/models.py:178-180,516-518; chemCPA MLP
 * madrigal/chemcpa/chemCPA/model.py:226-231; nn.TransformerEncoderLayer linears models.py:366;
 * embed2latent / latent2embed models.py:411,443; GIN and HGT projections (third-party wheels).
 * K, ldx, ldw multiples of 4 (zero-pad the inner dimension), x and w 16-byte aligned.  ldr == 0 broadcasts one
 * residual row.  The kernel stages "operand images" (K zero-padded to 32; bf16 modes: hi [+ lo] bf16 planes): the
 * image of x (and of w unless w_packed is given) is written into `workspace` by a fused pre-pass; a weight that is
 * reused can be packed once with mdg_pack_operand and passed as w_packed (w may then be NULL). */
size_t mdg_pack_operand_bytes(int64_t rows, int64_t K, int precision);        /* 0: the raw fp32 tensor is used as is */
int mdg_pack_operand(const float* src, int64_t ld, int64_t rows, int64_t K, int precision, void* dst, size_t dst_bytes,
                     void* stream);
/* Image of src^T from src [rows, cols] in one transposing pass (16-bit modes; mdg_pack_operand_bytes(cols, rows, precision) bytes):
 * the weight image of the backward product dx = g W (a dense block with weight W^T) without materialising W^T. */
int mdg_pack_operand_transposed(const float* src, int64_t ld, int64_t rows, int64_t cols, int precision, void* dst, size_t dst_bytes,
                                void* stream);
size_t mdg_linear_workspace_bytes(int64_t M, int64_t N, int64_t K, int precision, int w_is_packed);
int mdg_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, const void* w_packed, float* y, int64_t ldy, int64_t M,
               int64_t N, int64_t K, const float* bias, const float* scale, const float* shift, int activation,
               const float* residual, int64_t ldr, float alpha, float beta, int precision, void* workspace,
               size_t workspace_bytes, void* stream);
/* Training form: y = dropout(act(x W^T + b), drop_p, drop_seed) + beta * residual in the same epilogue (nn.TransformerEncoderLayer's
 * x + dropout1(out_proj(...)), x + dropout2(linear2(...)), dropout(activation(linear1(x)))): the mask and scale are exactly those of
 * mdg_dropout(seed) on the contiguous [M,N] result, so the fused block equals the three separate launches bit for bit. */
int mdg_linear_dropout(const float* x, int64_t ldx, const float* w, int64_t ldw, const void* w_packed, float* y, int64_t ldy,
                       int64_t M, int64_t N, int64_t K, const float* bias, int act, const float* residual, int64_t ldr, float beta,
                       float drop_p, uint64_t drop_seed, int precision, void* workspace, size_t workspace_bytes, void* stream);

/* Row-wise LayerNorm (nn.LayerNorm, biased variance): y = (x - mean) / sqrt(var + eps) * gamma + beta.
 * norm1/norm2 and the x-attn norms of the fusion transformer, models.py:366,372-373; LayerNorm inside
 * MLPAdaptor, models.py:492.  d multiple of 4, <= 2048. */
int mdg_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy, int64_t rows,
                  int64_t d, float eps, void* stream);
/* LayerNorm that also writes y as the packed operand image of the dense block consuming it (mdg_pack_operand_bytes(rows, d,
 * precision) bytes, bit-identical to mdg_linear's own pre-pass; 16-bit modes, d a multiple of 32; y may be NULL when only
 * the image is wanted), and mdg_linear on such an
 * image: the norm -> projection pairs of nn.TransformerEncoderLayer (models.py:366) without the pre-pass in between. */
int mdg_layernorm_packed(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy, int64_t rows,
                         int64_t d, float eps, int precision, void* y_packed, size_t y_packed_bytes, void* stream);
size_t mdg_linear_packed_x_workspace_bytes(int64_t M, int64_t N, int64_t K, int precision, int w_is_packed);
int mdg_linear_packed_x(const void* x_packed, int64_t M, int64_t K, const float* w, int64_t ldw, const void* w_packed, float* y,
                        int64_t ldy, int64_t N, const float* bias, int act, const float* residual, int64_t ldr, float alpha,
                        float beta, int precision, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------- cross-modal fusion ---- */

/* Token sequence of the fusion transformer: seq[i] = [cls?] str kg cv [bottleneck x nb] tx_0..tx_15
 * (each a 128-float row), optional per-token L2 normalisation, then + pe[s] for s < pe_len.
 * Replaces the stack / cat / repeat glue and PositionEncoding* of madrigal/models/models.py:772-775,
 * 799-804,818-822,849-852,581-603.  tx_emb is [16 * n_src, 128], cell-line major.  rows (nullable) selects
 * source drug rows (the multi-modal subset of fusion='transformer_uni_proj', models.py:784-790).
 * token_index (nullable, [n_tok] values i*S+s): emit only those tokens, densely packed -> seq [n_tok,128]. */
int mdg_assemble_tokens(const float* str_emb, const float* kg_emb, const float* cv_emb, const float* tx_emb,
                        const float* bottleneck, const float* cls, const float* pe, const int64_t* rows, const int64_t* token_index,
                        int64_t n_tok, float* seq, int64_t n, int64_t n_src, int nb, int has_cls, int pe_len, int normalize,
                        int64_t D, void* stream);

/* Multi-head self-attention core for S <= 32 tokens per drug (scores, masks, softmax, PV) on the fp32
 * matrix cores; projections are mdg_linear calls.  qkv [n*S, 3d] = q|k|v (in_proj output), out [n*S, d].
 * kpm_bits[i] bit j = key j of drug i is padding (src_key_padding_mask), src_bits[s] bit j = query s
 * may not attend key j (the [S,S] bottleneck mask, models.py:813-816); probs (nullable) receives the
 * per-head attention weights [n,H,S,S] (need_weights=True, average_attn_weights=False, models.py:388-399).
 * Compact mode (row_start != NULL): padded tokens are not stored at all -- qkv / out hold only each drug's
 * live tokens, rows row_start[i] .. row_start[i+1]-1 (at most 32), and row_bits[r] bit j says that query row r
 * may not attend its drug's j-th live key (kpm_bits / src_bits are ignored, probs must be NULL).  Padded
 * tokens never influence live ones, so the live rows equal the dense result.
 * Replaces nn.MultiheadAttention inside nn.TransformerEncoderLayer, models.py:366-367,412. */
int mdg_fusion_attention(const float* qkv, int64_t ld, float* out, int64_t ldo, const uint32_t* kpm_bits,
                         const uint32_t* src_bits, float* probs, const int64_t* row_start, const uint32_t* row_bits, int64_t n,
                         int S, int H, int dh, void* stream);

/* Cross-attention pooling with one learned query shared by all drugs: out[i,h,:] = softmax_j(q_h . K_ijh / sqrt(dh)) V_ijh.
 * q_proj [H*dh] (already projected), kv_proj [n*Tk, 2*H*dh] = K|V of the Tk allowed key tokens of each drug.
 * Replaces x_attn_mha_layer, madrigal/models/models.py:430-438. */
int mdg_xattn_pool(const float* q_proj, const float* kv_proj, int64_t ld, float* out, int64_t ldo, int64_t n, int Tk, int H, int dh,
                   void* stream);

/* y = x / max(||x||_2, 1e-12) per row (F.normalize; models.py:849-850,858-859,872,892-893,947-949). d % 4 == 0. */
int mdg_l2_normalize(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t d, void* stream);

/* Masked pooling over the S <= 32 tokens [n,S,128] of each drug; tokens whose bit is set in mask_bits[i] are
 * skipped.  mode 0 = mean, 1 = sum, 2 = max.  Replaces the torch_scatter scatter_mean / scatter_add /
 * scatter_max call sites madrigal/models/models.py:447,451,873,878. */
int mdg_token_pool(const float* tokens, const uint32_t* mask_bits, float* out, int64_t n, int S, int64_t D, int mode, void* stream);

/* ------------------------------------------------------------------- graph aggregations ---- */

/* out[v] = (self_coef_add + *self_coef_dev) * x_self[v] + sum_{e in CSR row v} w[e] * x[col[e]]   (mean: sum / count)
 * col == NULL: the row's neighbours are the contiguous rows [rowptr[v], rowptr[v+1]) of x (segment read-out).
 * Replaces torchdrug GraphIsomorphismConv.message_and_aggregate (sparse.mm + scatter_add) and MeanReadout
 * behind madrigal/models/models.py:217,720-721.  F multiple of 4, <= 256. */
int mdg_csr_aggregate(const float* x, int64_t ldx, const int64_t* rowptr, const int64_t* col, const float* edge_weight,
                      const float* x_self, int64_t ld_self, const float* self_coef_dev, float self_coef_add, int mean, float* out,
                      int64_t ldo, int64_t n_dst, int64_t F, void* stream);

size_t mdg_hgt_attention_workspace_bytes(int64_t n_items, int heads);

/* HGT edge attention + aggregation for one destination node type: per-head softmax over ALL incoming edges
 * of a node (every edge type), weighted sum of relation-transformed values, optional GELU.
 * q [n_dst, ldq>=128]; edge e reads k' at kv + col[e]*ldkv and v' 128 floats later (relation transforms and
 * p_rel/sqrt(D) already applied; ldkv >= 128, with ldkv == 128 the value is simply the next row);
 * col[e] = kv row of edge e, edges sorted by destination; work items (item_dst, item_begin, item_end) split
 * long destination rows, item_ptr [n_dst+1] = items of each destination.
 * Replaces PyG HGTConv.propagate/message (edge softmax + scatter-add) behind models.py:76-79,90-94. */
int mdg_hgt_attention(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const int64_t* col, const int64_t* item_dst,
                      const int64_t* item_begin, const int64_t* item_end, int64_t n_items, const int64_t* item_ptr, float* out,
                      int64_t ldo, int64_t n_dst, int heads, int64_t F, int apply_gelu, void* workspace, size_t workspace_bytes,
                      void* stream);
/* The same for ALL destination node types of a conv in one launch: destinations numbered across the types, query row of
 * destination d at q_base + q_off[d] floats (the types' projection rows differ in width), items / edges / item_ptr of the types
 * concatenated, out [n_dst,128].  (PyG HGTConv runs its message passing per edge type: models.py:76-79.) */
int mdg_hgt_attention_rows(const float* q_base, const int64_t* q_off, const float* kv, int64_t ldkv, const int64_t* col,
                           const int64_t* item_dst, const int64_t* item_begin, const int64_t* item_end, int64_t n_items,
                           const int64_t* item_ptr, float* out, int64_t ldo, int64_t n_dst, int heads, int apply_gelu,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------ losses ---- */

/* InfoNCE finish of SimCLR_NovelDDI.contrastive_loss (madrigal/models/simclr.py:74-108): given
 * sim = F F^T [2B,2B] of the 2B L2-normalised features, apply the too-hard-negative mask ([B,B] bytes,
 * nullable; tiled 2x2, filled with -1e9), drop the diagonal, divide by the temperature -> logits
 * [2B,2B-1] (nullable), labels [2B,2B-1] (nullable, 1 at the other view of the same drug),
 * row_loss [2B] and loss [1] = mean soft-label cross entropy. */
int mdg_infonce_finish(const float* sim, const uint8_t* too_hard_neg, float* logits, float* labels, float* row_loss, float* loss,
                       int64_t B, float temperature, void* stream);
/* Backward of mdg_infonce_finish: dsim [2B,2B] = d loss / d sim (zero diagonal, zero at too-hard negatives), scaled by
 * dloss[0] (device scalar).  The caller chains it through sim = F F^T: dF = (dsim + dsim^T) F. */
int mdg_infonce_bwd(const float* sim, const uint8_t* too_hard_neg, const float* dloss, float* dsim, int64_t B, float temperature,
                    void* stream);

/* pred[e] = f(scores[label[e], head[e], tail[e]]) with f = sigmoid (apply_sigmoid) or identity, and the
 * mean BCE of pred against target with nn.BCELoss's -100 clamp: term [n] and loss [1] (both nullable
 * together).  Replaces the advanced-index gather + BCELoss of train_ddi_batch.py:285-288. */
int mdg_gather_bce(const float* scores, int64_t n_labels, int64_t n_head, int64_t n_tail, const int64_t* label, const int64_t* head,
                   const int64_t* tail, const float* target, float* pred, float* term, float* loss, int64_t n, int apply_sigmoid,
                   void* stream);

/* ------------------------------------------------------------------------- tuning switches ---- *
 * Environment variables read by the library (csrc: once, re-read after mdg_tuning_reload()) or by the Python host layer.  Each
 * changes speed or selects between paths that produce the same results within the tests' bounds; none is needed for normal use.
 * Round 5 removed every switch whose experiment was decided (35 of them: store schedules, wave counts, stream-K, the LSD
 * look-back, ...).  What is left, with defaults:
 *   MDG_RANKS_MSD        [1]  0: rank normalisation by the four-pass LSD sort only (what N > 5793 and flagged outcomes take)
 *   MDG_RANKS_GROUP      [8]  outcomes per launch group of the MSD path (its scratch is reused from group to group)
 *   MDG_RANKS_DIRECT     [0]  test hook: the LSD sort's last pass stores ranks one by one (what N > 16256 takes) at any N
 *   MDG_BILINEAR_SYMMETRIC [1] 0: z_head == z_tail takes the general sweep instead of the symmetric one
 *   MDG_LINEAR_TILE      [0 = by shape]  128 / 256: force the dense block's tile shape
 *   MDG_LINEAR_RAWX      [1]  0: the 128-tile kernel's x through an operand pre-pass instead of rounded while staged
 *   MDG_LINEAR_TAIL128   [1]  0: no row split of a dense block's last partial round of 256-tiles
 *   MDG_KG_GRAPH         [per step kind]  0 / 1: the KG pass of a training step eager / replayed as hipGraphs
 *   MDG_SHARE_SIDES      [1]  0: dropout-free encoders run once per side of a step even when both sides are one batch
 *   MDG_FUSE_SIDES       [1]  0: the two sides of a finetune step through the fusion transformer in two passes
 *   MDG_FUSE_VIEWS       [1]  0: the two views of a contrastive step through the per-row stages in two passes
 *   MDG_SHARD_KG         [1]  0: multi-GPU inference keeps the KG encoder replicated instead of destination-partitioned
 * Build / bench only: MDG_EXTRA_HIPCC_FLAGS (madrigal_amd/build.py, e.g. -DMDG_RANK_STAMPS), MDG_BENCH_* (bench.py). */

/* ---------------------------------------------------------------------- rank normalisation ---- */

size_t mdg_rank_normalize_workspace_bytes(int64_t n_outcomes, int64_t N);
/* 1 when a call of this size takes the exact-layout MSD path first (the default up to N = 5793, MDG_RANKS_MSD=0 turns it off: one
 * partition + an in-LDS sort per bucket; ranks.hip), 0 when it runs the four-pass LSD sort only.  On the MSD path the first
 * n_outcomes uint32 words of the workspace hold, after the call, one flag per outcome: non-zero = that outcome was handed to the LSD
 * kernels (point masses of equal scores: a bucket of 65 536 keys or more, or more than 1 024 keys in one fine bin).  Diagnostics:
 * the ranks are the same bits. */
int mdg_rank_normalize_fast_path(int64_t n_outcomes, int64_t N);

/* Per outcome l: out[l,i,j] = out[l,j,i] = rank(scores[l,i,j] among the strict lower triangle i > j, ascending,
 * 1-based, ties by flat index) / (N(N-1)/2), diagonal 0; the division is done in double and rounded to fp32
 * like numpy's int64 / float -> float32 store.  Replaces classwise_normalized_rank_3d_numpy + run_slice,
 * notebooks/normalize_scores.py:36-74 (two CPU argsorts of N^2 keys per outcome).  scores, out [n_outcomes,N,N];
 * out may alias scores only if n_outcomes fits one call.  Scores must be finite and < 1e7 (the reference's
 * mask value). */
int mdg_rank_normalize(const float* scores, float* out, int64_t n_outcomes, int64_t N, void* workspace, size_t workspace_bytes,
                       void* stream);
/* The same on row-pitched tensors (mdg_bilinear_allpairs_ld): scores[(l * N + i) * lds + j], out[(l * N + i) * ldo + j]. */
int mdg_rank_normalize_ld(const float* scores, int64_t lds, float* out, int64_t ldo, int64_t n_outcomes, int64_t N, void* workspace,
                          size_t workspace_bytes, void* stream);
/* The same from the order keys of the strict lower triangle as mdg_bilinear_allpairs_ld(..., MDG_EPI_TRIKEYS) writes them
 * (keys[(l * N + i) * ldk + j] for j < i; nothing else of `keys` is read): the head's store stream is halved and the
 * ranks equal those of the materialised scores bit for bit.  out may be the memory of keys (n_outcomes of one call). */
int mdg_rank_normalize_keys_ld(const uint32_t* keys, int64_t ldk, float* out, int64_t ldo, int64_t n_outcomes, int64_t N,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Elementwise geometric mean of K <= 8 equally shaped fp32 tensors (the 5-seed ensembling of normalised ranks,
 * scipy.stats.mstats.gmean in notebooks/generate_embeddings.ipynb): out = exp(mean_k log x_k) in fp32; 0 where any
 * x_k <= 0.  inputs_host is a HOST array of K device pointers (16-byte aligned); n = elements of each tensor. */
int mdg_gmean(const float* const* inputs_host, int K, float* out, int64_t n, void* stream);

/* Parameter space of the HGT conv in training (PyG 2.3 HGTConv, madrigal/models/models.py:76-79,90-94): the composite projection
 * weights of every projected node type from the live parameters -- rows [Wq_t | K_r V_r for every used relation r leaving t] with
 * K_r = blockdiag_h(k_rel[h,r]^T p_rel[r][h] / sqrt(D)) Wk_t, V_r = blockdiag_h(v_rel[h,r]^T) Wv_t (biases likewise) -- and the
 * gradients of kqv_lin / k_rel / v_rel / p_rel from the gradient of those rows.  w_ptrs / b_ptrs: DEVICE arrays of n_types float
 * pointers (kqv weight [3F,cin], rows K | Q | V; bias [3F]); p_ptrs: device array of n_edge_types pointers ([heads] each); rel_*:
 * per used relation its edge type, its source type (index into w_ptrs) and the first row of its K block in big_w (V block: + F);
 * type_row: first row of each type's block.  bwd writes grads = [per type: dW (3F x cin) | db (3F)] | dk_rel | dv_rel | dp_rel
 * [n_edge_types, heads] (zero for unused relations).  Exact fp32, fixed summation order. */
int mdg_hgt_composite_fwd(const void* w_ptrs, const void* b_ptrs, const float* k_rel, const float* v_rel, const void* p_ptrs, const int* rel_r,
                          const int* rel_src, const int* rel_row, int n_rel, const int* type_row, int n_types, float* big_w, float* big_b,
                          int cin, int heads, int n_edge_types, int F, void* stream);
int mdg_hgt_composite_bwd(const void* w_ptrs, const void* b_ptrs, const float* k_rel, const float* v_rel, const void* p_ptrs, const int* rel_r,
                          const int* rel_src, const int* rel_row, int n_rel, const int* type_row, int n_types, const float* dbig_w,
                          const float* dbig_b, float* grads, int cin, int heads, int n_edge_types, int F, void* stream);

/* mdg_hgt_attention that also returns the softmax statistics stats [n_dst, heads, 2] = (max logit, denominator) the
 * backward pass needs; and that backward pass.  dout is the gradient at the attention output BEFORE the activation
 * (call the forward with apply_gelu = 0 and differentiate the activation separately), out_pre that forward output.
 * Reversed edge lists (built by the caller from col / the destination of each edge): the nnz edges stably sorted by
 * key row; t_edge[e'] = forward edge id, t_dst[e'] = its destination; the n_src_rows distinct key rows t_row are cut
 * into work items (t_item_begin/end, t_item_ptr [n_src_rows+1]; t_item_row [n_src_items] = index into t_row of each
 * item's key row, or null: with it a key row of ONE item is written by the gather kernel itself, not through a partial).
 * dq [n_dst,128]; dkv has the layout of kv (lddkv = 128: value rows follow key rows) and only the rows of this call's
 * key rows are written — zero-fill it first.  kv16: optional bf16 mirror of kv (mdg_f32_to_bf16; ldkv must be 128). */
int mdg_hgt_attention_stats(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const int64_t* col,
                            const int64_t* item_dst, const int64_t* item_begin, const int64_t* item_end, int64_t n_items,
                            const int64_t* item_ptr, float* out, int64_t ldo, int64_t n_dst, int heads, int64_t F,
                            int apply_gelu, float* stats, const void* kv16, void* workspace, size_t workspace_bytes, void* stream);
size_t mdg_hgt_attention_bwd_workspace_bytes(int64_t nnz, int64_t n_items, int64_t n_src_items, int heads);
int mdg_hgt_attention_bwd(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const int64_t* col, int64_t nnz,
                          const int64_t* item_dst, const int64_t* item_begin, const int64_t* item_end, int64_t n_items,
                          const int64_t* item_ptr, int64_t n_dst, const float* dout, int64_t lddo, const float* out_pre,
                          int64_t ldp, const float* stats, int heads, const int64_t* t_edge, const int64_t* t_dst,
                          const int64_t* t_item_begin, const int64_t* t_item_end, int64_t n_src_items,
                          const int64_t* t_item_ptr, const int64_t* t_row, int64_t n_src_rows, const int64_t* t_item_row,
                          float* dq, int64_t lddq, float* dkv, int64_t lddkv, const void* kv16, void* workspace, size_t workspace_bytes,
                          void* stream);
/* fp32 -> bf16 (round to nearest even), n a multiple of 8: the 16-bit mirror of the projection buffer the two entry points above
 * gather their k' | v' rows from when kv16 is non-null (the step's reduced-precision "bf16" mode: half the bytes per edge; q, the
 * softmax statistics, every sum and every gradient stay fp32).  kv16 null = rows read from kv itself. */
int mdg_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);

/* ------------------------------------------------------- gathered head (finetune step) ---- */
/* ---- index plumbing of the gathered head's plan (madrigal_amd.ops.triple_plan; the reference feeds a new batch of labelled
 * triples to every step, train_ddi_batch.py:231-354, so the plan is rebuilt per step).  All index arrays are int64 on the device.
 *
 * mdg_plan_gather: for the label-sorted order perm: hs[i] = heads[perm[i]], ts[i] = tails[perm[i]], skey[i] = labels[perm[i]] * n_head
 *   + hs[i], inv_perm[perm[i]] = i; *status |= 1 (a label outside [0, n_labels)) | 2 (a head / tail outside its table).
 * mdg_plan_lower_bounds: out[b] = first i with vals[i] >= b * scale, b = 0 .. n_bounds - 1 (vals ascending, val_bytes = 2 / 4 / 8:
 *   int16 / int32 / int64): CSR pointers of a sorted index list.
 * mdg_plan_cut_count / _fill: the lists of a CSR pointer ptr[n + 1] cut into pieces of <= size entries: first[i] = pieces of the lists
 *   before i (first[n] = totals[0] = all pieces), totals[1] = the longest list; which[j] = the list of piece j (which may be NULL),
 *   start[j] = its first entry, start[total] = ptr[n].  (tiles of 32 and chunks of 256 / 512 triples or pairs of one label, pieces of 64 rows of one drug)
 * mdg_plan_pair_flags: flag[i] = 1 where skey[i] != skey[i - 1] (flag[0] = 0): its inclusive prefix sum is the (label, head) pair of
 *   every sorted triple.  mdg_plan_pair_table: pair_ptr[p] = first triple of pair p (pair_ptr[P] = T), pair_drug[p] = its head drug.
 * mdg_plan_take: out[i] = src[idx[i]] where 0 <= idx[i] < T, else fill. */
int mdg_plan_gather(const int64_t* perm, const int64_t* labels, const int64_t* heads, const int64_t* tails, int64_t T, int64_t n_labels,
                    int64_t n_head, int64_t n_tail, int64_t* hs, int64_t* ts, int64_t* skey, int64_t* inv_perm, int32_t* status, void* stream);
int mdg_plan_lower_bounds(const void* vals, int val_bytes, int64_t T, int64_t scale, int64_t n_bounds, int64_t* out, void* stream);
int mdg_plan_cut_count(const int64_t* ptr, int64_t n, int64_t size, int64_t* first, int64_t* totals, void* stream);
int mdg_plan_cut_fill(const int64_t* ptr, const int64_t* first, int64_t n, int64_t size, int64_t total, int64_t* which, int64_t* start,
                      void* stream);
int mdg_plan_pair_flags(const int64_t* skey, int64_t T, int64_t* flag, void* stream);
int mdg_plan_pair_table(const int64_t* skey, const int64_t* pair_of, int64_t T, int64_t n_head, int64_t P, int64_t* pair_ptr,
                        int64_t* pair_drug, void* stream);
int mdg_plan_take(const int64_t* src, const int64_t* idx, int64_t n, int64_t T, int64_t fill, int64_t* out, void* stream);

/* train_ddi_batch.py:285-288 computes sigmoid(model(...)) [L,N,N] and reads T (label, head, tail) entries of it.  These
 * entry points compute only those entries: score[t] = z_head[head[t]]^T w[label] z_tail[tail[t]].
 * The triples are sorted by label by the caller and cut into tiles of <= 32 triples of ONE label: tile_start [n_tiles+1]
 * (triple offsets), tile_label [n_tiles]; head / tail / score / dscore are in that sorted order.  D must be 128. */
int mdg_bilinear_gather(const float* z_head, const float* z_tail, const float* w, const int64_t* head, const int64_t* tail,
                        const int64_t* tile_start, const int64_t* tile_label, int64_t n_tiles, float* score, int64_t D,
                        void* stream);
/* Backward: gz_head_rows / gz_tail_rows [T,128] receive one gradient row per triple (dscore[t] * w z_tail, dscore[t] *
 * w^T z_head; the caller sums them per drug with mdg_csr_aggregate — no atomics); w_t is w transposed per label (pass
 * w itself when it is symmetric).  dw [n_labels,128,128] (optional) is accumulated through dw_partial
 * [n_chunks,128,128]: chunk_start [n_chunks+1] cuts the sorted triples into chunks of <= 256 triples of one label and
 * label_chunk_ptr [n_labels+1] lists each label's chunks. */
int mdg_bilinear_gather_bwd(const float* z_head, const float* z_tail, const float* w, const float* w_t, const int64_t* head,
                            const int64_t* tail, const int64_t* tile_start, const int64_t* tile_label, int64_t n_tiles,
                            const int64_t* chunk_start, int64_t n_chunks, const int64_t* label_chunk_ptr, int64_t n_labels,
                            const float* dscore, float* gz_head_rows, float* gz_tail_rows, float* dw_partial, float* dw, int64_t D,
                            void* stream);
/* The same with the arithmetic mode of the step: in the 16-bit modes dW runs on the split-bf16 matrix cores (three products of the
 * hi / lo halves of ds * z_head and z_tail, fp32 accumulation: fp32-grade, ~4e-6 of max) as a TN product over the gathered rows;
 * MDG_PREC_F32 = the exact fp32 MFMA kernel of mdg_bilinear_gather_bwd.  The per-triple gradient rows are exact fp32 either way. */
int mdg_bilinear_gather_bwd_prec(const float* z_head, const float* z_tail, const float* w, const float* w_t, const int64_t* head,
                                 const int64_t* tail, const int64_t* tile_start, const int64_t* tile_label, int64_t n_tiles,
                                 const int64_t* chunk_start, int64_t n_chunks, const int64_t* label_chunk_ptr, int64_t n_labels,
                                 const float* dscore, float* gz_head_rows, float* gz_tail_rows, float* dw_partial, float* dw,
                                 int64_t D, int precision, void* stream);
/* Pair-compressed gathered head (same sums as mdg_bilinear_gather / _bwd, grouped per (label, drug) PAIR before the 128 x 128
 * products: the finetune batch holds more labelled triples than (outcome, drug) pairs, train_ddi_batch.py:285-288).
 *   mdg_bilinear_matvec_rows: rows_out[p] = W[tile_label] z[row_index[p]] for pairs cut into tiles of <= 32 pairs of one label
 *       (row_index NULL = identity); exact-fp32 matrix cores.
 *   mdg_gather_rowdot: out[t] = a[ia[t]] . b[ib[t]] over rows of 128 floats (the per-triple score from the per-pair rows).
 * mdg_bilinear_gather_bwd with n_tiles = 0 runs only its dW part; there `tail` NULL means row t of z_tail and `dscore` NULL
 * means weight 1 (the dW sum over pairs). */
int mdg_bilinear_matvec_rows(const float* z, const float* w, const int64_t* row_index, const int64_t* tile_start, const int64_t* tile_label,
                             int64_t n_tiles, float* rows_out, int64_t D, void* stream);
/* The per-pair products in the step's arithmetic mode: MDG_PREC_F32 = the call above; the 16-bit modes run them on the split-bf16
 * matrix cores (z rows split in registers, W from hi / lo bf16 images made inside the call in `workspace`: three products, fp32
 * accumulation -- fp32-grade, ~4e-6 of max).  w [n_labels,128,128]. */
size_t mdg_bilinear_matvec_rows_workspace_bytes(int64_t n_labels, int precision);
int mdg_bilinear_matvec_rows_prec(const float* z, const float* w, int64_t n_labels, const int64_t* row_index, const int64_t* tile_start,
                                  const int64_t* tile_label, int64_t n_tiles, float* rows_out, int64_t D, int precision, void* workspace,
                                  size_t workspace_bytes, void* stream);
int mdg_gather_rowdot(const float* a, const int64_t* ia, const float* b, const int64_t* ib, float* out, int64_t n, int64_t D, void* stream);

/* nn.BCELoss(sigmoid(score), target) per element (log clamp at -100) and/or its gradient w.r.t. the logit times
 * grad_scale (1/T for reduction='mean').  madrigal/utils.py:616-619. */
int mdg_bce_logits(const float* score, const float* target, float* term, float* dscore, int64_t n, float grad_scale, void* stream);
/* Backward of mdg_symmetrize: dw_original = triu(dw_sym) + triu(dw_sym^T, 1). */
int mdg_symmetrize_bwd(const float* dw_sym, float* dw_original, int64_t n_labels, int64_t D, void* stream);

/* ------------------------------------------------------- optimizer ---- */
/* torch.optim.AdamW semantics (madrigal/utils.py:600-613) for every parameter tensor in one launch.  The tensors are cut
 * into chunks of mdg_adamw_chunk_elems() elements: chunk_ptrs [n_chunks,4] = device addresses of the chunk inside
 * (param, grad, exp_avg, exp_avg_sq), chunk_lens [n_chunks], chunk_tensor [n_chunks] = row of hyper [n_tensors,8] =
 * {lr, beta1, beta2, eps, weight_decay, 1/(1-beta1^t), 1/sqrt(1-beta2^t), 0}.  All tables live in device memory. */
int mdg_adamw_chunk_elems(void);
int mdg_adamw_multi(const int64_t* chunk_ptrs, const int32_t* chunk_lens, const int32_t* chunk_tensor, const float* hyper,
                    int64_t n_chunks, void* stream);

/* ------------------------------------------------------- backward-pass building blocks ---- */
/* (the finetune step of train_ddi_batch.py:285-354: loss.backward() through the modules above) */

/* Weight (and bias) gradient of y = x W^T + b:  dw[n,k] = sum_m g[m,n] x[m,k],  dbias[n] = sum_m g[m,n] (optional, NULL to
 * skip)  (g [M,N] = dL/dy, x [M,K] the layer input, both row-major with unit inner stride).  The reduction over the M rows
 * is split across the grid (exact fp32 MFMA, partials summed in a fixed order); workspace from
 * mdg_grad_weight_workspace_bytes. */
size_t mdg_grad_weight_workspace_bytes(int64_t M, int64_t N, int64_t K);
int mdg_grad_weight(const float* g, int64_t ldg, const float* x, int64_t ldx, float* dw, float* dbias, int64_t M, int64_t N,
                    int64_t K, void* workspace, size_t workspace_bytes, void* stream);
/* The same in the arithmetic mode of the step's dense blocks.  MDG_PREC_F32: the exact kernel above.  MDG_PREC_BF16 / BF16X3 (N, K, ldg,
 * ldx multiples of 4, 16-byte aligned operands; anything else takes the exact kernel): g and x rounded to bf16 (bf16x3: split hi + lo,
 * three products) while they are staged, fp32 accumulation on v_mfma_f32_16x16x32_bf16 -- what mdg_linear_tn does for large outputs,
 * here for the small ones whose cost is reading g and x once.  dbias is summed from the fp32 g in every mode. */
int mdg_grad_weight_prec(const float* g, int64_t ldg, const float* x, int64_t ldx, float* dw, float* dbias, int64_t M, int64_t N,
                         int64_t K, int precision, void* workspace, size_t workspace_bytes, void* stream);

/* Grouped dense block: G independent products y_g = alpha_g * act(x_g W_g^T + b_g) + beta_g * r_g sharing K, in ONE launch.
 * Replaces the per-node-type linear layers of PyG HGTConv (kqv_lin / out_lin, a HeteroDictLinear each: one GEMM per node
 * type; call site madrigal/models/models.py:76-79,90-94).  The x rows of all groups are stacked in one [rows_total, K] matrix,
 * the W rows (and biases) of all groups in one [w_rows_total, K] matrix (packed with mdg_pack_operand, or raw fp32 where the
 * mode needs no image); `tiles` is a DEVICE table of n_tiles descriptors of mdg_linear_group_tile_words() int64 words, one per
 * 128 x 128 output tile:
 *   0 a_row0  1 a_row_end  2 b_row0  3 b_row_end   tile origin / end of the group's rows in the stacked x and W
 *   4 m_base  5 n_base                             the group's first x row / first W row
 *   6 y_off   7 ldy                                output element (m, n) -> y[y_off + (m - m_base) * ldy + (n - n_base)]
 *   8 res_off (-1: none)  9 ldr                    residual element, same addressing relative to `residual`
 *   10 alpha (float bits) | beta (float bits) << 32          11 unused
 * Offsets and row strides must be multiples of 4 floats.  Workspace (the packed x): mdg_linear_grouped_workspace_bytes. */
int mdg_linear_group_tile_words(void);
size_t mdg_linear_grouped_workspace_bytes(int64_t rows_total, int64_t K, int precision);
int mdg_linear_grouped(const float* x, int64_t ldx, int64_t rows_total, int64_t K, const float* w, int64_t ldw, const void* w_packed,
                       int64_t w_rows_total, const float* bias, const int64_t* tiles, int64_t n_tiles, float* y,
                       const float* residual, int act, int precision, void* workspace, size_t workspace_bytes, void* stream);

/* y[N,K] = g^T x for row-major g [M,N] (row stride ldg) and x [M,K]: the weight gradient dW = dY^T X of a wide layer
 * (autograd of nn.Linear).  Both operands are re-laid out (reduction index M innermost, padded to 64) by one transposing pack
 * launch, then the mdg_linear tile kernel runs: no separate transposes of g and x. */
size_t mdg_linear_tn_workspace_bytes(int64_t M, int64_t N, int64_t K, int precision);
int mdg_linear_tn(const float* g, int64_t ldg, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                  int precision, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of a wide dense block y = x W^T + b from ONE pass over g = dL/dy [M,N] (16-bit operand modes): the pass writes the operand
 * image of g (for dx = g W through mdg_linear_packed_x with K = N; NULL to skip), the image of g^T (for dW through
 * mdg_linear_tn_packed_g) and dbias = column sums of the fp32 g (NULL to skip).  The separate calls (mdg_linear on g, mdg_linear_tn,
 * mdg_colsum) read g three times.  mdg_linear_backward_pack_bytes(M, N, precision, which): 0 = bytes of the row image, 1 = of the
 * transposed image, 2 = of the workspace (partial column sums). */
size_t mdg_linear_backward_pack_bytes(int64_t M, int64_t N, int precision, int which);
int mdg_linear_backward_pack(const float* g, int64_t ldg, int64_t M, int64_t N, int precision, void* row_image, void* t_image,
                             float* dbias, float drop_p, uint64_t drop_seed, void* workspace, size_t workspace_bytes, void* stream);
/* (drop_p > 0: g is first passed through the backward of the dropout mdg_linear_dropout applied with the same p and seed --
 * element (m, n) kept iff the counter-based hash of (seed, m * N + n) says so, scaled by 1 / (1 - p) -- on load, no pass of its own.) */
size_t mdg_linear_tn_packed_g_workspace_bytes(int64_t M, int64_t N, int64_t K, int precision);
int mdg_linear_tn_packed_g(const void* gt_image, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                           int precision, void* workspace, size_t workspace_bytes, void* stream);

/* out[c, r] = in[r, c]   (dW = dY^T X and dX = dY W are mdg_linear calls on transposed operands) */
int mdg_transpose(const float* in, int64_t ldi, float* out, int64_t ldo, int64_t rows, int64_t cols, void* stream);

/* out[c] = beta * out[c] + sum_r x[r, c], fixed summation order (bias / LayerNorm / learned-token gradients). */
size_t mdg_colsum_workspace_bytes(int64_t rows, int64_t cols);
int mdg_colsum(const float* x, int64_t ldx, float* out, int64_t rows, int64_t cols, float beta, void* workspace,
               size_t workspace_bytes, void* stream);

/* y = act(pre) and dx = dy * act'(pre), elementwise over n contiguous floats (training keeps the pre-activation). */
int mdg_activation_fwd(const float* pre, float* y, int64_t n, int activation, void* stream);
int mdg_activation_bwd(const float* dy, const float* pre, float* dx, int64_t n, int activation, void* stream);
/* y = dropout(act(pre), p, seed) and its backward dx = act'(pre) * dropout-backward(dy) in one pass each (the FFN's
 * dropout(activation(linear1(x))) of nn.TransformerEncoderLayer in training mode); same mask as mdg_dropout(seed). */
int mdg_activation_dropout_fwd(const float* pre, float* y, int64_t n, int activation, float p, uint64_t seed, void* stream);
int mdg_activation_dropout_bwd(const float* dy, const float* pre, float* dx, int64_t n, int activation, float p, uint64_t seed, void* stream);

/* out[i] = alpha * a[i] + beta * b[i mod nb] (residual adds of the training path; nb < n broadcasts a row over rows). */
int mdg_axpby(const float* a, const float* b, float* out, int64_t n, int64_t nb, float alpha, float beta, void* stream);

/* out[i] = x[i] * scalar[0], scalar in device memory (scaling a stored gradient by the incoming scalar gradient). */
int mdg_mul_device_scalar(const float* x, const float* scalar, float* out, int64_t n, void* stream);

/* HGTConv skip gate (PyG 2.3.1 HGTConv.forward): out = s * o + (1 - s) * x with s = sigmoid(skip[0]) read from device
 * memory; backward: d_o, d_x and rowdot[v] = s (1-s) sum_c dout (o - x) whose sum over rows is d skip. */
int mdg_gated_residual(const float* o, const float* x, const float* skip, float* out, int64_t n, void* stream);
int mdg_gated_residual_bwd(const float* dout, const float* o, const float* x, const float* skip, float* d_o, float* d_x,
                           float* rowdot, int64_t rows, int64_t cols, void* stream);

/* Inverted dropout y = x * keep / (1-p); keep is a counter-based hash of (seed, element index), so calling it again
 * with the same seed on dy is the backward pass (nn.Dropout of the transformer / MLPs / position encoder). */
int mdg_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);

/* nn.BatchNorm1d in training mode over the rows of x [rows, cols] (torchdrug MultiLayerPerceptron batch_norm,
 * chemCPA MLP), fused with the following activation: y = act((x - mean) * rstd * gamma + beta); running statistics are
 * updated in place (momentum, unbiased variance).  stats [5*cols] receives mean | rstd | scale | shift for the backward, and the biased batch variance. */
size_t mdg_batchnorm_workspace_bytes(int64_t rows, int64_t cols);
int mdg_batchnorm_train_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float* y, int64_t ldy, float* stats, int64_t rows, int64_t cols, float eps,
                            float momentum, int activation, void* workspace, size_t workspace_bytes, void* stream);
/* dy is the gradient at the BN output (before the activation); x, dy, dx contiguous [rows, cols]. */
int mdg_batchnorm_train_bwd(const float* dy, const float* x, const float* stats, float* dx, float* dgamma, float* dbeta,
                            int64_t rows, int64_t cols, void* workspace, size_t workspace_bytes, void* stream);

/* One more momentum update of running_mean / running_var with the batch statistics of an earlier mdg_batchnorm_train_fwd
 * (stats[0:C] mean, stats[4C:5C] biased variance): nn.BatchNorm1d seeing the same rows again.  The reference encodes head side and tail
 * side of one step separately (madrigal/models/models.py:945-946); when both sides are the same batch, the encoders without
 * dropout (GIN, chemCPA) produce the same output twice -- here they run once and their BatchNorm layers replay the update. */
int mdg_batchnorm_replay_update(const float* stats, float* running_mean, float* running_var, int64_t rows, int64_t cols, float eps,
                                float momentum, void* stream);

/* The phases of mdg_batchnorm_train_fwd / _bwd as separate calls, for SyncBatchNorm over drug-sharded ranks: the caller
 * all-reduces the per-column sums between phases (count = rows over all ranks).
 *   mdg_col_reduce mode 0: out[c] = sum_r x;  1: sum_r (x - center[c])^2;  2: sum_r x * (y - center[c]) * rstd[c]
 *   mdg_batchnorm_finalize phase 0: stats[0:C] = sum / count;  phase 1: rstd | scale | shift | variance (stats has 5C entries) from sqsum / count and the
 *   running-statistics update;  then mdg_affine_act(x, stats + 2C, stats + 3C) applies the normalisation.
 *   mdg_batchnorm_bwd_apply: dx from the (all-reduced) sum_dy and sum_dy_xhat.
 * count_dev (optional): the total row count as a device double (itself all-reduced) read by the kernel instead of the
 * host value — no host round trip inside the step. */
int mdg_col_reduce(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* center, const float* rstd, float* out,
                   int64_t rows, int64_t cols, int mode, void* workspace, size_t workspace_bytes, void* stream);
int mdg_batchnorm_finalize(const float* sum, const float* sqsum, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* stats, double count, const double* count_dev, int64_t cols, float eps,
                           float momentum, int phase, void* stream);
int mdg_batchnorm_bwd_apply(const float* dy, const float* x, const float* stats, const float* sum_dy, const float* sum_dy_xhat,
                            float* dx, int64_t rows, int64_t cols, double count, const double* count_dev, void* stream);

/* y = act(x * scale[c] + shift[c]) per column (eval-mode BatchNorm inside a differentiated graph; shift may be NULL). */
int mdg_affine_act(const float* x, int64_t ldx, const float* scale, const float* shift, float* y, int64_t ldy, int64_t rows,
                   int64_t cols, int activation, void* stream);

/* Training-mode variants of the fusion kernels (nn.MultiheadAttention(dropout=p) drops attention WEIGHTS,
 * models.py:363-379): same arguments as mdg_fusion_attention / mdg_xattn_pool plus the dropout probability and the
 * seed of the counter-based mask; the backward entry points regenerate the mask from the same seed and recompute
 * the attention weights from qkv (nothing else is kept from the forward pass).
 * mdg_fusion_attention_bwd: dqkv [rows, 3d] receives dQ | dK | dV for every row of every tile.
 * mdg_xattn_pool_bwd: dkv [n*Tk, 2d] receives dK | dV; dq_part [n, d] the per-drug gradient of the projected query
 * (the caller sums it over drugs with mdg_colsum). */
int mdg_fusion_attention_dropout(const float* qkv, int64_t ld, float* out, int64_t ldo, const uint32_t* kpm_bits,
                                 const uint32_t* src_bits, float* probs, const int64_t* row_start, const uint32_t* row_bits,
                                 int64_t n, int S, int H, int dh, float p_drop, uint64_t seed, void* stream);
int mdg_fusion_attention_bwd(const float* qkv, int64_t ld, const float* dout, int64_t lddo, float* dqkv, int64_t lddq,
                             const uint32_t* kpm_bits, const uint32_t* src_bits, const int64_t* row_start,
                             const uint32_t* row_bits, int64_t n, int S, int H, int dh, float p_drop, uint64_t seed, void* stream);
int mdg_xattn_pool_dropout(const float* q_proj, const float* kv_proj, int64_t ld, float* out, int64_t ldo, int64_t n, int Tk,
                           int H, int dh, float p_drop, uint64_t seed, void* stream);
int mdg_xattn_pool_bwd(const float* q_proj, const float* kv_proj, int64_t ld, const float* dout, int64_t lddo, float* dkv,
                       int64_t lddkv, float* dq_part, int64_t n, int Tk, int H, int dh, float p_drop, uint64_t seed, void* stream);

/* Backward of mdg_assemble_tokens (rows == NULL): dstr/dkg/dcv [n,128] and dtx [16*n,128] must be zero-filled (tokens
 * that were not emitted leave no gradient); dlearned / dpe are zero-filled [n, S, 128] scratch receiving the per-drug
 * gradients of the cls / bottleneck tokens and of the position table (sum over drugs with mdg_colsum). */
int mdg_assemble_tokens_bwd(const float* dseq, const float* str_emb, const float* kg_emb, const float* cv_emb, const float* tx_emb,
                            const float* bottleneck, const float* cls, const int64_t* token_index, int64_t n_tok, float* dstr,
                            float* dkg, float* dcv, float* dtx, float* dlearned, float* dpe, int64_t n, int nb, int has_cls,
                            int pe_len, int normalize, int64_t D, void* stream);

/* dx of y = x / max(|x|_2, 1e-12) (F.normalize). */
int mdg_l2_normalize_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx, int64_t lddx, int64_t rows,
                         int64_t d, void* stream);

/* LayerNorm backward (statistics recomputed from x): dx, dgamma, dbeta.  d <= 2048. */
size_t mdg_layernorm_bwd_workspace_bytes(int64_t rows, int64_t d);
int mdg_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, float* dx, int64_t lddx,
                      float* dgamma, float* dbeta, int64_t rows, int64_t d, float eps, void* workspace, size_t workspace_bytes,
                      void* stream);
/* dx = LayerNorm-backward(dy) + extra: x feeds the norm AND a residual connection (pre-norm transformer blocks), `extra` is the
 * gradient that arrives through the residual; added while dx is written instead of by a pass of its own. */
int mdg_layernorm_bwd_add(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* extra,
                          int64_t ldextra, float* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int64_t d, float eps,
                          void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MADRIGAL_HIP_H */
