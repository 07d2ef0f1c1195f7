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
 *   - re-entrant, no global mutable state, no internal threads; safe under hipGraph capture.
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
  MDG_PREC_BF16 = 2    /* operands rounded to bf16 once, fp32 accumulate (cfg "bf16")          */
};

/* What the all-pairs head does with each score tile. */
enum mdg_bilinear_epilogue {
  MDG_EPI_STORE = 0,         /* out[l,i,j] = S            (raw logits, as the reference returns) */
  MDG_EPI_STORE_SIGMOID = 1, /* out[l,i,j] = sigmoid(S)   (train_ddi_batch.py:285)               */
  MDG_EPI_ROWSTATS = 2       /* nothing is materialised: stats[l,i,0] = sum_j S, stats[l,i,1] =
                                max_j S (roofline stress runs whose [L,N,N] cannot exist)        */
};

const char* mdg_last_error(void);
/* "gfx950" build tag, and the ABI version (bumped on any signature change). */
const char* mdg_build_arch(void);
int mdg_abi_version(void);

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

#ifdef __cplusplus
}
#endif
#endif /* MADRIGAL_HIP_H */
