/*
 * aslp_ctc.h -- CTC loss C ABI of libaslp_hip.so (boundary B5 of SURVEY.md §8b).
 *
 * Drop-in for the reference's vendored Warp-CTC interface, src/warp-ctc/include/ctc.h:16-122:
 * same names, argument order, status enum and ctcComputeInfo layout, so
 * src/aslp-nnet/warp-ctc.cc:33-198 (WarpCtc::EvalGpu) links against this library unchanged.
 * Semantics follow the reference CPU implementation (cpu_ctc.h:158-428), which is the parity
 * oracle: blank = 0; softmax is applied internally; `gradients` is w.r.t. the UN-normalised
 * activations and must be zeroed by the caller (rows t >= input_lengths[n] and infeasible
 * utterances are left untouched); an utterance with L + repeats > T gets cost 0.
 *
 * Only info.loc == CTC_GPU is served (activations / gradients / workspace in device memory,
 * labels / lengths / costs in host memory, kernels on info.stream).  CTC_CPU returns
 * CTC_STATUS_EXECUTION_FAILED: this library has no CPU path.
 */
#ifndef ASLP_CTC_H_
#define ASLP_CTC_H_
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct CUstream_st *CUstream; /* = hipStream_t; kept for source compatibility (ctc.h:14) */

typedef enum {
  CTC_STATUS_SUCCESS = 0,
  CTC_STATUS_MEMOPS_FAILED = 1,
  CTC_STATUS_INVALID_VALUE = 2,
  CTC_STATUS_EXECUTION_FAILED = 3,
  CTC_STATUS_UNKNOWN_ERROR = 4
} ctcStatus_t; /* ctc.h:16-22 */

const char *ctcGetStatusString(ctcStatus_t status); /* ctc.h:28 */

typedef enum { CTC_CPU = 0, CTC_GPU = 1 } ctcComputeLocation; /* ctc.h:30-33 */

struct ctcComputeInfo { /* ctc.h:40-46 */
  ctcComputeLocation loc;
  union {
    unsigned int num_threads;
    CUstream stream;
  };
};
#ifndef __cplusplus
typedef struct ctcComputeInfo ctcComputeInfo;
#endif

/* ctc.h:88-97.  activations / gradients: (t, n, p) dense, element (t*minibatch + n)*alphabet_size + p */
ctcStatus_t compute_ctc_loss(const float *const activations, float *gradients, const int *const flat_labels,
                             const int *const label_lengths, const int *const input_lengths, int alphabet_size, int minibatch,
                             float *costs, void *workspace, struct ctcComputeInfo info);

/* ctc.h:116-120 */
ctcStatus_t get_workspace_size(const int *const label_lengths, const int *const input_lengths, int alphabet_size, int minibatch,
                               struct ctcComputeInfo info, size_t *size_bytes);

int get_warpctc_version(void);

/* Extension used by the host engine's WarpCtc loss (replaces the de-stride / malloc / row copy-back
 * sequence of aslp-nnet/warp-ctc.cc:85-147): same computation on row-padded device matrices, row
 * (t*minibatch + n) at acts + (t*minibatch + n)*ld_acts; gradient rows of frames t < input_lengths[n]
 * are written, all others left untouched; workspace is library-owned; runs on the aslp_set_stream() stream. */
ctcStatus_t aslp_ctc_loss_strided(const float *acts, int ld_acts, float *grads, int ld_grads, const int *flat_labels,
                                  const int *label_lengths, const int *input_lengths, int alphabet_size, int minibatch, float *costs);


/* Eesen CTC on POST-softmax outputs (Ctc::EvalParallel, aslp-nnet/ctc-loss.cc:115-227; kernels
 * aslp-cudamatrix/cu-kernels.cu:3276-3534): diff rows of frames t < frame_num[n] receive y - posterior (the
 * unclipped value the reference forms before its loss check), others are left untouched;
 * pzx_host[n] = log p(z|x) (-1e30 = the reference's log_zero_ when there is no alignment). */
ctcStatus_t aslp_eesen_ctc_mseq(const float *net_out, int ld, float *diff, int ld_diff, const int *flat_labels, const int *label_lengths,
                                const int *frame_num, int alphabet_size, int minibatch, float *pzx_host);

#ifdef __cplusplus
}
#endif
#endif
