/* libstem_dp.so -- gradient exchange of a data-parallel rank over RCCL, issued natively.
 *
 * Replaces, for this path, what the reference gets from torch.nn.parallel.DistributedDataParallel
 * (stem_roi/train_stem_roi.py:509-631 wraps its models; stem/trainSTEM.py is single-device): sum all-reduces of slices of
 * the flat gradient buffer while backward still runs.  Why a library of its own and not torch.distributed: on this chip a
 * PENDING wait in the communication stream's hardware queue costs the compute streams 2.4 ms per training step,
 * so a collective must not be enqueued before its input is final, and the stream that consumes the result must be released by
 * something that exists before the collective does.  Here
 *   - stem_dp_submit() records events on the streams that produced a slice and hands the slice to a helper thread, which polls
 *     the events on the HOST and only then enqueues ncclAllReduce on the communicator's stream (which therefore never holds an
 *     unsatisfied wait);
 *   - stem_dp_fence() makes a stream wait for a flag (hipStreamWaitValue32 on signal memory) that the helper writes, on the
 *     communicator's stream, behind the last collective submitted so far.
 * Every rank must submit the same slices in the same order.  Plain C ABI, one handle per communicator; no torch types.
 * Python binding: spatiotemporalentropymodel_amd/distributed.py (_NativeIssuer); reference-side use: INTEGRATION.md. */
#ifndef STEM_DP_H
#define STEM_DP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define STEM_DP_ID_BYTES 128

/* Construction is two steps, so that the ranks can AGREE between them (one MIN all-reduce of "prepared fine" over whatever
 * channel carries the id): everything that can fail on one rank alone happens in stem_dp_prepare / stem_dp_unique_id, and either
 * every rank enters the collective stem_dp_connect or none does -- a rank that failed locally must never leave its peers inside
 * ncclCommInitRank. */

/* rank 0, local: a fresh communicator id (ncclGetUniqueId); ship the 128 bytes to the other ranks by any means */
int stem_dp_unique_id(unsigned char *id128);
/* every rank, LOCAL (no communication): HIP device `device`, its wait-value capability (-4 without), the flag in signal memory,
 * the communication stream.  *handle is NULL on failure. */
int stem_dp_prepare(void **handle, int device);
/* every rank, COLLECTIVE: ncclCommInitRank on the prepared handle, checks ncclCommCount == world, starts the helper thread */
int stem_dp_connect(void *handle, const unsigned char *id128, int world, int rank);
/* stem_dp_prepare + stem_dp_connect in one call (single-rank use, tests); a failed connect destroys the handle */
int stem_dp_create(void **handle, const unsigned char *id128, int world, int rank, int device);
/* ranks of the communicator as RCCL itself reports them (ncclCommCount); negative on error */
int stem_dp_nranks(void *handle);
/* buf[0 .. count) (fp32, in place) <- sum over ranks, once everything enqueued so far on streams[0 .. n) has completed */
int stem_dp_submit(void *handle, void *const *streams, int n, float *buf, size_t count);
/* work enqueued on `stream` after this call starts after every exchange submitted so far has completed */
int stem_dp_fence(void *handle, void *stream);
/* 0, or the (negative) status of this rank's first failure (helper thread or caller side).  Failure is abort-all: the rank's
 * communicator is aborted (ncclCommAbort: its queued collectives end, the flag is released from the host so that no stream
 * waits for ever), every later submit / fence returns the status, and the caller must leave with a non-zero exit code -- its
 * launcher then stops the peers, which would otherwise wait inside a collective this rank never joins.  Gradients of the step
 * in which it happened are NOT reduced: check the status after the step's fence and before trusting the replicas. */
int stem_dp_status(void *handle);
/* host-initiated abort-all with status `code` (negative) and message `why`: same state as a helper failure */
int stem_dp_abort(void *handle, int code, const char *why);
int stem_dp_destroy(void *handle);
const char *stem_dp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
