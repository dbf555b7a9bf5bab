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

/* rank 0: a fresh communicator id (ncclGetUniqueId); ship the 128 bytes to the other ranks by any means */
int stem_dp_unique_id(unsigned char *id128);
/* all ranks, collectively: communicator + helper thread + communication stream + flag for HIP device `device` */
int stem_dp_create(void **handle, const unsigned char *id128, int world, int rank, int device);
/* buf[0 .. count) (fp32, in place) <- sum over ranks, once everything enqueued so far on streams[0 .. n) has completed */
int stem_dp_submit(void *handle, void *const *streams, int n, float *buf, size_t count);
/* work enqueued on `stream` after this call starts after every exchange submitted so far has completed */
int stem_dp_fence(void *handle, void *stream);
/* 0, or the (negative) status of the first collective that failed in the helper thread */
int stem_dp_status(void *handle);
int stem_dp_destroy(void *handle);
const char *stem_dp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
