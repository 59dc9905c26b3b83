/*
 * bbme_rccl.h -- C-ABI of libbbme_rccl.so: the multi-GPU step of a sequence without torch.
 *
 * The reference has no multi-GPU code: frame pairs are independent (an MF object holds all state of one
 * pair, motion_framework.h:37-46), so pair p runs on GPU p mod N with no exchange, and the one collective is the gather
 * of the finished fields on the root.  This library is that gather for callers that do not use torch.distributed: it
 * sits on the public C-ABI of libbbme.so (include/bbme.h) and on RCCL's ncclGather (rccl/rccl.h).  The communicator is
 * the caller's -- one rank per process (ncclCommInitRank) or all ranks of one process (ncclCommInitAll) alike.
 */
#ifndef BBME_RCCL_H
#define BBME_RCCL_H

#include "bbme.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Enqueues, on the context's stream (behind its bbme_estimate), ncclGather of the context's result as its compact cell
 * grid -- one packed int16 (dx, dy) pair per 2x2 cell of level 0, moved as int32 words (NCCL has no int16), 16x smaller
 * than the dense field and the same information -- to rank `root` of `nccl_comm` (an ncclComm_t).  d_recv: on the root,
 * device memory for world_size * (padded_h / 2) * (padded_w / 2) words, rank r's grid at offset r; ignored elsewhere.
 * With several ranks in one process, bracket the calls of all ranks with ncclGroupStart / ncclGroupEnd. */
int bbme_gather_cells(bbme_ctx *ctx, void *nccl_comm, int root, int32_t *d_recv);

/* Root side: expands gathered grid `rank` of d_recv to the dense padded field (copy_to_all_pixels,
 * motion_framework.cpp:815-826) in d_flow, on the context's stream (bbme_expand_cells_device). */
int bbme_expand_gathered(bbme_ctx *ctx, const int32_t *d_recv, int rank, float *d_flow);

#ifdef __cplusplus
}
#endif
#endif /* BBME_RCCL_H */
