// bbme_rccl.cpp -- libbbme_rccl.so: the gather of a sequence's results over RCCL (include/bbme_rccl.h).  Host C++ on top
// of the public C-ABI of libbbme.so; no torch, no kernels of its own.
#include "bbme_rccl.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>

namespace {
thread_local char g_err[256];
int fail_rccl(const char *what, ncclResult_t r)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, ncclGetErrorString(r));
    fprintf(stderr, "bbme_rccl: %s\n", g_err);
    return BBME_ERR_HIP;
}
}  // namespace

extern "C" {

int bbme_gather_cells(bbme_ctx *ctx, void *nccl_comm, int root, int32_t *d_recv)
{
    if (!ctx || !nccl_comm) return BBME_ERR_INVALID;
    int pw = 0, ph = 0, pairs = 1;
    if (int rc = bbme_batch_size(ctx, &pairs)) return rc;
    if (pairs > 1) {                              // one cell grid per rank: a batched context would silently ship pair 0 only
        snprintf(g_err, sizeof g_err, "bbme_gather_cells: not available on a batched context (%d pairs)", pairs);
        fprintf(stderr, "bbme_rccl: %s\n", g_err);
        return BBME_ERR_UNSUPPORTED;
    }
    if (int rc = bbme_get_geometry(ctx, &pw, &ph, nullptr, nullptr)) return rc;
    const int16_t *cells = nullptr;
    if (int rc = bbme_cells_device(ctx, &cells)) return rc;
    void *stream = nullptr;
    if (int rc = bbme_get_stream(ctx, &stream)) return rc;
    const size_t words = (size_t)(pw / 2) * (ph / 2);
    const ncclResult_t r = ncclGather(cells, d_recv, words, ncclInt32, root, static_cast<ncclComm_t>(nccl_comm),
                                      static_cast<hipStream_t>(stream));                      // rccl.h:745
    return r == ncclSuccess ? BBME_OK : fail_rccl("ncclGather", r);
}

int bbme_expand_gathered(bbme_ctx *ctx, const int32_t *d_recv, int rank, float *d_flow)
{
    if (!ctx || !d_recv || !d_flow || rank < 0) return BBME_ERR_INVALID;
    int pw = 0, ph = 0;
    if (int rc = bbme_get_geometry(ctx, &pw, &ph, nullptr, nullptr)) return rc;
    const size_t words = (size_t)(pw / 2) * (ph / 2);
    return bbme_expand_cells_device(ctx, reinterpret_cast<const int16_t *>(d_recv + (size_t)rank * words), d_flow);
}

}  // extern "C"
