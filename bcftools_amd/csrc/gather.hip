// gather.hip -- what follows the call stage when a region is sharded over several GPUs: the records that
// will be written are compacted on the device (call record, mpileup-stage site record, GT and trimmed PL planes of the
// site, packed back to back in site order), and the shards' buffers are gathered to rank 0 over RCCL (xGMI):
// ncclGroupStart; rank 0 posts one ncclRecv per peer, every peer one ncclSend; ncclGroupEnd.  Rank order is genomic order
// (contiguous region shards, the reference's `-r` regions + `bcftools concat`, mpileup.c:652-683, vcfconcat.c:420), so
// the concatenation is the VCF order and the host only walks it.
//
// RCCL is loaded on first use (dlopen): a single-GPU caller, or a Python process that already carries torch's copy,
// never maps a second one through this library.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "kernels.h"

using namespace bcfgpu;

extern "C" void *bcfgpu_internal_ws(bcfgpu_ctx *c, int slot, size_t bytes);
extern "C" int bcfgpu_internal_device(bcfgpu_ctx *c, hipStream_t *stream, const float **q2p);
extern "C" const bcfgpu_cfg *bcfgpu_internal_cfg(const bcfgpu_ctx *c);
extern "C" void *bcfgpu_internal_pinned(bcfgpu_ctx *c, int slot, size_t bytes);
int bcfgpu_set_error(int code, const char *what);

namespace bcfgpu {

static_assert(sizeof(bcfgpu_call_rec) % 16 == 0, "records are packed at 16-byte steps");

__device__ __forceinline__ uint32_t rec_bytes(const bcfgpu_call_site &cs, int n_smpl, int variants_only)
{
    if (cs.ret < 0 || (variants_only && cs.ret == 0)) return 0;                 // vcfcall.c:1140-1144: not written
    if (variants_only == 2 && cs.nals_new < 2) return 0;                        // no ALT allele called: what `call -v` leaves out
    const int nn = cs.nals_new, ng = cs.pl_dropped ? 0 : nn * (nn + 1) / 2;
    const uint32_t b = (uint32_t)sizeof(bcfgpu_call_rec) + ((2u * n_smpl + 3u) & ~3u) + 4u * ng * n_smpl;
    return (b + 15u) & ~15u;
}

__global__ __launch_bounds__(256) void compact_size_kernel(const bcfgpu_call_site *cs, int n_sites, int n_smpl, int variants_only,
                                                           unsigned long long *size, unsigned long long *counts)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    uint32_t b = 0;
    if (k < n_sites) b = rec_bytes(cs[k], n_smpl, variants_only);
    if (k <= n_sites) size[k] = b;
    // the number of records: one add per wavefront
    const unsigned long long m = __builtin_amdgcn_ballot_w64(b != 0);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(&counts[1], (unsigned long long)__popcll(m));
}

// one workgroup per site: header, then the GT planes, then the kept PL planes.  counts: [0] bytes of all records, [1] records
// (compact_size_kernel), [2] set when the records do not fit cap_bytes (nothing is written then)
__global__ __launch_bounds__(256) void compact_copy_kernel(const bcfgpu_call_site *cs, const bcfgpu_site *ms, const int8_t *gt, const int32_t *pl,
                                                           int n_gt_planes, int n_smpl, int n_sites, int site0, int variants_only,
                                                           const unsigned long long *off, unsigned char *buf, unsigned long long cap_bytes,
                                                           unsigned long long *counts)
{
    const int k = blockIdx.x, tid = threadIdx.x;
    const unsigned long long total = off[n_sites];
    if (k == 0 && tid == 0) { counts[0] = total; counts[2] = total > cap_bytes ? 1ull : 0ull; }
    if (total > cap_bytes) return;
    const uint32_t bytes = rec_bytes(cs[k], n_smpl, variants_only);
    if (!bytes) return;
    unsigned char *dst = buf + off[k];
    const int nn = cs[k].nals_new, ng = cs[k].pl_dropped ? 0 : nn * (nn + 1) / 2;
    if (tid == 0) {
        bcfgpu_call_rec h;
        memset(&h, 0, sizeof h);
        h.site = site0 + k; h.n_gt = ng; h.bytes = bytes;
        h.call = cs[k];
        if (ms) h.mplp = ms[k];
        *reinterpret_cast<bcfgpu_call_rec*>(dst) = h;
    }
    int8_t *g = reinterpret_cast<int8_t*>(dst + sizeof(bcfgpu_call_rec));
    for (int i = tid; i < 2 * n_smpl; i += 256) g[i] = gt[(size_t)k * 2 * n_smpl + i];
    // the PL planes start 4-byte aligned: sizeof(bcfgpu_call_rec) is a multiple of 16 and 2*n_smpl is padded up to 4
    int32_t *p = reinterpret_cast<int32_t*>(dst + sizeof(bcfgpu_call_rec) + ((2 * n_smpl + 3) & ~3));
    for (int i = tid; i < ng * n_smpl; i += 256) p[i] = pl[(size_t)k * n_gt_planes * n_smpl + i];
}

}  // namespace bcfgpu

// Everything is queued on the context's stream; nothing comes back to the host here.
extern "C" int bcfgpu_compact_calls_async(bcfgpu_ctx *ctx, int32_t n_sites, int32_t site0, const bcfgpu_site *msite, const bcfgpu_call_out *cout,
                                          int32_t n_gt_planes, int32_t variants_only, void *d_buf, uint64_t cap_bytes, uint64_t *d_counts)
{
    if (!ctx || !cout || !cout->site || !cout->gt || !cout->pl || !d_counts || n_sites < 0 || n_gt_planes < 1 || (cap_bytes && !d_buf))
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_compact_calls_async: bad arguments");
    hipStream_t st;
    if (bcfgpu_internal_device(ctx, &st, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_compact_calls_async: bad context");
    if (hipMemsetAsync(d_counts, 0, 4 * sizeof(uint64_t), st) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_compact_calls_async: counters");
    if (n_sites == 0) return 0;
    const int S = bcfgpu_internal_cfg(ctx)->n_smpl;
    unsigned long long *d_size = (unsigned long long*)bcfgpu_internal_ws(ctx, 35, ((size_t)n_sites + 1) * 8 + 64);
    if (!d_size) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_compact_calls_async: device workspace");
    hipLaunchKernelGGL(compact_size_kernel, dim3((n_sites + 256) / 256), dim3(256), 0, st, cout->site, n_sites, S, variants_only, d_size,
                       reinterpret_cast<unsigned long long*>(d_counts));
    size_t tmp = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, d_size, d_size, n_sites + 1, st) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "scan");
    void *d_tmp = bcfgpu_internal_ws(ctx, 36, tmp + 64);
    if (!d_tmp) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_compact_calls_async: device workspace");
    if (hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp, d_size, d_size, n_sites + 1, st) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "scan");
    hipLaunchKernelGGL(compact_copy_kernel, dim3(n_sites), dim3(256), 0, st, cout->site, msite, cout->gt, cout->pl, n_gt_planes, S, n_sites, site0,
                       variants_only, d_size, (unsigned char*)d_buf, (unsigned long long)cap_bytes, reinterpret_cast<unsigned long long*>(d_counts));
    if (hipGetLastError() != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_compact_calls_async: launch");
    return 0;
}

extern "C" int bcfgpu_compact_counts(bcfgpu_ctx *ctx, const uint64_t *d_counts, uint64_t *n_bytes, uint32_t *n_rec)
{
    if (!ctx || !d_counts || !n_bytes) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_compact_counts: bad arguments");
    hipStream_t st;
    if (bcfgpu_internal_device(ctx, &st, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_compact_counts: bad context");
    uint64_t *h = (uint64_t*)bcfgpu_internal_pinned(ctx, 1, 4 * sizeof(uint64_t));
    if (!h) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_compact_counts: staging");
    if (hipMemcpyAsync(h, d_counts, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_compact_counts: read");
    *n_bytes = h[0]; if (n_rec) *n_rec = (uint32_t)h[1];
    if (h[2]) return bcfgpu_set_error(BCFGPU_E_RANGE, "bcfgpu_compact_calls: the buffer is too small for the records (see bcfgpu_call_rec)");
    return 0;
}

extern "C" int bcfgpu_compact_calls(bcfgpu_ctx *ctx, int32_t n_sites, int32_t site0, const bcfgpu_site *msite, const bcfgpu_call_out *cout,
                                    int32_t n_gt_planes, int32_t variants_only, void *d_buf, uint64_t cap_bytes,
                                    uint64_t *n_bytes, uint32_t *n_rec)
{
    if (!n_bytes) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_compact_calls: bad arguments");
    *n_bytes = 0; if (n_rec) *n_rec = 0;
    uint64_t *d_counts = (uint64_t*)bcfgpu_internal_ws(ctx, 37, 64);
    if (!d_counts) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_compact_calls: device workspace");
    const int rc = bcfgpu_compact_calls_async(ctx, n_sites, site0, msite, cout, n_gt_planes, variants_only, d_buf, cap_bytes, d_counts);
    if (rc) return rc;
    return bcfgpu_compact_counts(ctx, d_counts, n_bytes, n_rec);
}

// ---- RCCL, loaded on first use --------------------------------------------------------------------------------------
namespace {
typedef void *ncclComm_t;
typedef int ncclResult_t;
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (r.lib) {
            r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
            r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
            r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
            if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv) { dlclose(r.lib); r.lib = nullptr; }
        }
    }
    return r.lib ? &r : nullptr;
}
const int NCCL_INT8 = 0;            // ncclInt8 / ncclChar
}  // namespace

struct bcfgpu_comm { int n = 0; std::vector<ncclComm_t> comm; std::vector<bcfgpu_ctx*> ctx; };

extern "C" int bcfgpu_comm_init_all(bcfgpu_ctx *const *ctxs, int32_t n, bcfgpu_comm **out)
{
    if (!ctxs || n < 1 || !out) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_comm_init_all: bad arguments");
    *out = nullptr;
    std::vector<int> dev(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_comm_init_all: NULL context");
        dev[i] = bcfgpu_internal_cfg(ctxs[i])->device;
        for (int j = 0; j < i; ++j) if (dev[j] == dev[i]) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_comm_init_all: two ranks on one device (RCCL wants one device per rank)");
    }
    bcfgpu_comm *c = new bcfgpu_comm();
    c->n = n; c->ctx.assign(ctxs, ctxs + n); c->comm.assign(n, nullptr);
    if (n > 1) {
        Rccl *R = rccl();
        if (!R) { delete c; return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_comm_init_all: librccl.so could not be loaded"); }
        const ncclResult_t rc = R->CommInitAll(c->comm.data(), n, dev.data());
        if (rc) { delete c; return bcfgpu_set_error(BCFGPU_E_HIP, R->GetErrorString ? R->GetErrorString(rc) : "ncclCommInitAll failed"); }
        // one line, once per communicator: which transport the ordered gather runs on (the first multi-GPU box tells us)
        fprintf(stderr, "[bcfgpu] gather transport: RCCL (ncclCommInitAll over %d devices; grouped ncclSend / ncclRecv to rank 0)\n", n);
    }
    *out = c;
    return 0;
}

extern "C" void bcfgpu_comm_destroy(bcfgpu_comm *c)
{
    if (!c) return;
    Rccl *R = c->n > 1 ? rccl() : nullptr;
    if (R) for (ncclComm_t x : c->comm) if (x) R->CommDestroy(x);
    delete c;
}

// Every rank calls this from its own thread with the same `counts`.  Rank 0 receives the peers' bytes behind its own
// (d_recv must hold the sum of counts; its own share is copied there device to device); the others send d_send.
extern "C" int bcfgpu_gather_bytes(bcfgpu_comm *c, int32_t rank, const void *d_send, const uint64_t *counts, void *d_recv)
{
    if (!c || rank < 0 || rank >= c->n || !counts) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gather_bytes: bad arguments");
    hipStream_t st;
    if (bcfgpu_internal_device(c->ctx[rank], &st, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gather_bytes: bad context");
    if (rank == 0) {
        if (counts[0] && (!d_recv || !d_send)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gather_bytes: NULL buffer");
        if (counts[0] && d_recv != d_send && hipMemcpyAsync(d_recv, d_send, counts[0], hipMemcpyDeviceToDevice, st) != hipSuccess)
            return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_gather_bytes: local copy");
    }
    if (c->n == 1) return 0;
    Rccl *R = rccl();
    if (!R) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_gather_bytes: librccl.so could not be loaded");
    ncclResult_t rc = R->GroupStart();
    if (rank == 0) {
        uint64_t off = counts[0];
        for (int p = 1; p < c->n && !rc; ++p) {
            if (counts[p]) rc = R->Recv((char*)d_recv + off, counts[p], NCCL_INT8, p, c->comm[0], st);
            off += counts[p];
        }
    } else if (counts[rank]) rc = R->Send(d_send, counts[rank], NCCL_INT8, 0, c->comm[rank], st);
    const ncclResult_t rc2 = R->GroupEnd();
    if (rc || rc2) return bcfgpu_set_error(BCFGPU_E_HIP, R->GetErrorString ? R->GetErrorString(rc ? rc : rc2) : "RCCL send/recv failed");
    return 0;
}
