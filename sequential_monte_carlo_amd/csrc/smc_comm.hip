// smc_comm.hip -- theta sharded over the GPUs of one node from a host that has no torch.distributed (the Julia
// wrapper): the few collectives of the samplers inside libsmchip.so, over RCCL (xGMI).
//   smc_outer_reweight      reweight(logZ) of smc_samplers.jl:232,249,265,298,338: all-gather of the ranks' slices,
//                           then normalize() on every rank (identical results everywhere)
//   smc_comm_all_gather     the moved (theta, logZ, accepted) slices after rejuvenate!
//   smc_comm_exchange_slots resample!(smc) of the online sampler (smc_samplers.jl:74-84): whole filters move between
//                           ranks, pack -> grouped ncclSend/ncclRecv (an all-to-all) -> unpack
// RCCL is opened with dlopen at the first smc_comm_* call: libsmchip.so itself links only libamdhip64, loads on a
// box without RCCL, and shares the copy a host process may already have loaded (PyTorch bundles its own).
#include "../../include/smc_hip.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstring>
#include <string>
#include <vector>

extern "C" int smc_set_error_(int code, const char* msg);   // smc_capi.hip: thread-local message behind smc_last_error()

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return SMC_OK;
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return smc_set_error_(SMC_EHIP, "smc_comm: librccl.so not found (dlopen)");
    Rccl r;
    r.lib = lib;
#define SYM(field, name)                                                                      \
    *(void**)(&r.field) = dlsym(lib, name);                                                   \
    if (!r.field) return smc_set_error_(SMC_EHIP, "smc_comm: symbol " name " missing in librccl.so");
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl = r;
    return SMC_OK;
}

}  // namespace

struct smc_comm_s {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    double* d_send = nullptr;   // device staging of the all-gathers
    double* d_recv = nullptr;
    size_t cap = 0;             // doubles per rank the staging holds
    void* d_xs = nullptr;       // packed filter slots going out / coming in (exchange_slots)
    void* d_xr = nullptr;
    size_t xcap = 0;            // bytes
};

#define HIPC(expr)                                                                                   \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) return smc_set_error_(SMC_EHIP, (std::string(#expr) + ": " + hipGetErrorString(_e)).c_str()); \
    } while (0)
#define NCCLC(expr)                                                                                  \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess) return smc_set_error_(SMC_EHIP, (std::string(#expr) + ": " + g_rccl.GetErrorString(_r)).c_str()); \
    } while (0)

extern "C" int smc_comm_unique_id(void* id) {
    static_assert(sizeof(ncclUniqueId) == SMC_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) return smc_set_error_(SMC_EINVAL, "smc_comm_unique_id: NULL argument");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    NCCLC(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return SMC_OK;
}

extern "C" int smc_comm_create(const void* id, int rank, int world, int device, smc_comm* out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return smc_set_error_(SMC_EINVAL, "smc_comm_create: bad argument");
    *out = nullptr;
    int rc = load_rccl();
    if (rc) return rc;
    HIPC(hipSetDevice(device));
    smc_comm_s* c = new smc_comm_s();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return smc_set_error_(SMC_EHIP, (std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)).c_str());
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        g_rccl.CommDestroy(c->comm);
        delete c;
        return smc_set_error_(SMC_EHIP, hipGetErrorString(e));
    }
    *out = c;
    return SMC_OK;
}

extern "C" int smc_comm_destroy(smc_comm c) {
    if (!c) return SMC_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    (void)hipFree(c->d_send); (void)hipFree(c->d_recv); (void)hipFree(c->d_xs); (void)hipFree(c->d_xr);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return SMC_OK;
}

extern "C" int smc_comm_rank(smc_comm c, int* rank, int* world) {
    if (!c) return smc_set_error_(SMC_EINVAL, "smc_comm_rank: NULL communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return SMC_OK;
}

static int ensure_staging(smc_comm c, size_t n) {
    if (n <= c->cap) return SMC_OK;
    HIPC(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_send); (void)hipFree(c->d_recv);
    c->d_send = c->d_recv = nullptr; c->cap = 0;
    HIPC(hipMalloc((void**)&c->d_send, n * 8));
    HIPC(hipMalloc((void**)&c->d_recv, n * 8 * (size_t)c->world));
    c->cap = n;
    return SMC_OK;
}

// every rank contributes n doubles; all [world][n] come back on every rank (rank order)
extern "C" int smc_comm_all_gather(smc_comm c, const double* local, int64_t n, double* all) {
    if (!c || !local || !all || n <= 0) return smc_set_error_(SMC_EINVAL, "smc_comm_all_gather: bad argument");
    HIPC(hipSetDevice(c->device));
    int rc = ensure_staging(c, (size_t)n);
    if (rc) return rc;
    HIPC(hipMemcpyAsync(c->d_send, local, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    NCCLC(g_rccl.AllGather(c->d_send, c->d_recv, (size_t)n, ncclFloat64, c->comm, c->stream));
    HIPC(hipMemcpyAsync(all, c->d_recv, (size_t)n * 8 * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return SMC_OK;
}

// reweight(logw) with the entries of logw sharded over the ranks (smc_samplers.jl:232,249,265,298,338): smc_host_reweight of
// the concatenated vector, bit for bit, whatever the number of ranks (the specification is a sum of integers over fixed
// segments of SMC_OUTER_SEG entries: smc_outer.hip).  Slices made of whole segments and no weight vector asked for: only the
// segment records travel (32 bytes per SMC_OUTER_SEG entries); otherwise the slices themselves.
extern "C" int smc_outer_reweight(smc_comm c, const double* logw_local, int64_t n_local, double* logw_all, double* w_all,
                                  double* logmu, double* ess) {
    if (!c || !logw_local || n_local <= 0) return smc_set_error_(SMC_EINVAL, "smc_outer_reweight: bad argument");
    const int64_t n = n_local * c->world;
    if (!w_all && !logw_all && n_local % SMC_OUTER_SEG == 0) {
        const int64_t ns = n_local / SMC_OUTER_SEG;
        std::vector<uint64_t> mine((size_t)ns * 4), all((size_t)ns * 4 * (size_t)c->world);
        int rc = smc_host_outer_records(logw_local, n_local, mine.data());
        if (rc) return rc;
        // the records travel as 8-byte words (no arithmetic touches them on the way)
        rc = smc_comm_all_gather(c, reinterpret_cast<const double*>(mine.data()), ns * 4, reinterpret_cast<double*>(all.data()));
        if (rc) return rc;
        return smc_host_outer_combine(all.data(), ns * c->world, n, logmu, ess);
    }
    std::vector<double> tmp;
    double* all = logw_all;
    if (!all) { tmp.resize((size_t)n); all = tmp.data(); }
    int rc = smc_comm_all_gather(c, logw_local, n_local, all);
    if (rc) return rc;
    return smc_host_reweight(all, n, w_all, logmu, ess);
}

// Who sends which slot to whom (pure host arithmetic, exported so that it can be tested without GPUs).  What rank `rank`
// sends to rank r: its own slots among the ancestors of r's slots, in r's slot order (send_idx: LOCAL indices, grouped by
// destination, send_cnt[r] of them); what it receives from rank s: one packed slot for every local slot whose ancestor lives
// on s, in local slot order (dest_idx: LOCAL indices grouped by source, recv_cnt[s] of them, M/world in total).
extern "C" int smc_comm_plan_exchange(const int32_t* a, int64_t M, int rank, int world, int32_t* send_idx, int64_t* send_cnt,
                                      int64_t* n_send, int32_t* dest_idx, int64_t* recv_cnt) {
    if (!a || !send_idx || !send_cnt || !n_send || !dest_idx || !recv_cnt || M <= 0 || world < 1 || rank < 0 || rank >= world || M % world)
        return smc_set_error_(SMC_EINVAL, "smc_comm_plan_exchange: bad argument");
    const int64_t per = M / world, lo = rank * per;
    int64_t ns = 0, nr = 0;
    for (int r = 0; r < world; ++r) {
        send_cnt[r] = 0;
        for (int64_t m = r * per; m < (r + 1) * per; ++m)
            if (a[m] / per == rank) { send_idx[ns++] = (int32_t)(a[m] - lo); ++send_cnt[r]; }
    }
    for (int s = 0; s < world; ++s) {
        recv_cnt[s] = 0;
        for (int64_t m = 0; m < per; ++m)
            if (a[lo + m] / per == s) { dest_idx[nr++] = (int32_t)m; ++recv_cnt[s]; }
    }
    *n_send = ns;
    return SMC_OK;
}

// resample!(smc) with the filters of `h` sharded over the ranks: a[m] (m = 0..M-1, GLOBAL indices, the same vector on
// every rank) is the ancestor of global slot m; rank r holds the slots [r M/world, (r+1) M/world).  Value copies.
extern "C" int smc_comm_exchange_slots(smc_comm c, smc_handle h, const int32_t* a, int64_t M) {
    if (!c || !h || !a || M <= 0 || M % c->world) return smc_set_error_(SMC_EINVAL, "smc_comm_exchange_slots: bad argument");
    HIPC(hipSetDevice(c->device));
    const int W = c->world;
    const int64_t per = M / W;
    for (int64_t m = 0; m < M; ++m)
        if (a[m] < 0 || a[m] >= M) return smc_set_error_(SMC_EINVAL, "smc_comm_exchange_slots: ancestor out of range");
    int64_t sb = 0;
    int rc = smc_slot_bytes(h, &sb);
    if (rc) return rc;
    std::vector<int32_t> send_idx((size_t)M), dest_idx((size_t)per);
    std::vector<int64_t> send_cnt((size_t)W, 0), recv_cnt((size_t)W, 0);
    int64_t n_send = 0;
    rc = smc_comm_plan_exchange(a, M, c->rank, W, send_idx.data(), send_cnt.data(), &n_send, dest_idx.data(), recv_cnt.data());
    if (rc) return rc;
    send_idx.resize((size_t)n_send);
    const size_t ns = send_idx.size(), nr = dest_idx.size();   // nr == per
    const size_t need = (ns > nr ? ns : nr) * (size_t)sb;
    if (need > c->xcap) {
        HIPC(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_xs); (void)hipFree(c->d_xr);
        c->d_xs = c->d_xr = nullptr; c->xcap = 0;
        HIPC(hipMalloc(&c->d_xs, need));
        HIPC(hipMalloc(&c->d_xr, need));
        c->xcap = need;
    }
    if (ns && (rc = smc_pack_slots(h, send_idx.data(), (int64_t)ns, c->d_xs))) return rc;   // synchronises the handle's stream
    // every Send / Recv sits between GroupStart and GroupEnd, and GroupEnd runs whatever happened in between: a group left open on
    // this thread would swallow every later collective of the communicator (and the peers would wait for ever).  The first error
    // is reported after the group has been closed; the filter slots of `h` are then undefined (some may have been received).
    ncclResult_t first = g_rccl.GroupStart();
    const char* where = "ncclGroupStart";
    if (first == ncclSuccess) {
        size_t so = 0, ro = 0;
        for (int r = 0; r < W; ++r) {
            const size_t nsb = (size_t)send_cnt[(size_t)r] * (size_t)sb, nrb = (size_t)recv_cnt[(size_t)r] * (size_t)sb;
            if (nsb && first == ncclSuccess) { first = g_rccl.Send((const char*)c->d_xs + so, nsb, ncclUint8, r, c->comm, c->stream); where = "ncclSend"; }
            if (nrb && first == ncclSuccess) { first = g_rccl.Recv((char*)c->d_xr + ro, nrb, ncclUint8, r, c->comm, c->stream); where = "ncclRecv"; }
            so += nsb; ro += nrb;
        }
        const ncclResult_t end = g_rccl.GroupEnd();
        if (first == ncclSuccess && end != ncclSuccess) { first = end; where = "ncclGroupEnd"; }
    }
    if (first != ncclSuccess) {
        (void)hipStreamSynchronize(c->stream);
        return smc_set_error_(SMC_EHIP, (std::string("smc_comm_exchange_slots: ") + where + ": " + g_rccl.GetErrorString(first) +
                                         " (the filter slots of the handle are undefined now)").c_str());
    }
    HIPC(hipStreamSynchronize(c->stream));     // the unpack kernel runs on the handle's own stream
    if (nr && (rc = smc_unpack_slots(h, dest_idx.data(), (int64_t)nr, c->d_xr))) return rc;
    return SMC_OK;
}
