// Multi-GPU boundary of the chop path in C: one process per GPU, RCCL over xGMI.
//
// The reference is single-process (run.py:167-202); what shards is chop_forward's tile list, and the one exchange step is
// "every rank's raw HR tiles -> rank 0" in front of the blend (utils/utils.py:372-445), plus the broadcast of the blended
// intermediate between the stages of a model chain (run.py:424-426).  These entry points are that exchange for a C / ctypes
// user; innfer_amd/parallel.py drives the same pattern through torch.distributed (whose "nccl" backend is this same RCCL).
//
//   * the partition is innfer_shard_tiles(): an even, contiguous split of the row-major tile list (798 tiles over 8 ranks ->
//     100 x 6, 99 x 2);
//   * innfer_gather_tiles(): ONE group of point-to-point transfers -- rank r > 0 ncclSend()s exactly its share (no padding),
//     rank 0 ncclRecv()s every share straight into its slot of the [n_tiles, tile_bytes] buffer the blend kernel reads.  On
//     the fully connected xGMI topology every peer pushes over its own link into rank 0;
//   * RCCL is bound at run time (dlopen of the librccl.so.1 the process already carries -- PyTorch-ROCm bundles one -- or the
//     system one): libinnfer_amd.so itself has no link-time dependency on it and single-GPU users never load it.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

using namespace innfer;

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.ok) return INNFER_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);            // the copy this process already runs (one RCCL per process)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return set_error(INNFER_ERR_UNSUPPORTED, "comm: librccl.so.1 not found (%s)", dlerror());
    g_rccl.handle = h;
#define BIND(field, sym)                                                                              \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, #sym);                                              \
    if (!g_rccl.field) return set_error(INNFER_ERR_UNSUPPORTED, "comm: librccl has no symbol " #sym)
    BIND(GetUniqueId, ncclGetUniqueId);
    BIND(CommInitRank, ncclCommInitRank);
    BIND(CommDestroy, ncclCommDestroy);
    BIND(GroupStart, ncclGroupStart);
    BIND(GroupEnd, ncclGroupEnd);
    BIND(Send, ncclSend);
    BIND(Recv, ncclRecv);
    BIND(Broadcast, ncclBroadcast);
    BIND(GetErrorString, ncclGetErrorString);
#undef BIND
    g_rccl.ok = true;
    return INNFER_OK;
}

#define INNFER_NCCL(expr)                                                                                \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess)                                                                           \
            return set_error(INNFER_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));         \
    } while (0)

}  // namespace

struct innfer_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
};

extern "C" int innfer_shard_tiles(int n_tiles, int nranks, int rank, int* first, int* count) {
    if (n_tiles < 0 || nranks < 1 || rank < 0 || rank >= nranks || !first || !count)
        return set_error(INNFER_ERR_INVALID, "shard_tiles: n_tiles=%d nranks=%d rank=%d", n_tiles, nranks, rank);
    const int base = n_tiles / nranks, rem = n_tiles % nranks;
    *first = rank * base + (rank < rem ? rank : rem);
    *count = base + (rank < rem ? 1 : 0);
    return INNFER_OK;
}

extern "C" int innfer_comm_unique_id(void* h_id) {
    if (!h_id) return set_error(INNFER_ERR_INVALID, "comm_unique_id: null buffer");
    if (int rc = load_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == INNFER_COMM_ID_BYTES, "ncclUniqueId size");
    INNFER_NCCL(g_rccl.GetUniqueId((ncclUniqueId*)h_id));
    return INNFER_OK;
}

extern "C" int innfer_comm_init(innfer_comm_t* out, const void* h_id, int rank, int nranks) {
    if (!out || !h_id || nranks < 1 || rank < 0 || rank >= nranks)
        return set_error(INNFER_ERR_INVALID, "comm_init: rank=%d nranks=%d", rank, nranks);
    if (int rc = load_rccl()) return rc;
    innfer_comm* c = new innfer_comm();
    c->rank = rank; c->nranks = nranks;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return set_error(INNFER_ERR_HIP, "comm_init: hipGetDevice failed"); }
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return set_error(INNFER_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
    }
    *out = c;
    return INNFER_OK;
}

extern "C" void innfer_comm_destroy(innfer_comm_t c) {
    if (!c) return;
    if (c->comm && g_rccl.ok) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

extern "C" int innfer_comm_rank(innfer_comm_t c) { return c ? c->rank : INNFER_ERR_INVALID; }
extern "C" int innfer_comm_size(innfer_comm_t c) { return c ? c->nranks : INNFER_ERR_INVALID; }

extern "C" int innfer_gather_tiles(innfer_comm_t c, void* d_tiles, size_t tile_bytes, int n_tiles, void* stream) {
    if (!c || !d_tiles || !tile_bytes || n_tiles < 0) return set_error(INNFER_ERR_INVALID, "gather_tiles: bad arguments");
    if (c->nranks == 1 || n_tiles == 0) return INNFER_OK;
    hipStream_t s = (hipStream_t)stream;
    int first = 0, count = 0;
    if (c->rank != 0) {
        innfer_shard_tiles(n_tiles, c->nranks, c->rank, &first, &count);
        if (!count) return INNFER_OK;
        INNFER_NCCL(g_rccl.Send(d_tiles, (size_t)count * tile_bytes, ncclUint8, 0, c->comm, s));
        return INNFER_OK;
    }
    INNFER_NCCL(g_rccl.GroupStart());
    for (int r = 1; r < c->nranks; ++r) {
        innfer_shard_tiles(n_tiles, c->nranks, r, &first, &count);
        if (!count) continue;
        ncclResult_t res = g_rccl.Recv((char*)d_tiles + (size_t)first * tile_bytes, (size_t)count * tile_bytes, ncclUint8, r, c->comm, s);
        if (res != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return set_error(INNFER_ERR_HIP, "ncclRecv from rank %d failed: %s", r, g_rccl.GetErrorString(res));
        }
    }
    INNFER_NCCL(g_rccl.GroupEnd());
    return INNFER_OK;
}

extern "C" int innfer_comm_broadcast(innfer_comm_t c, void* d_buf, size_t bytes, int root, void* stream) {
    if (!c || !d_buf || root < 0 || root >= c->nranks) return set_error(INNFER_ERR_INVALID, "comm_broadcast: bad arguments");
    if (c->nranks == 1 || !bytes) return INNFER_OK;
    INNFER_NCCL(g_rccl.Broadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, (hipStream_t)stream));
    return INNFER_OK;
}
