// comm.cpp -- the multi-GPU exchange of the distance path behind the C ABI: RCCL over xGMI, one communicator rank per process.
//
// The reference has no counterpart (one process, OpenMP: src/pairsnp.hpp:380-382).  north_star / SURVEY.md 8e: the N x N pair space
// is block-partitioned over the GPUs of one node, every rank holds the packed alignment (one broadcast), no collective during
// compute, and the per-rank distance panels are exchanged with an all-gather at the end.  What a host needs for that, and nothing
// more:
//     tracs_comm_unique_id / tracs_comm_create     ncclGetUniqueId / ncclCommInitRank on the current device
//     tracs_bcast_planes                           the packed planes of the rank that read the FASTA -> every rank (ncclBroadcast)
//     tracs_allgather_panels                       in place: rank q's block of a buffer every rank lays out alike -> every rank
//                                                  (row panels of the pair matrix are contiguous: partition.py's fold pairing puts
//                                                  rank q's two chunks at offsets that are NOT in rank order, so this is a group of
//                                                  in-place ncclBroadcasts -- one per rank -- instead of an ncclAllGather + copies)
//     tracs_allreduce                              small agreements (key tables of transcluster, value ranges of the panels)
//     tracs_reduce_scatter                         SITE shards: every rank counts its slice of the sites for ALL pairs (d and the
//                                                  compared-sites counts are sums over sites), rank q receives rows q of the sums
//     tracs_alltoall                               SITE shards, the compact form: every rank packs the upper-triangle cells of every
//                                                  other rank's rows (exchange.hip: 16 bits per cell where the slice's values fit)
//                                                  and block q goes to rank q -- point to point over every xGMI link at once,
//                                                  summed by the receiver (a ring reduce-scatter moves the same bytes over ONE link)
//     tracs_send / tracs_recv                      variable-length COO payloads to the rank that writes the CSV
// RCCL is opened when the first communicator is made (dlopen of librccl.so.1: a process that already holds an RCCL -- PyTorch's --
// gets that one), so a single-GPU host needs no RCCL at all.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllToAll)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_open()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.handle) return TRACS_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { tracs::set_error(std::string("RCCL not found (librccl.so.1): ") + (dlerror() ? dlerror() : "")); return TRACS_E_HIP; }
    Rccl r;
    r.handle = h;
#define TRACS_SYM(field, name) do { r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name)); \
        if (!r.field) { tracs::set_error(std::string("RCCL symbol missing: ") + name); return TRACS_E_HIP; } } while (0)
    TRACS_SYM(GetUniqueId, "ncclGetUniqueId");
    TRACS_SYM(CommInitRank, "ncclCommInitRank");
    TRACS_SYM(CommDestroy, "ncclCommDestroy");
    TRACS_SYM(Broadcast, "ncclBroadcast");
    TRACS_SYM(AllReduce, "ncclAllReduce");
    TRACS_SYM(ReduceScatter, "ncclReduceScatter");
    TRACS_SYM(Send, "ncclSend");
    TRACS_SYM(Recv, "ncclRecv");
    TRACS_SYM(GroupStart, "ncclGroupStart");
    TRACS_SYM(GroupEnd, "ncclGroupEnd");
    TRACS_SYM(AllToAll, "ncclAllToAll");
    TRACS_SYM(CommCount, "ncclCommCount");
    TRACS_SYM(CommUserRank, "ncclCommUserRank");
    TRACS_SYM(GetVersion, "ncclGetVersion");
    TRACS_SYM(GetErrorString, "ncclGetErrorString");
#undef TRACS_SYM
    g_rccl = r;
    return TRACS_OK;
}

#define TRACS_NCCL_CHECK(expr)                                                                              \
    do {                                                                                                    \
        ncclResult_t r__ = (expr);                                                                          \
        if (r__ != ncclSuccess) {                                                                           \
            tracs::set_error(std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "RCCL error")); \
            return TRACS_E_HIP;                                                                             \
        }                                                                                                   \
    } while (0)

}  // namespace

struct tracs_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int tracs_comm_unique_id(void *id, size_t cap)
{
    if (!id || cap < TRACS_COMM_ID_BYTES) { tracs::set_error("tracs_comm_unique_id: buffer of TRACS_COMM_ID_BYTES bytes wanted"); return TRACS_E_ARG; }
    static_assert(TRACS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "tracs_hip.h: TRACS_COMM_ID_BYTES");
    const int rc = rccl_open();
    if (rc) return rc;
    ncclUniqueId u;
    TRACS_NCCL_CHECK(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return TRACS_OK;
}

int tracs_comm_create(const void *id, int rank, int world, tracs_comm **out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { tracs::set_error("tracs_comm_create: bad argument"); return TRACS_E_ARG; }
    *out = nullptr;
    const int rc = rccl_open();
    if (rc) return rc;
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    auto *c = new tracs_comm();
    c->rank = rank; c->world = world;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { tracs::set_error(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); delete c; return TRACS_E_HIP; }
    *out = c;
    return TRACS_OK;
}

void tracs_comm_free(tracs_comm *c)
{
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

// what RCCL itself reports for the communicator (ncclCommUserRank / ncclCommCount), not what the caller passed in
int tracs_comm_rank(const tracs_comm *c)
{
    int r = -1;
    if (!c || !c->comm || !g_rccl.CommUserRank || g_rccl.CommUserRank(c->comm, &r) != ncclSuccess) return -1;
    return r;
}

int tracs_comm_world(const tracs_comm *c)
{
    int w = 0;
    if (!c || !c->comm || !g_rccl.CommCount || g_rccl.CommCount(c->comm, &w) != ncclSuccess) return 0;
    return w;
}

int tracs_rccl_version(void)
{
    int v = 0;
    if (rccl_open() != TRACS_OK || g_rccl.GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

int tracs_bcast(tracs_comm *c, void *buf, size_t bytes, int root, void *stream_)
{
    if (!c || (!buf && bytes) || root < 0 || root >= c->world) { tracs::set_error("tracs_bcast: bad argument"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // (pieces of at most 1 GiB: a single count stays far below 2^31 elements whatever the RCCL build does with it)
    const size_t step = (size_t)1 << 30;
    for (size_t o = 0; o < bytes; o += step) {
        const size_t len = bytes - o < step ? bytes - o : step;
        char *p = static_cast<char *>(buf) + o;
        TRACS_NCCL_CHECK(g_rccl.Broadcast(p, p, len, ncclChar, root, c->comm, stream));
    }
    return TRACS_OK;
}

int tracs_bcast_planes(tracs_comm *c, tracs_alignment *a, int root, void *stream)
{
    if (!c || !a) { tracs::set_error("tracs_bcast_planes: NULL argument"); return TRACS_E_ARG; }
    const int rc = tracs_bcast(c, tracs_alignment_planes(a), tracs_alignment_bytes(a), root, stream);
    if (rc) return rc;
    return c->rank == root ? TRACS_OK : tracs_alignment_touch(a);      // the planes changed under the handle: derived forms are stale
}

int tracs_allgather_panels(tracs_comm *c, void *base, const size_t *offsets, size_t bytes, void *stream_)
{
    if (!c || !base || !offsets) { tracs::set_error("tracs_allgather_panels: NULL argument"); return TRACS_E_ARG; }
    if (bytes == 0) return TRACS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TRACS_NCCL_CHECK(g_rccl.GroupStart());
    for (int q = 0; q < c->world; q++) {
        char *p = static_cast<char *>(base) + offsets[q];
        const ncclResult_t r = g_rccl.Broadcast(p, p, bytes, ncclChar, q, c->comm, stream);
        if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); tracs::set_error(std::string("ncclBroadcast: ") + g_rccl.GetErrorString(r)); return TRACS_E_HIP; }
    }
    TRACS_NCCL_CHECK(g_rccl.GroupEnd());
    return TRACS_OK;
}

int tracs_allreduce(tracs_comm *c, void *buf, size_t count, int dtype, int op, void *stream_)
{
    if (!c || (!buf && count)) { tracs::set_error("tracs_allreduce: NULL argument"); return TRACS_E_ARG; }
    static const ncclDataType_t types[] = {ncclInt64, ncclFloat64, ncclUint32, ncclUint8};
    static const ncclRedOp_t ops[] = {ncclSum, ncclMax, ncclMin};
    if (dtype < 0 || dtype > 3 || op < 0 || op > 2) { tracs::set_error("tracs_allreduce: dtype 0..3 (i64, f64, u32, u8), op 0..2 (sum, max, min)"); return TRACS_E_ARG; }
    if (count == 0) return TRACS_OK;
    TRACS_NCCL_CHECK(g_rccl.AllReduce(buf, buf, count, types[dtype], ops[op], c->comm, static_cast<hipStream_t>(stream_)));
    return TRACS_OK;
}

int tracs_reduce_scatter(tracs_comm *c, void *buf, size_t count_per_rank, int dtype, int op, void *stream_)
{
    if (!c || (!buf && count_per_rank)) { tracs::set_error("tracs_reduce_scatter: NULL argument"); return TRACS_E_ARG; }
    static const ncclDataType_t types[] = {ncclInt64, ncclFloat64, ncclUint32, ncclUint8};
    static const size_t widths[] = {8, 8, 4, 1};
    static const ncclRedOp_t ops[] = {ncclSum, ncclMax, ncclMin};
    if (dtype < 0 || dtype > 3 || op < 0 || op > 2) { tracs::set_error("tracs_reduce_scatter: dtype 0..3 (i64, f64, u32, u8), op 0..2 (sum, max, min)"); return TRACS_E_ARG; }
    if (count_per_rank == 0) return TRACS_OK;
    // in place: rank q's reduced block replaces block q of its own buffer (the other blocks keep this rank's partial values)
    char *mine = static_cast<char *>(buf) + (size_t)c->rank * count_per_rank * widths[dtype];
    TRACS_NCCL_CHECK(g_rccl.ReduceScatter(buf, mine, count_per_rank, types[dtype], ops[op], c->comm, static_cast<hipStream_t>(stream_)));
    return TRACS_OK;
}

int tracs_alltoall(tracs_comm *c, const void *send, void *recv, size_t block_bytes, void *stream_)
{
    if (!c || ((!send || !recv) && block_bytes)) { tracs::set_error("tracs_alltoall: NULL argument"); return TRACS_E_ARG; }
    if (block_bytes == 0) return TRACS_OK;
    // block q of `send` -> rank q, where it lands as block `rank` of `recv`: every pair of ranks talks over its own xGMI link
    TRACS_NCCL_CHECK(g_rccl.AllToAll(send, recv, block_bytes, ncclChar, c->comm, static_cast<hipStream_t>(stream_)));
    return TRACS_OK;
}

int tracs_send(tracs_comm *c, const void *buf, size_t bytes, int peer, void *stream_)
{
    if (!c || (!buf && bytes) || peer < 0 || peer >= c->world || peer == c->rank) { tracs::set_error("tracs_send: bad argument"); return TRACS_E_ARG; }
    if (bytes == 0) return TRACS_OK;
    TRACS_NCCL_CHECK(g_rccl.Send(buf, bytes, ncclChar, peer, c->comm, static_cast<hipStream_t>(stream_)));
    return TRACS_OK;
}

int tracs_recv(tracs_comm *c, void *buf, size_t bytes, int peer, void *stream_)
{
    if (!c || (!buf && bytes) || peer < 0 || peer >= c->world || peer == c->rank) { tracs::set_error("tracs_recv: bad argument"); return TRACS_E_ARG; }
    if (bytes == 0) return TRACS_OK;
    TRACS_NCCL_CHECK(g_rccl.Recv(buf, bytes, ncclChar, peer, c->comm, static_cast<hipStream_t>(stream_)));
    return TRACS_OK;
}

}  // extern "C"
