// lzx_comm.hip -- the per-iteration exchange of the row-partitioned Lanczos loop.
//
// Two transports behind the same two operations (sum one double across ranks; all-gather the owned
// slices of a vector into every rank's full-length copy):
//   * RCCL (one process per GPU, xGMI): ncclAllReduce on 1 double, ncclAllGather of n_loc_pad doubles.
//     librccl.so.1 is resolved with dlopen at communicator creation, so liblzx.so itself loads on a
//     machine without RCCL and shares the copy a host program (e.g. PyTorch) may already have loaded.
//   * local (all handles in one process -- the reference's two-cards model generalised,
//     parallel-two-cards/lib/cu_lanczos.cu:125,158): device-to-device copies ordered by events.
//   * peer windows (one process per rank, lzx_ipc.hip): buffers mapped across processes, data pushed by the sender's
//     kernel, ordering by sequence numbers in device memory; dispatched from the same operations below.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "lzx_internal.h"

namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // optional (NCCL >= 2.18 / the RCCL of ROCm 6+): a second communicator over the same ranks for the exchange stream
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *) = nullptr;
};
RcclApi g_rccl;

int rccl_load()
{
    if (g_rccl.lib) return LZX_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *nm : names) {
        lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) LZX_FAIL(LZX_ERR_COMM, "cannot load librccl: %s", dlerror());
#define SYM(field, name)                                                              \
    do {                                                                              \
        *reinterpret_cast<void **>(&g_rccl.field) = dlsym(lib, name);                 \
        if (!g_rccl.field) LZX_FAIL(LZX_ERR_COMM, "librccl lacks symbol %s", name);   \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    *reinterpret_cast<void **>(&g_rccl.CommSplit) = dlsym(lib, "ncclCommSplit");   // may be absent: one communicator then
    g_rccl.lib = lib;
    return LZX_OK;
}

#define LZX_NCCL(call)                                                                        \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess)                                                                \
            LZX_FAIL(LZX_ERR_COMM, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
    } while (0)

}  // namespace

extern "C" int lzx_comm_unique_id(uint8_t id[128])
{
    if (!id) LZX_FAIL(LZX_ERR_ARG, "lzx_comm_unique_id: null id");
    LZX_TRY(rccl_load());
    ncclUniqueId u;
    LZX_NCCL(g_rccl.GetUniqueId(&u));
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, 128);
    return LZX_OK;
}

extern "C" int lzx_comm_init_rank(lzx_handle c, const uint8_t id[128], int rank, int world)
{
    if (!c || !id || world < 1 || rank < 0 || rank >= world) LZX_FAIL(LZX_ERR_ARG, "lzx_comm_init_rank: bad argument");
    if (c->d_row_ptr) LZX_FAIL(LZX_ERR_STATE, "wire the communicator before handing over the graph");
    if (c->comm_kind != 0) LZX_FAIL(LZX_ERR_STATE, "handle already has a communicator");
    LZX_TRY(rccl_load());
    LZX_HIP(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclComm_t comm = nullptr;
    LZX_NCCL(g_rccl.CommInitRank(&comm, world, u, rank));
    c->nccl_comm = comm;
    c->comm_kind = 2;
    c->world = world;
    c->rank = rank;
    // The exchange stream gets a communicator of its own: RCCL serialises the operations of ONE communicator in issue
    // order whatever streams they are given, so the sparse send / receive group of chunk 1 (stream2) would otherwise
    // queue behind -- or ahead of -- the two-double all-reduce of the main stream instead of overlapping it.
    // ncclCommSplit is itself collective; whether the split communicator is USED must not be a per-rank decision (a rank on
    // the parent and a peer on the split one would wait for each other for ever): all ranks agree on the outcome, and unless
    // it succeeded everywhere everybody stays on the one communicator.
    c->nccl_comm2 = nullptr;
    ncclComm_t comm2 = nullptr;
    bool split_ok = false, all_ok = false;
    if (g_rccl.CommSplit) split_ok = g_rccl.CommSplit(comm, 0, rank, &comm2, nullptr) == ncclSuccess && comm2;
    LZX_TRY(lzx_comm_agree(c, g_rccl.CommSplit ? split_ok : false, &all_ok));
    if (all_ok) c->nccl_comm2 = comm2;
    else if (split_ok && comm2) (void)g_rccl.CommDestroy(comm2);
    return LZX_OK;
}

int lzx_comm_agree(lzx_ctx *c, bool ok, bool *all_ok)
{
    *all_ok = ok;
    if (c->comm_kind == 3) return lzx_comm_ipc_agree(c, ok, all_ok);
    if (c->comm_kind != 2 || !c->nccl_comm) return LZX_OK;
    double v = ok ? 1.0 : 0.0;
    LZX_HIP(hipSetDevice(c->device));
    // d_scal[7]: allocated with the handle, so nothing here can fail for lack of memory
    LZX_HIP(hipMemcpyAsync(c->d_scal + 7, &v, sizeof(double), hipMemcpyHostToDevice, c->stream));
    LZX_NCCL(g_rccl.AllReduce(c->d_scal + 7, c->d_scal + 7, 1, ncclDouble, ncclMin, static_cast<ncclComm_t>(c->nccl_comm), c->stream));
    LZX_HIP(hipMemcpyAsync(&v, c->d_scal + 7, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    LZX_HIP(hipStreamSynchronize(c->stream));
    *all_ok = v == 1.0;
    return LZX_OK;
}

extern "C" int lzx_create_group(lzx_handle *out, int n_devices, const int *device_ids)
{
    if (!out || n_devices < 1 || n_devices > 64) LZX_FAIL(LZX_ERR_ARG, "lzx_create_group: bad argument (1 .. 64 handles)");
    for (int i = 0; i < n_devices; ++i) out[i] = nullptr;
    int rc = LZX_OK;
    for (int i = 0; i < n_devices && rc == LZX_OK; ++i) rc = lzx_create(&out[i], device_ids ? device_ids[i] : i);
    if (rc == LZX_OK && n_devices > 1) rc = lzx_comm_init_local(out, n_devices);
    if (rc != LZX_OK) {   // (the failing call's message stands)
        for (int i = 0; i < n_devices; ++i) {
            if (out[i]) lzx_destroy(out[i]);
            out[i] = nullptr;
        }
    }
    return rc;
}

extern "C" int lzx_comm_init_local(lzx_handle *hs, int world)
{
    if (!hs || world < 1) LZX_FAIL(LZX_ERR_ARG, "lzx_comm_init_local: bad argument");
    for (int p = 0; p < world; ++p) {
        if (!hs[p]) LZX_FAIL(LZX_ERR_ARG, "lzx_comm_init_local: null handle");
        if (hs[p]->d_row_ptr) LZX_FAIL(LZX_ERR_STATE, "wire the communicator before handing over the graph");
        if (hs[p]->comm_kind != 0) LZX_FAIL(LZX_ERR_STATE, "handle already has a communicator");
    }
    for (int p = 0; p < world; ++p) {
        lzx_ctx *c = hs[p];
        c->peers = new lzx_ctx *[world];
        for (int q = 0; q < world; ++q) c->peers[q] = hs[q];
        c->comm_kind = 1;
        c->world = world;
        c->rank = p;
        // peers on other GPUs: let copies go straight over xGMI where the platform allows it
        c->mail_ok = world <= 64;
        for (int q = 0; q < world; ++q) {
            if (hs[q]->device == c->device) continue;
            (void)hipSetDevice(c->device);
            int can = 0;
            bool ok = false;
            if (hipDeviceCanAccessPeer(&can, c->device, hs[q]->device) == hipSuccess && can) {
                hipError_t e = hipDeviceEnablePeerAccess(hs[q]->device, 0);
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                if (!ok) (void)hipGetLastError();
            }
            // Mailboxes only among handles of ONE device: a peer GPU's kernel stores reach this GPU's memory past its L2, and
            // nothing tested here says a line of the mailbox this GPU's L2 still holds from two iterations ago is dropped in
            // time (no box with two GPUs was available in any round).  Across devices the pair travels by copies, as before.
            (void)ok;
            c->mail_ok = false;
        }
    }
    // the mailboxes of the two-double reduction (lzx_internal.h: d_mail); without them the group reduces through copies
    for (int p = 0; p < world; ++p) {
        lzx_ctx *c = hs[p];
        (void)hipSetDevice(c->device);
        if (hipMalloc(reinterpret_cast<void **>(&c->d_mail), sizeof(double) * 2 * 64 * 2) != hipSuccess) {
            (void)hipGetLastError();
            c->d_mail = nullptr;
        } else {
            (void)hipMemset(c->d_mail, 0, sizeof(double) * 2 * 64 * 2);
        }
    }
    return LZX_OK;
}

void lzx_comm_release(lzx_ctx *c)
{
    if (c->comm_kind == 2 && c->nccl_comm2 && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(c->nccl_comm2));
    if (c->comm_kind == 2 && c->nccl_comm && g_rccl.CommDestroy) {
        (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(c->nccl_comm));
    }
    c->nccl_comm = c->nccl_comm2 = nullptr;
    lzx_comm_ipc_release(c);
    if (c->d_mail) (void)hipFree(c->d_mail);
    c->d_mail = nullptr;
    c->mail_ok = false;
    delete[] c->peers;
    c->peers = nullptr;
    c->comm_kind = 0;
    c->world = 1;
    c->rank = 0;
}

static hipStream_t pick(lzx_ctx *c, bool s2) { return s2 ? c->stream2 : c->stream; }
// the communicator of the stream an operation is queued on (the exchange stream has its own when RCCL can split)
static ncclComm_t pick_comm(lzx_ctx *c, bool s2) { return static_cast<ncclComm_t>(s2 && c->nccl_comm2 ? c->nccl_comm2 : c->nccl_comm); }

// Everything queued so far on every handle's `from` stream happens before what is queued next on every handle's
// `to` stream (all pairs, own handle included when the two streams differ).
int lzx_comm_order(std::vector<lzx_ctx *> &cs, bool from_stream2, bool to_stream2)
{
    for (lzx_ctx *c : cs) {
        LZX_HIP(hipSetDevice(c->device));
        LZX_HIP(hipEventRecord(from_stream2 ? c->ev_phase2 : c->ev_phase, pick(c, from_stream2)));
    }
    for (lzx_ctx *c : cs) {
        LZX_HIP(hipSetDevice(c->device));
        for (lzx_ctx *p : cs)
            if (p != c || from_stream2 != to_stream2)
                LZX_HIP(hipStreamWaitEvent(pick(c, to_stream2), from_stream2 ? p->ev_phase2 : p->ev_phase, 0));
    }
    return LZX_OK;
}

// Make every stream in cs wait for everything queued so far on every other stream in cs.
static int cross_barrier(std::vector<lzx_ctx *> &cs, bool s2 = false) { return lzx_comm_order(cs, s2, s2); }

bool lzx_comm_mail_usable(std::vector<lzx_ctx *> &cs)
{
    if (cs.empty() || cs[0]->comm_kind != 1 || (int)cs.size() != cs[0]->world) return false;
    for (lzx_ctx *c : cs)
        if (!c->mail_ok || !c->d_mail) return false;
    return true;
}

// Every handle's reduce kernel writes [u_j . w, ||u_j||^2] of its rank into slot [rank] of EVERY handle's mailbox of this
// parity; one cross-handle barrier later every handle's vector kernel may read its own mailbox.
int lzx_comm_mail_reduce2(std::vector<lzx_ctx *> &cs, u32 parity, bool first)
{
    const u32 world = (u32)cs.size();
    for (lzx_ctx *c : cs) {
        LZX_HIP(hipSetDevice(c->device));
        MailPeers mp;
        for (u32 q = 0; q < 64; ++q) mp.slot[q] = q < world ? cs[q]->d_mail + ((size_t)parity * 64 + (size_t)c->rank) * 2 : nullptr;
        LZX_TRY(lzx_launch_reduce2_mail(c, c->d_partials, lzx_spmv_partials(c), c->d_partials2, first ? 0 : c->np2_last, mp, world));
    }
    return cross_barrier(cs);
}


bool lzx_comm_fused_reduce2(const lzx_ctx *c) { return c->comm_kind == 3; }

int lzx_comm_reduce2_allreduce(lzx_ctx *c, const double *pa, u32 na, const double *pb, u32 nb)
{
    return lzx_comm_ipc_allreduce(c, 0, 2, 0, pa, na, pb, nb);
}

__global__ void k_sum_ranks_n(const double *vals, int world, u32 count, double *out)
{
    for (u32 i = 0; i < count; ++i) {
        double s = 0.0;
        for (int p = 0; p < world; ++p) s += vals[(size_t)p * count + i];  // rank order: fixed
        out[i] = s;
    }
}

// d_scal[slot + i] <- sum over ranks of d_scal[slot + i], i < count, on every handle.
int lzx_comm_allreduce_sum(std::vector<lzx_ctx *> &cs, u32 slot, u32 count)
{
    lzx_ctx *c0 = cs[0];
    if (!lzx_exchanges(c0)) return LZX_OK;
    if (c0->comm_kind == 2) {
        LZX_NCCL(g_rccl.AllReduce(c0->d_scal + slot, c0->d_scal + slot, count, ncclDouble, ncclSum,
                                  static_cast<ncclComm_t>(c0->nccl_comm), c0->stream));
        return LZX_OK;
    }
    if (c0->comm_kind == 3) return lzx_comm_ipc_allreduce(c0, slot, count, 0);
    // local: rank 0 collects, sums in rank order, hands the totals back
    const int world = c0->world;
    if ((int)cs.size() != world) LZX_FAIL(LZX_ERR_STATE, "local communicator needs all %d handles", world);
    if (slot + count > 8 || (size_t)world * count > 64) LZX_FAIL(LZX_ERR_LIMIT, "local all-reduce: %u values on %d ranks", count, world);
    LZX_TRY(cross_barrier(cs));
    LZX_HIP(hipSetDevice(c0->device));
    for (int p = 0; p < world; ++p)
        LZX_HIP(hipMemcpyAsync(c0->d_scal + 8 + (size_t)p * count, cs[p]->d_scal + slot, sizeof(double) * count, hipMemcpyDefault, c0->stream));
    hipLaunchKernelGGL(k_sum_ranks_n, dim3(1), dim3(1), 0, c0->stream, c0->d_scal + 8, world, count, c0->d_scal + slot);
    for (int p = 1; p < world; ++p)
        LZX_HIP(hipMemcpyAsync(cs[p]->d_scal + slot, c0->d_scal + slot, sizeof(double) * count, hipMemcpyDefault, c0->stream));
    LZX_TRY(cross_barrier(cs));
    return LZX_OK;
}

// dst_full[i][p * cnt ...] <- src_loc[p][0 .. cnt) for every rank p, on every handle i; on the handles' main streams
// or (on_stream2) on their exchange streams.
int lzx_comm_allgather(std::vector<lzx_ctx *> &cs, const double *const *src_loc, double *const *dst_full, size_t cnt,
                       bool on_stream2, bool peers_idle)
{
    lzx_ctx *c0 = cs[0];
    if (cnt == 0) return LZX_OK;
    if (!lzx_exchanges(c0)) {
        if (dst_full[0] != src_loc[0])
            LZX_HIP(hipMemcpyAsync(dst_full[0], src_loc[0], cnt * sizeof(double), hipMemcpyDeviceToDevice, pick(c0, on_stream2)));
        return LZX_OK;
    }
    if (c0->comm_kind == 2) {
        LZX_NCCL(g_rccl.AllGather(src_loc[0], dst_full[0], cnt, ncclDouble, pick_comm(c0, on_stream2), pick(c0, on_stream2)));
        return LZX_OK;
    }
    if (c0->comm_kind == 3) return lzx_comm_ipc_allgather(c0, src_loc[0], dst_full[0], cnt, on_stream2, peers_idle);
    const int world = c0->world;
    if ((int)cs.size() != world) LZX_FAIL(LZX_ERR_STATE, "local communicator needs all %d handles", world);
    LZX_TRY(cross_barrier(cs, on_stream2));
    for (int i = 0; i < world; ++i) {
        LZX_HIP(hipSetDevice(cs[i]->device));
        for (int p = 0; p < world; ++p) {
            double *dst = dst_full[i] + (size_t)p * cnt;
            if (dst == src_loc[p]) continue;
            LZX_HIP(hipMemcpyAsync(dst, src_loc[p], cnt * sizeof(double), hipMemcpyDefault, pick(cs[i], on_stream2)));
        }
    }
    LZX_TRY(cross_barrier(cs, on_stream2));
    return LZX_OK;
}

// Chunk 1 of the per-iteration exchange in its sparse form (lzx_graph.hip, k_sx_mark): rank i receives from rank r
// only the entries of r's slice that i's rows reference, packed, into its own segment for r behind chunk 0.  Every
// rank packs once for all peers (one gather kernel), then one grouped send/receive per peer (RCCL: ncclSend / ncclRecv
// inside one group, xGMI point to point -- no ring, no rank forwards what another one needs) or, inside one process,
// one device-to-device copy per pair.  On the exchange streams.
int lzx_comm_sparse_chunk1(std::vector<lzx_ctx *> &cs, const double *const *slice_loc, bool peers_idle)
{
    lzx_ctx *c0 = cs[0];
    const int world = c0->world;
    if (!c0->sparse) LZX_FAIL(LZX_ERR_STATE, "sparse exchange was not prepared");
    if (c0->comm_kind == 3) {
        LZX_TRY(lzx_launch_sx_pack(c0, slice_loc[0], c0->stream2));
        return lzx_comm_ipc_sparse_chunk1(c0, peers_idle);
    }
    if (c0->comm_kind == 2) {
        lzx_ctx *c = c0;
        const int me = c->rank;
        ncclComm_t comm = pick_comm(c, true);
        LZX_TRY(lzx_launch_sx_pack(c, slice_loc[0], c->stream2));
        double *seg = c->d_xbuf + (size_t)world * c->xs0;
        const u32 own = c->sx_send_off[me + 1] - c->sx_send_off[me];
        if (own)
            LZX_HIP(hipMemcpyAsync(seg + c->sx_recv_off[me], c->d_sx_sendbuf + c->sx_send_off[me], sizeof(double) * own,
                                   hipMemcpyDeviceToDevice, c->stream2));
        LZX_NCCL(g_rccl.GroupStart());
        for (int p = 0; p < world; ++p) {
            if (p == me) continue;
            const u32 ns = c->sx_send_off[p + 1] - c->sx_send_off[p], nr = c->sx_recv_off[p + 1] - c->sx_recv_off[p];
            if (ns) LZX_NCCL(g_rccl.Send(c->d_sx_sendbuf + c->sx_send_off[p], ns, ncclDouble, p, comm, c->stream2));
            if (nr) LZX_NCCL(g_rccl.Recv(seg + c->sx_recv_off[p], nr, ncclDouble, p, comm, c->stream2));
        }
        LZX_NCCL(g_rccl.GroupEnd());
        return LZX_OK;
    }
    if ((int)cs.size() != world) LZX_FAIL(LZX_ERR_STATE, "local communicator needs all %d handles", world);
    for (int i = 0; i < world; ++i) {
        LZX_HIP(hipSetDevice(cs[i]->device));
        LZX_TRY(lzx_launch_sx_pack(cs[i], slice_loc[i], cs[i]->stream2));
    }
    LZX_TRY(cross_barrier(cs, true));
    for (int i = 0; i < world; ++i) {
        lzx_ctx *c = cs[i];
        LZX_HIP(hipSetDevice(c->device));
        double *seg = c->d_xbuf + (size_t)world * c->xs0;
        for (int r = 0; r < world; ++r) {
            const u32 nr = c->sx_recv_off[r + 1] - c->sx_recv_off[r];
            const u32 ns = cs[r]->sx_send_off[i + 1] - cs[r]->sx_send_off[i];
            if (nr != ns) LZX_FAIL(LZX_ERR_STATE, "sparse exchange: rank %d expects %u entries of rank %d, which packs %u", i, nr, r, ns);
            if (nr)
                LZX_HIP(hipMemcpyAsync(seg + c->sx_recv_off[r], cs[r]->d_sx_sendbuf + cs[r]->sx_send_off[i], sizeof(double) * nr,
                                       hipMemcpyDefault, c->stream2));
        }
    }
    LZX_TRY(cross_barrier(cs, true));
    return LZX_OK;
}

// One-off check when the graph is reshaped (RCCL transport): every rank derives by itself what it sends to and receives
// from each peer in the sparse chunk (k_sx_mark + a scan) -- nothing is negotiated, so a disagreement would only show as
// a grouped ncclSend / ncclRecv that never completes, or (equal lengths, other members) as a silently wrong product.  All
// ranks gather everybody's [send counts | receive counts | send hashes | receive hashes] and check EVERY pair (what p packs
// for q == what q expects from p, in length and in content): all of them reach the same verdict, so a mismatch is an error on
// every rank at once (LZX_ERR_STATE), never a hang.
void lzx_sx_check_message(const lzx_ctx *c, std::vector<u32> &mine)
{
    const u32 world = (u32)c->world;
    mine.assign(6 * (size_t)world, 0u);
    for (u32 p = 0; p < world; ++p) {
        mine[p] = c->sx_send_off[p + 1] - c->sx_send_off[p];
        mine[world + p] = c->sx_recv_off[p + 1] - c->sx_recv_off[p];
        const u64 hs = p < c->sx_send_hash.size() ? c->sx_send_hash[p] : 0, hr = p < c->sx_recv_hash.size() ? c->sx_recv_hash[p] : 0;
        mine[2 * world + 2 * p] = (u32)hs;
        mine[2 * world + 2 * p + 1] = (u32)(hs >> 32);
        mine[4 * world + 2 * p] = (u32)hr;
        mine[4 * world + 2 * p + 1] = (u32)(hr >> 32);
    }
}

int lzx_sx_check_pairs(const std::vector<u32> &all, u32 world)
{
    const size_t M = 6 * (size_t)world;
    for (u32 p = 0; p < world; ++p)
        for (u32 q = 0; q < world; ++q) {
            const u32 *mp = all.data() + (size_t)p * M, *mq = all.data() + (size_t)q * M;
            const u32 sends = mp[q], expects = mq[world + p];
            if (sends != expects)
                LZX_FAIL(LZX_ERR_STATE, "sparse exchange: rank %u packs %u entries for rank %u, which expects %u", p, sends, q, expects);
            if (mp[2 * world + 2 * q] != mq[4 * world + 2 * p] || mp[2 * world + 2 * q + 1] != mq[4 * world + 2 * p + 1])
                LZX_FAIL(LZX_ERR_STATE, "sparse exchange: rank %u packs %u entries for rank %u, which expects as many but OTHER ones "
                                        "(the matrix handed over is not symmetric, or the ranks hold different graphs)", p, sends, q);
        }
    return LZX_OK;
}

int lzx_comm_check_sparse_local(std::vector<lzx_ctx *> &cs)
{
    const u32 world = (u32)cs.size();
    if (world < 2 || !cs[0]->sparse) return LZX_OK;
    std::vector<u32> all, mine;
    for (lzx_ctx *c : cs) {
        if (!c->sparse || (u32)c->world != world || c->sx_send_off.size() != world + 1 || c->sx_recv_off.size() != world + 1)
            LZX_FAIL(LZX_ERR_STATE, "sparse exchange: the handles of the group were handed graphs with different exchange options");
        lzx_sx_check_message(c, mine);
        all.insert(all.end(), mine.begin(), mine.end());
    }
    return lzx_sx_check_pairs(all, world);
}

int lzx_comm_check_sparse(lzx_ctx *c)
{
    if (c->comm_kind == 3 && c->sparse) return lzx_comm_ipc_check_sparse(c);
    if (c->comm_kind != 2 || !c->sparse) return LZX_OK;
    const u32 world = (u32)c->world;
    std::vector<u32> mine;
    lzx_sx_check_message(c, mine);
    std::vector<u32> all(mine.size() * world);
    u32 *d = nullptr;
    LZX_HIP(hipSetDevice(c->device));
    // the staging buffer is a rank-local allocation in front of a collective: agree on it first (a rank that could not
    // allocate would otherwise leave its peers in the all-gather below)
    hipError_t ea = hipMalloc(reinterpret_cast<void **>(&d), sizeof(u32) * (mine.size() + all.size()));
    if (ea != hipSuccess) { (void)hipGetLastError(); d = nullptr; }
    bool all_ok = false;
    const int rca = lzx_comm_agree(c, ea == hipSuccess, &all_ok);
    if (rca != LZX_OK || !all_ok) {
        if (d) (void)hipFree(d);
        if (rca != LZX_OK) return rca;
        LZX_FAIL(ea == hipSuccess ? LZX_ERR_STATE : LZX_ERR_NOMEM, "sparse exchange check: %s", ea == hipSuccess ? "a peer rank could not allocate its staging buffer" : hipGetErrorString(ea));
    }
    hipError_t e = hipMemcpyAsync(d, mine.data(), sizeof(u32) * mine.size(), hipMemcpyHostToDevice, c->stream);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess)
        r = g_rccl.AllGather(d, d + mine.size(), mine.size(), ncclUint32, static_cast<ncclComm_t>(c->nccl_comm), c->stream);
    if (e == hipSuccess && r == ncclSuccess)
        e = hipMemcpyAsync(all.data(), d + mine.size(), sizeof(u32) * all.size(), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (r != ncclSuccess) LZX_FAIL(LZX_ERR_COMM, "sparse exchange check: %s", g_rccl.GetErrorString(r));
    LZX_HIP(e);
    return lzx_sx_check_pairs(all, world);
}

// N4 (SURVEY 8 f): the per-iteration all-gather with the slices rounded to fp32 -- half the bytes on the wire.  Every
// rank converts its slice, the floats are gathered, and every rank widens ALL slices (its own too: all ranks must
// multiply the same vector) into the fp64 buffer the SpMV reads; sums stay fp64.  Main streams.
int lzx_comm_allgather_fp32(std::vector<lzx_ctx *> &cs, const double *const *slice_loc, bool peers_idle)
{
    lzx_ctx *c0 = cs[0];
    const int world = c0->world;
    const size_t cnt = c0->xs;
    for (size_t i = 0; i < cs.size(); ++i) {
        LZX_HIP(hipSetDevice(cs[i]->device));
        LZX_TRY(lzx_launch_to_f32(cs[i], slice_loc[i], cs[i]->d_xf32_send, cnt));
    }
    if (c0->comm_kind == 2) {
        LZX_NCCL(g_rccl.AllGather(c0->d_xf32_send, c0->d_xf32_full, cnt, ncclFloat, static_cast<ncclComm_t>(c0->nccl_comm), c0->stream));
    } else if (c0->comm_kind == 3) {
        LZX_TRY(lzx_comm_ipc_allgather_f32(c0, c0->d_xf32_send, c0->d_xf32_full, cnt, peers_idle));
    } else {
        if ((int)cs.size() != world) LZX_FAIL(LZX_ERR_STATE, "local communicator needs all %d handles", world);
        LZX_TRY(cross_barrier(cs));
        for (int i = 0; i < world; ++i) {
            LZX_HIP(hipSetDevice(cs[i]->device));
            for (int p = 0; p < world; ++p)
                LZX_HIP(hipMemcpyAsync(cs[i]->d_xf32_full + (size_t)p * cnt, cs[p]->d_xf32_send, cnt * sizeof(float), hipMemcpyDefault, cs[i]->stream));
        }
        LZX_TRY(cross_barrier(cs));
    }
    for (size_t i = 0; i < cs.size(); ++i) {
        LZX_HIP(hipSetDevice(cs[i]->device));
        LZX_TRY(lzx_launch_to_f64(cs[i], cs[i]->d_xf32_full, cs[i]->d_xbuf, (u64)world * cnt));
    }
    return LZX_OK;
}
