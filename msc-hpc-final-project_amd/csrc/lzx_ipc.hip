// lzx_ipc.hip -- the third transport of the row-partitioned loop's exchange (lzx_comm.hip has the other two): one process
// per rank, every rank's receive buffers mapped into every peer with HIP's inter-process memory handles, data PUSHED by
// the sender's own kernel straight into the peers' buffers (xGMI point to point between GPUs, plain stores on one GPU), and
// ordering by sequence numbers in a small window of device memory per rank instead of a collective library:
//   * put      one kernel copies this rank's piece into every peer's buffer; its last workgroup to finish writes the
//              operation's sequence number into flag[stream][me] of every peer's window (release, system scope);
//   * wait     a ONE-WAVEFRONT kernel on the receiving stream spins until flag[stream][p] >= that number for every p
//              (acquire, system scope) -- with a deadline on the 100 MHz wall clock: a peer that never arrives ends the
//              wait, marks the window and the host reports LZX_ERR_COMM at its next synchronisation; no wave spins for ever;
//   * all-reduce of <= 8 doubles: one single-wavefront kernel stores the rank's values + sequence number into slot [me] of
//              every peer's mailbox, waits for the world slots of its own, adds them IN RANK ORDER (all ranks get the same bits).
// Only single wavefronts ever spin, so ranks that share one GPU (how this path is tested on a one-GPU box:
// tests/ipc_ranks.py, 2 and 4 processes) cannot starve each other of compute units.
// Replaces, like the other transports, parallel-two-cards/lib/cu_lanczos.cu:62-67 (peer access), 125 / 158 (the two
// cudaMemcpyPeer of every iteration) and the host round trips of its scalar reductions (:104-105, 119-120).
//
// Failure model (round 5): a deadline that expires is FATAL for the communicator.  The rank that saw it marks its
// state broken (every later collective fails at once with LZX_ERR_COMM, nothing is retried on advanced sequence numbers) and
// poisons the err word of every peer's window, which every spinning wavefront polls beside its flag: the peers fail fast
// instead of meeting a board message or a mailbox value of a different operation.  Board messages carry the operation's tag
// and the board's sequence number; a mismatch is the same fatal error.
//
// Why the flags may share a stream's channel: every signal of a rank on one stream is issued in that stream's order and
// carries a number one higher than the last, so "flag >= s" also says that everything the rank put before s has landed.
// Why two mailbox / board parities suffice: a rank can post operation s + 2 only after it completed s + 1, which needed
// every peer's contribution to s + 1, which each peer queued behind its own read of s.
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>

#include "lzx_internal.h"
#include "lzx_spmv_body.h"
#include "lzx_reduce.h"

namespace {

constexpr u32 IPC_MAGIC = 0x4c5a5849u;   // "LZXI"
constexpr u32 IPC_BOARD = 2048;          // bytes one rank can post in a host-level all-gather (the sparse check: 6 * 64 words + head)

struct MailSlot {
    double v[8];
    unsigned long long seq;
    unsigned long long pad[7];
};
static_assert(sizeof(MailSlot) == 128, "one slot = one 128-byte line");

// one rank's window (device memory, fine-grained where the platform exports it; zeroed at creation)
struct Window {
    unsigned long long flag[2][64];          // [stream][source rank] last sequence number that rank signalled here
    MailSlot mail[2][64];                    // [parity][source rank]
    unsigned int done[2];                    // put kernels of this rank: workgroups finished (per stream; back to 0 at the end)
    unsigned int err;                        // 1 + rank that did not arrive before the deadline
    unsigned int pad;
    unsigned char stage[IPC_BOARD];          // this rank's outgoing board message
    unsigned char board[2][64][IPC_BOARD];   // [parity][source rank]
};

struct Blob {   // what lzx_comm_ipc_export hands to the peers (LZX_IPC_BLOB bytes)
    u32 magic, finegrained;
    int pid, device;
    u64 ptr;    // the window's address in the exporting process (ranks of ONE process use it directly)
    u64 nonce;  // per-process random number: a pid alone may repeat across PID namespaces or after reuse
    hipIpcMemHandle_t handle;
};
static_assert(sizeof(Blob) <= LZX_IPC_BLOB, "blob fits");

struct BufMsg {   // one exported receive buffer on the board
    u64 ptr, bytes;
    hipIpcMemHandle_t handle;
};
struct PublishMsg {
    u32 ok, n_bufs;
    int pid, pad;
    u64 nonce;
    BufMsg buf[LZX_IPC_BUFS];
};
struct BoardHead {   // in front of every board message: which collective this is, and the how-many-th of the board
    u32 tag, seq;
};
enum : u32 { TAG_AGREE = 0xa1u, TAG_PUBLISH = 0xb2u, TAG_SPARSE = 0xc3u };
static_assert(sizeof(PublishMsg) + sizeof(BoardHead) <= IPC_BOARD, "publish message fits the board");

// this process's nonce (Blob, PublishMsg): same pid AND same nonce = a rank of this very process, whose pointers may be used as they are
u64 process_nonce()
{
    static const u64 v = [] {
        u64 x = 0;
        if (FILE *f = fopen("/dev/urandom", "rb")) {
            if (fread(&x, sizeof x, 1, f) != 1) x = 0;
            fclose(f);
        }
        x ^= (u64)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull ^ ((u64)getpid() << 32);
        return x ? x : 1ull;
    }();
    return v;
}

struct FlagPeers { unsigned long long *flag[64]; };
struct MailPeersIpc { MailSlot *slot[64]; };
template <typename T> struct PutArgs {
    T *dst[64];
    const T *src[64];
    u32 cnt[64];
};

// The polls are RELAXED system-scope loads (they bypass the caches by themselves); the one acquire fence follows the successful
// poll.  An acquire per poll would invalidate this XCD's L2 every few hundred nanoseconds -- under an SpMV that is running on the
// other stream at that very moment.
// err: this rank's own err word -- set by an expired wait of this rank or poisoned by a peer that gave up (lzx_comm_ipc_check): either
// ends the spin at once.
__device__ __forceinline__ bool spin_until(const unsigned long long *flag, u64 seq, u64 deadline_ticks, const unsigned int *err)
{
    const u64 t0 = wall_clock64();
    bool ok = true;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > deadline_ticks || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { ok = false; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    return ok;
}

__global__ void __launch_bounds__(64) k_ipc_signal(FlagPeers peers, u32 world, u64 seq)
{
    if (threadIdx.x < world) __hip_atomic_store(peers.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(64) k_ipc_wait(const unsigned long long *flags, u32 world, u64 seq, u64 deadline_ticks, unsigned int *err)
{
    if (threadIdx.x < world && !spin_until(flags + threadIdx.x, seq, deadline_ticks, err)) atomicCAS(err, 0u, 1u + threadIdx.x);
}

// a rank that gave up tells everybody: err word of every peer's window (the peers' spinning wavefronts poll it)
struct ErrPeers { unsigned int *err[64]; };
__global__ void __launch_bounds__(64) k_ipc_poison(ErrPeers peers, u32 world, u32 me)
{
    if (threadIdx.x < world && threadIdx.x != me) atomicCAS_system(peers.err[threadIdx.x], 0u, 0x80000000u | (1u + me));
}

// WT: the data as write-through stores (system-scope relaxed atomic stores: nothing stays dirty in this XCD's L2, so the
// workgroup's release fence below has nothing to write back)
template <typename T, bool WT>
__global__ void __launch_bounds__(1024) k_ipc_put(PutArgs<T> a, u32 world, u32 first_peer, FlagPeers sig, u64 seq, unsigned int *done)
{
    // peers in a rotation that starts behind this rank: at any moment the ranks of a node write to different peers
    for (u32 t = 0; t < world; ++t) {
        const u32 p = (first_peer + t) % world;
        const T *__restrict__ s = a.src[p];
        T *__restrict__ d = a.dst[p];
        const u32 n = a.cnt[p];
        for (u32 i = blockIdx.x * 1024u + threadIdx.x; i < n; i += gridDim.x * 1024u) {
            if (WT) __hip_atomic_store(d + i, s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else d[i] = s[i];
        }
    }
    // One release per workgroup, not per wavefront (a system-scope release walks this XCD's whole L2: 16 of them per workgroup
    // made the kernel four times slower): the barrier orders every wavefront's stores before wavefront 0's fence, whose
    // release is cumulative over them; the counter chains the workgroups to the last one, which signals the peers.
    __syncthreads();
    __shared__ bool last;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        if (last) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (last && threadIdx.x < world) __hip_atomic_store(sig.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// op 0: sum in rank order; 1: minimum.  pa != nullptr: the rank's two values are first closed from block partials
// (sum pa[0..na), sum pb[0..nb)) -- k_reduce2 and the all-reduce in one launch.
__global__ void __launch_bounds__(LZX_VEC_BLOCK) k_ipc_allreduce(MailPeersIpc out, const MailSlot *in, double *scal, u32 count, u32 world, u64 seq,
                                                                 u64 deadline_ticks, unsigned int *err, int op, const double *pa, u32 na,
                                                                 const double *pb, u32 nb)
{
    __shared__ double vals[64][8];
    __shared__ double sh[4];
    __shared__ double own[8];
    __shared__ u32 timed_out;
    const u32 p = threadIdx.x;
    if (p == 0) timed_out = 0u;
    if (pa) {
        const double a = block_sum_fixed_256(pa, na, sh);
        __syncthreads();
        const double b = block_sum_fixed_256(pb, nb, sh);   // nb == 0 (first iteration): 0, as k_reduce2 leaves it
        if (p == 0) { own[0] = a; own[1] = b; }
    } else if (p < count) {
        own[p] = scal[p];
    }
    __syncthreads();
    if (p < world) {
        MailSlot *o = out.slot[p];
        for (u32 i = 0; i < count; ++i) o->v[i] = own[i];
        __hip_atomic_store(&o->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!spin_until(&in[p].seq, seq, deadline_ticks, err)) {
            atomicCAS(err, 0u, 1u + p);
            timed_out = 1u;
        }
        for (u32 i = 0; i < count; ++i) vals[p][i] = __hip_atomic_load(&in[p].v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (p < count && !timed_out) {   // a slot that never arrived holds an older operation's values: nothing is written
        double s = vals[0][p];
        for (u32 q = 1; q < world; ++q) s = op == 0 ? s + vals[q][p] : (vals[q][p] < s ? vals[q][p] : s);
        scal[p] = s;
    }
}

u64 deadline_ticks()
{
    // wall_clock64 counts at 100 MHz; LZX_IPC_TIMEOUT_MS overrides the 20 s default
    double ms = 20000.0;
    if (const char *e = getenv("LZX_IPC_TIMEOUT_MS")) {
        const double v = atof(e);
        if (v > 0) ms = v;
    }
    return (u64)(ms * 1e5);
}

hipStream_t pick(lzx_ctx *c, bool s2) { return s2 ? c->stream2 : c->stream; }

// a deadline expired earlier (here or, poisoned, on a peer): the sequence numbers of the ranks no longer describe the same
// operations, so nothing more is queued on this communicator
#define IPC_ALIVE(c)                                                                                                             \
    do {                                                                                                                         \
        if ((c)->ipc->broken)                                                                                                    \
            LZX_FAIL(LZX_ERR_COMM, "peer windows: the communicator is broken (a rank did not arrive within the deadline earlier); destroy the handles and wire new ones"); \
    } while (0)

FlagPeers flag_peers(lzx_ctx *c, bool s2)
{
    FlagPeers f{};
    for (int p = 0; p < c->world; ++p) f.flag[p] = &static_cast<Window *>(c->ipc->peer_win[p])->flag[s2 ? 1 : 0][c->rank];
    return f;
}

int queue_wait(lzx_ctx *c, bool s2, u64 seq)
{
    IPC_ALIVE(c);
    Window *w = static_cast<Window *>(c->ipc->win);
    hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(64), 0, pick(c, s2), w->flag[s2 ? 1 : 0], (u32)c->world, seq, c->ipc->deadline, &w->err);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

// every rank has queued everything before this point on that stream: signal, then wait for everybody's signal
int queue_barrier(lzx_ctx *c, bool s2)
{
    IPC_ALIVE(c);
    const u64 seq = ++c->ipc->seq[s2 ? 1 : 0];
    hipLaunchKernelGGL(k_ipc_signal, dim3(1), dim3(64), 0, pick(c, s2), flag_peers(c, s2), (u32)c->world, seq);
    LZX_HIP(hipGetLastError());
    return queue_wait(c, s2, seq);
}

template <typename T> int queue_put(lzx_ctx *c, bool s2, const PutArgs<T> &a)
{
    IPC_ALIVE(c);
    u64 total = 0;
    for (int p = 0; p < c->world; ++p) total += a.cnt[p];
    static const int grid_env = getenv("LZX_IPC_PUT_GRID") ? atoi(getenv("LZX_IPC_PUT_GRID")) : 0;
    static const int wt_env = getenv("LZX_IPC_PUT_WT") ? atoi(getenv("LZX_IPC_PUT_WT")) : 1;
    const u32 cap = grid_env > 0 ? (u32)grid_env : (u32)c->cu_count;   // 8 MB local copy: 64 workgroups 10.7 us, 256 8.9 us (one release each)
    const u32 grid = (u32)std::min<u64>(cap, std::max<u64>(1, (total + 8191) / 8192));
    const u64 seq = ++c->ipc->seq[s2 ? 1 : 0];
    Window *w = static_cast<Window *>(c->ipc->win);
    auto kern = wt_env ? k_ipc_put<T, true> : k_ipc_put<T, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, pick(c, s2), a, (u32)c->world, (u32)((c->rank + 1) % c->world), flag_peers(c, s2), seq,
                       &w->done[s2 ? 1 : 0]);
    LZX_HIP(hipGetLastError());
    return queue_wait(c, s2, seq);
}

void mark_broken(lzx_ctx *c);

// host-level all-gather of <= IPC_BOARD - 8 bytes per rank through the windows (main stream; returns after a synchronisation).
// Every message is headed by the operation's tag and the board's sequence number: ranks that have fallen out of step -- one of
// them retried after a failure the other never saw -- read a mismatch here, not each other's bytes as something they are not.
int host_allgather(lzx_ctx *c, u32 tag, const void *mine, size_t bytes, std::vector<unsigned char> &all)
{
    if (bytes + sizeof(BoardHead) > IPC_BOARD) LZX_FAIL(LZX_ERR_LIMIT, "peer windows: a board message of %zu bytes", bytes);
    IPC_ALIVE(c);
    lzx_ipc_state *s = c->ipc;
    Window *w = static_cast<Window *>(s->win);
    const BoardHead head{tag, (u32)s->board_seq};
    const u32 parity = (u32)(s->board_seq++ & 1u);
    const size_t words = (sizeof(BoardHead) + bytes + 7) / 8;
    unsigned char tmp[IPC_BOARD] = {};
    memcpy(tmp, &head, sizeof head);
    memcpy(tmp + sizeof head, mine, bytes);
    LZX_HIP(hipSetDevice(c->device));
    LZX_HIP(hipMemcpyAsync(w->stage, tmp, words * 8, hipMemcpyHostToDevice, c->stream));
    PutArgs<unsigned long long> a{};
    for (int p = 0; p < c->world; ++p) {
        a.dst[p] = reinterpret_cast<unsigned long long *>(static_cast<Window *>(s->peer_win[p])->board[parity][c->rank]);
        a.src[p] = reinterpret_cast<const unsigned long long *>(w->stage);
        a.cnt[p] = (u32)words;
    }
    LZX_TRY(queue_put(c, false, a));
    all.assign((size_t)c->world * bytes, 0);
    std::vector<unsigned char> raw((size_t)c->world * IPC_BOARD);
    LZX_HIP(hipMemcpyAsync(raw.data(), w->board[parity], raw.size(), hipMemcpyDeviceToHost, c->stream));
    LZX_HIP(hipStreamSynchronize(c->stream));
    LZX_TRY(lzx_comm_ipc_check(c));
    for (int p = 0; p < c->world; ++p) {
        BoardHead h;
        memcpy(&h, raw.data() + (size_t)p * IPC_BOARD, sizeof h);
        if (h.tag != head.tag || h.seq != head.seq) {
            mark_broken(c);
            LZX_FAIL(LZX_ERR_COMM, "peer windows: rank %d posted board message %u of operation %#x where this rank (%d) is at message %u of operation %#x: "
                                   "the ranks are out of step", p, h.seq, h.tag, c->rank, head.seq, head.tag);
        }
        memcpy(all.data() + (size_t)p * bytes, raw.data() + (size_t)p * IPC_BOARD + sizeof h, bytes);
    }
    return LZX_OK;
}

// this rank gives up on the communicator and says so in every peer's window (best effort: the peers may be gone)
void mark_broken(lzx_ctx *c)
{
    lzx_ipc_state *s = c->ipc;
    if (s->broken) return;
    s->broken = true;
    ErrPeers e{};
    for (int p = 0; p < c->world; ++p) e.err[p] = &static_cast<Window *>(s->peer_win[p])->err;
    hipLaunchKernelGGL(k_ipc_poison, dim3(1), dim3(64), 0, c->stream, e, (u32)c->world, (u32)c->rank);
    (void)hipGetLastError();
    (void)hipStreamSynchronize(c->stream);
}

void close_peer_bufs(lzx_ctx *c)
{
    lzx_ipc_state *s = c->ipc;
    for (int b = 0; b < LZX_IPC_BUFS; ++b) {
        for (int p = 0; p < 64; ++p) {
            if (s->buf[b].peer[p] && s->buf[b].opened[p]) (void)hipIpcCloseMemHandle(s->buf[b].peer[p]);
            s->buf[b].peer[p] = nullptr;
            s->buf[b].opened[p] = false;
            s->buf[b].peer_bytes[p] = 0;
        }
        s->buf[b].mine = nullptr;
        s->buf[b].bytes = 0;
    }
}

// which exported buffer holds ptr, and at what byte offset
bool locate(const lzx_ctx *c, const void *ptr, int *b_out, size_t *off_out)
{
    for (int b = 0; b < LZX_IPC_BUFS; ++b) {
        const char *base = static_cast<const char *>(c->ipc->buf[b].mine);
        if (base && static_cast<const char *>(ptr) >= base && static_cast<const char *>(ptr) < base + c->ipc->buf[b].bytes) {
            *b_out = b;
            *off_out = (size_t)(static_cast<const char *>(ptr) - base);
            return true;
        }
    }
    return false;
}

}  // namespace

extern "C" int lzx_comm_ipc_export(lzx_handle c, uint8_t blob[LZX_IPC_BLOB])
{
    if (!c || !blob) LZX_FAIL(LZX_ERR_ARG, "lzx_comm_ipc_export: bad argument");
    if (c->d_row_ptr) LZX_FAIL(LZX_ERR_STATE, "wire the communicator before handing over the graph");
    if (c->comm_kind != 0) LZX_FAIL(LZX_ERR_STATE, "handle already has a communicator");
    LZX_HIP(hipSetDevice(c->device));
    if (!c->ipc) c->ipc = new lzx_ipc_state();
    lzx_ipc_state *s = c->ipc;
    if (!s->win) {
        void *w = nullptr;
        // fine-grained: a peer's stores to the flags and mailboxes must not be shadowed by a line an L2 of this GPU still holds --
        // another GPU's, or (ranks sharing one GPU) another XCD's.  Ordinary device memory is NOT a fallback (round 5, ADVICE r4:
        // that its lines stay coherent among the eight XCD L2s of one GPU was asserted, never shown): a platform that cannot
        // allocate or export fine-grained device memory does not get this transport.
        if (hipExtMallocWithFlags(&w, sizeof(Window), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            LZX_FAIL(LZX_ERR_COMM, "peer windows: this platform gives no fine-grained device memory for the flags and mailboxes");
        }
        s->finegrained = true;
        if (hipMemset(w, 0, sizeof(Window)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipFree(w);
            LZX_FAIL(LZX_ERR_HIP, "peer windows: cannot clear the window");
        }
        s->win = w;
    }
    Blob b{};
    b.magic = IPC_MAGIC;
    b.finegrained = 1u;
    b.pid = (int)getpid();
    b.nonce = process_nonce();
    b.device = c->device;
    b.ptr = (u64)(uintptr_t)s->win;
    const hipError_t e = hipIpcGetMemHandle(&b.handle, s->win);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        LZX_FAIL(LZX_ERR_COMM, "peer windows: this platform does not export fine-grained device memory to other processes (%s)", hipGetErrorString(e));
    }
    memset(blob, 0, LZX_IPC_BLOB);
    memcpy(blob, &b, sizeof b);
    return LZX_OK;
}

extern "C" int lzx_comm_ipc_init(lzx_handle c, const uint8_t *blobs, int rank, int world)
{
    if (!c || !blobs || world < 1 || world > 64 || rank < 0 || rank >= world) LZX_FAIL(LZX_ERR_ARG, "lzx_comm_ipc_init: bad argument (at most 64 ranks)");
    if (c->d_row_ptr) LZX_FAIL(LZX_ERR_STATE, "wire the communicator before handing over the graph");
    if (c->comm_kind != 0) LZX_FAIL(LZX_ERR_STATE, "handle already has a communicator");
    if (!c->ipc || !c->ipc->win) LZX_FAIL(LZX_ERR_STATE, "lzx_comm_ipc_init: call lzx_comm_ipc_export on this handle first");
    lzx_ipc_state *s = c->ipc;
    LZX_HIP(hipSetDevice(c->device));
    Blob mine;
    memcpy(&mine, blobs + (size_t)rank * LZX_IPC_BLOB, sizeof mine);
    if (mine.magic != IPC_MAGIC || mine.pid != (int)getpid() || mine.nonce != process_nonce() || mine.ptr != (u64)(uintptr_t)s->win)
        LZX_FAIL(LZX_ERR_ARG, "lzx_comm_ipc_init: entry %d of the list is not this handle's own export", rank);
    for (int p = 0; p < world; ++p) {
        Blob b;
        memcpy(&b, blobs + (size_t)p * LZX_IPC_BLOB, sizeof b);
        if (b.magic != IPC_MAGIC) LZX_FAIL(LZX_ERR_ARG, "lzx_comm_ipc_init: entry %d is not an export", p);
        if (p == rank) { s->peer_win[p] = s->win; continue; }
        if (!b.finegrained) LZX_FAIL(LZX_ERR_COMM, "peer windows: the window of rank %d is not fine-grained memory", p);
        const bool same_process = b.pid == (int)getpid() && b.nonce == process_nonce();   // (a pid alone may repeat across PID namespaces)
        if (b.device != c->device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, c->device, b.device) != hipSuccess || !can)
                LZX_FAIL(LZX_ERR_COMM, "peer windows: GPU %d cannot access GPU %d of rank %d", c->device, b.device, p);
            if (same_process) {   // no handle is opened for a rank of this very process: switch the peer mapping on here
                const hipError_t pe = hipDeviceEnablePeerAccess(b.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    LZX_FAIL(LZX_ERR_COMM, "peer windows: cannot enable access from GPU %d to GPU %d of rank %d: %s", c->device, b.device, p, hipGetErrorString(pe));
                (void)hipGetLastError();
            }
        }
        if (same_process) { s->peer_win[p] = (void *)(uintptr_t)b.ptr; continue; }   // a rank of this very process
        void *ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, b.handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < p; ++q)
                if (s->win_opened[q]) { (void)hipIpcCloseMemHandle(s->peer_win[q]); s->win_opened[q] = false; }
            LZX_FAIL(LZX_ERR_COMM, "peer windows: cannot map the window of rank %d (pid %d, GPU %d): %s", p, b.pid, b.device, hipGetErrorString(e));
        }
        s->peer_win[p] = ptr;
        s->win_opened[p] = true;
    }
    s->deadline = deadline_ticks();
    c->comm_kind = 3;
    c->world = world;
    c->rank = rank;
    // everybody is here and reachable, or nobody goes on
    bool all_ok = false;
    int rc = lzx_comm_agree(c, true, &all_ok);
    if (rc != LZX_OK || !all_ok) {
        lzx_comm_ipc_release(c);
        c->comm_kind = 0;
        c->world = 1;
        c->rank = 0;
        if (rc != LZX_OK) return rc;
        LZX_FAIL(LZX_ERR_COMM, "peer windows: a peer rank failed to map the windows");
    }
    return LZX_OK;
}

void lzx_comm_ipc_release(lzx_ctx *c)
{
    if (!c->ipc) return;
    lzx_ipc_state *s = c->ipc;
    (void)hipSetDevice(c->device);
    close_peer_bufs(c);
    for (int p = 0; p < 64; ++p) {
        if (s->win_opened[p]) (void)hipIpcCloseMemHandle(s->peer_win[p]);
        s->win_opened[p] = false;
        s->peer_win[p] = nullptr;
    }
    if (s->win) (void)hipFree(s->win);
    delete s;
    c->ipc = nullptr;
}

// did a wait of this rank run into its deadline?  (host, after a synchronisation)
int lzx_comm_ipc_check(lzx_ctx *c)
{
    if (c->comm_kind != 3 || !c->ipc) return LZX_OK;
    unsigned int err = 0;
    Window *w = static_cast<Window *>(c->ipc->win);
    LZX_HIP(hipSetDevice(c->device));
    LZX_HIP(hipMemcpy(&err, &w->err, sizeof err, hipMemcpyDeviceToHost));
    if (err) {
        // fatal for the communicator (the word stays set, the state is marked, the peers are told)
        mark_broken(c);
        if (err & 0x80000000u)
            LZX_FAIL(LZX_ERR_COMM, "peer windows: rank %u gave up on the communicator (a deadline expired there); this rank is %d of %d", (err & 0x7fffffffu) - 1,
                     c->rank, c->world);
        LZX_FAIL(LZX_ERR_COMM, "peer windows: rank %u did not arrive within the deadline (LZX_IPC_TIMEOUT_MS, default 20 000); this rank is %d of %d", err - 1,
                 c->rank, c->world);
    }
    IPC_ALIVE(c);
    return LZX_OK;
}

int lzx_comm_ipc_agree(lzx_ctx *c, bool ok, bool *all_ok)
{
    unsigned char v = ok ? 1 : 0;
    std::vector<unsigned char> all;
    LZX_TRY(host_allgather(c, TAG_AGREE, &v, 1, all));
    *all_ok = true;
    for (unsigned char a : all) *all_ok = *all_ok && a == 1;
    return LZX_OK;
}

// The receive buffers of the freshly reshaped graph (d_xbuf, d_ybuf, d_xf32_full) become reachable for the peers: every rank
// posts their handles with its own verdict on the hand-over so far; unless every rank posted ok, all of them fail.
int lzx_comm_ipc_publish(lzx_ctx *c, bool ok)
{
    lzx_ipc_state *s = c->ipc;
    close_peer_bufs(c);
    PublishMsg m{};
    m.ok = ok ? 1u : 0u;
    m.pid = (int)getpid();
    m.nonce = process_nonce();
    void *bufs[LZX_IPC_BUFS] = {c->d_xbuf, c->d_ybuf, c->d_xf32_full};
    const size_t bytes[LZX_IPC_BUFS] = {sizeof(double) * c->xlen, sizeof(double) * c->iolen, sizeof(float) * (size_t)c->world * c->xs};
    LZX_HIP(hipSetDevice(c->device));
    for (int b = 0; ok && b < LZX_IPC_BUFS; ++b) {
        if (!bufs[b]) continue;
        m.buf[b].ptr = (u64)(uintptr_t)bufs[b];
        m.buf[b].bytes = bytes[b];
        if (hipIpcGetMemHandle(&m.buf[b].handle, bufs[b]) != hipSuccess) {
            (void)hipGetLastError();
            m.ok = 0;
            lzx_set_error("peer windows: cannot export a receive buffer of %zu bytes", bytes[b]);
        }
    }
    m.n_bufs = LZX_IPC_BUFS;
    std::vector<unsigned char> all;
    LZX_TRY(host_allgather(c, TAG_PUBLISH, &m, sizeof m, all));
    bool all_ok = true;
    for (int p = 0; p < c->world; ++p) {
        PublishMsg q;
        memcpy(&q, all.data() + (size_t)p * sizeof q, sizeof q);
        all_ok = all_ok && q.ok == 1;
    }
    bool mapped = all_ok;
    for (int p = 0; all_ok && mapped && p < c->world; ++p) {
        PublishMsg q;
        memcpy(&q, all.data() + (size_t)p * sizeof q, sizeof q);
        for (int b = 0; b < LZX_IPC_BUFS; ++b) {
            // (sizes may differ: the packed second chunk of the sparse exchange has a different length on every rank)
            if ((q.buf[b].ptr != 0) != (m.buf[b].ptr != 0)) {
                lzx_set_error("peer windows: rank %d has %s receive buffer %d, this rank %s (different options?)", p, q.buf[b].ptr ? "a" : "no", b,
                              m.buf[b].ptr ? "has one" : "has none");
                mapped = false;
                break;
            }
            s->buf[b].peer_bytes[p] = q.buf[b].bytes;
            if (!q.buf[b].ptr) continue;
            if (p == c->rank || (q.pid == (int)getpid() && q.nonce == process_nonce())) { s->buf[b].peer[p] = (void *)(uintptr_t)q.buf[b].ptr; continue; }
            void *ptr = nullptr;
            if (hipIpcOpenMemHandle(&ptr, q.buf[b].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                (void)hipGetLastError();
                lzx_set_error("peer windows: cannot map receive buffer %d of rank %d", b, p);
                mapped = false;
                break;
            }
            s->buf[b].peer[p] = ptr;
            s->buf[b].opened[p] = true;
        }
    }
    for (int b = 0; b < LZX_IPC_BUFS; ++b) {
        s->buf[b].mine = mapped ? bufs[b] : nullptr;
        s->buf[b].bytes = mapped ? bytes[b] : 0;
    }
    // second round: a rank that could not map tells the others before any of them pushes data
    bool all_mapped = false;
    LZX_TRY(lzx_comm_ipc_agree(c, mapped, &all_mapped));
    if (!all_ok || !all_mapped) {
        close_peer_bufs(c);
        if (!ok || !mapped) return ok ? LZX_ERR_COMM : LZX_ERR_STATE;   // this rank's own message is already set
        LZX_FAIL(LZX_ERR_STATE, "graph hand-over: a peer rank failed while reshaping its share or mapping the receive buffers (see that rank's error)");
    }
    return LZX_OK;
}

void lzx_comm_ipc_unpublish(lzx_ctx *c)
{
    if (c->comm_kind == 3 && c->ipc) close_peer_bufs(c);
}

int lzx_comm_ipc_allreduce(lzx_ctx *c, u32 slot, u32 count, int op, const double *pa, u32 na, const double *pb, u32 nb)
{
    if (count > 8 || (pa && count != 2)) LZX_FAIL(LZX_ERR_LIMIT, "peer windows: all-reduce of %u values", count);
    IPC_ALIVE(c);
    lzx_ipc_state *s = c->ipc;
    Window *w = static_cast<Window *>(s->win);
    const u64 seq = ++s->mail_seq;
    const u32 parity = (u32)(seq & 1u);
    MailPeersIpc out{};
    for (int p = 0; p < c->world; ++p) out.slot[p] = &static_cast<Window *>(s->peer_win[p])->mail[parity][c->rank];
    // (only wavefront 0 of the workgroup ever spins; the other three are done after the block sums)
    hipLaunchKernelGGL(k_ipc_allreduce, dim3(1), dim3(LZX_VEC_BLOCK), 0, c->stream, out, w->mail[parity], c->d_scal + slot, count, (u32)c->world, seq,
                       s->deadline, &w->err, op, pa, na, pb, nb);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

// dst_full + p * cnt on every rank <- rank p's src_loc[0 .. cnt); dst_full must lie in one of the published buffers
template <typename T> static int ipc_allgather_t(lzx_ctx *c, const T *src_loc, T *dst_full, size_t cnt, bool s2, bool peers_idle)
{
    int b = 0;
    size_t off = 0;
    if (!locate(c, dst_full, &b, &off)) LZX_FAIL(LZX_ERR_STATE, "peer windows: the all-gather's destination is not a published receive buffer");
    for (int p = 0; p < c->world; ++p)
        if (off + sizeof(T) * cnt * ((size_t)c->rank + 1) > c->ipc->buf[b].peer_bytes[p])
            LZX_FAIL(LZX_ERR_STATE, "peer windows: all-gather beyond the receive buffer of rank %d", p);
    if (cnt > 0xffffffffull) LZX_FAIL(LZX_ERR_LIMIT, "peer windows: slice of %zu entries", cnt);
    // the peers may still be reading what this overwrites unless the caller knows otherwise (in the loop an all-reduce
    // separates every SpMV from the next exchange)
    if (!peers_idle) LZX_TRY(queue_barrier(c, s2));
    PutArgs<T> a{};
    for (int p = 0; p < c->world; ++p) {
        a.dst[p] = reinterpret_cast<T *>(static_cast<char *>(c->ipc->buf[b].peer[p]) + off) + (size_t)c->rank * cnt;
        a.src[p] = src_loc;
        a.cnt[p] = (u32)cnt;
    }
    if (a.dst[c->rank] == src_loc) a.cnt[c->rank] = 0;
    return queue_put(c, s2, a);
}

int lzx_comm_ipc_allgather(lzx_ctx *c, const double *src_loc, double *dst_full, size_t cnt, bool s2, bool peers_idle)
{
    return ipc_allgather_t<double>(c, src_loc, dst_full, cnt, s2, peers_idle);
}

int lzx_comm_ipc_allgather_f32(lzx_ctx *c, const float *src_loc, float *dst_full, size_t cnt, bool peers_idle)
{
    return ipc_allgather_t<float>(c, src_loc, dst_full, cnt, false, peers_idle);
}

// sparse chunk 1: the packed piece for peer p goes behind chunk 0 of p's d_xbuf, at the offset p's own lists give this rank
int lzx_comm_ipc_sparse_chunk1(lzx_ctx *c, bool peers_idle)
{
    lzx_ipc_state *s = c->ipc;
    if (s->sx_dst_off.size() != (size_t)c->world) LZX_FAIL(LZX_ERR_STATE, "peer windows: the sparse exchange was not checked");
    if (!s->buf[0].mine) LZX_FAIL(LZX_ERR_STATE, "peer windows: receive buffers are not published");
    if (!peers_idle) LZX_TRY(queue_barrier(c, true));
    PutArgs<double> a{};
    for (int p = 0; p < c->world; ++p) {
        a.dst[p] = static_cast<double *>(s->buf[0].peer[p]) + (size_t)c->world * c->xs0 + s->sx_dst_off[p];
        a.src[p] = c->d_sx_sendbuf + c->sx_send_off[p];
        a.cnt[p] = c->sx_send_off[p + 1] - c->sx_send_off[p];
        if (sizeof(double) * ((size_t)c->world * c->xs0 + s->sx_dst_off[p] + a.cnt[p]) > s->buf[0].peer_bytes[p])
            LZX_FAIL(LZX_ERR_STATE, "peer windows: sparse chunk beyond the receive buffer of rank %d", p);
    }
    return queue_put(c, true, a);
}

// every rank's [send counts | receive counts | content hashes] on the board: all pairs checked by all ranks (the same verdict
// everywhere), and this rank learns where its piece starts inside every peer's packed chunk 1
int lzx_comm_ipc_check_sparse(lzx_ctx *c)
{
    const u32 world = (u32)c->world;
    std::vector<u32> mine;
    lzx_sx_check_message(c, mine);
    std::vector<unsigned char> raw;
    LZX_TRY(host_allgather(c, TAG_SPARSE, mine.data(), sizeof(u32) * mine.size(), raw));
    std::vector<u32> all(mine.size() * world);
    memcpy(all.data(), raw.data(), sizeof(u32) * all.size());
    LZX_TRY(lzx_sx_check_pairs(all, world));
    c->ipc->sx_dst_off.assign(world, 0);
    for (u32 p = 0; p < world; ++p) {
        u64 off = 0;
        for (int r = 0; r < c->rank; ++r) off += all[(size_t)p * mine.size() + world + r];   // what p receives from the ranks before this one
        c->ipc->sx_dst_off[p] = off;
    }
    return LZX_OK;
}
