// ifa_comm.hip -- collectives of the multi-GPU partitions behind the C ABI (RCCL over xGMI).
//
// What they replace in the reference: the host-driven exchange of
// GpuInferenceWorker::DistributeAndMergeTensors / MergeTensors / DeviceCopy
// (src/transformer/inference_worker.cc:2148-2335) over GpuInfGlobalData's mutex + 1 us sleeps
// (src/transformer/gpu_inf_global_data.cu:25-199):
//   BY_TENSOR : the per-rank partial [T][dim] F16 products after wo and after w2 are summed over the
//               group                                          -> ifa_allreduce_sum_f16
//   BY_LAYER  : the [T][dim] F16 layer output goes to the next device group -> ifa_send / ifa_recv
//   the last group announces the chosen token                  -> ifa_broadcast
//   vocabulary-sharded lm_head: (best value, id) per rank      -> ifa_allgather
// Every call only enqueues on the caller's stream (explicit stream argument, no hidden synchronisation,
// capturable into a hipGraph); errors are codes + ifa_last_error().  One communicator per (rank, group):
//   * one process per GPU (bench.py under torch.distributed.run): rank 0 calls ifa_comm_unique_id, ships the 128
//     bytes to its peers by any host channel, every rank calls ifa_comm_init_rank;
//   * one process, one host thread per GPU (the C++ InferenceEngine, like the reference's one sslib::Thread per
//     GPU, inference_engine.cc:1203-1206): ifa_comm_init_all creates the communicators of all ranks at once.
//
// Loopback groups: RCCL refuses a communicator with two ranks on one device, which is all a 1-GPU box can offer.  When
// ifa_comm_init_all is given a device list with duplicates, the ranks become an in-process group whose collectives are
// plain kernels / copies on that device with a host rendezvous between the rank threads (host-synchronous, not
// capturable: ifa_comm_capturable() == 0).  Same results by construction (sums in half, rank order); it exists so that
// the multi-rank paths -- engine threads, slicing, merges, hand-over, distributed argmax -- run in the GPU tests.
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "ifa_host.h"
#include "ifa_device.h"

namespace {

struct LocalGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool aborted = false;       // set by ifa_comm_abort: every waiter wakes up and every later call fails
    const void *sendp[64];
    void *recvp[64];
    struct Slot { const void *src = nullptr; size_t bytes = 0; bool full = false, refused = false; } slots[64][64];
    // false: the group was aborted (a peer failed): the caller must return an error instead of touching peer buffers
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t g = gen;
        if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g || aborted; });
        return !aborted;
    }
    void abort()
    {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};

struct PtrTable { const void *send[64]; void *recv[64]; };

// recv_r[i] = ((send_0[i] + send_1[i]) + send_2[i]) ... in half, rank order (MergeTensors, inference_worker.cc:2197-2260)
__global__ void __launch_bounds__(256) k_local_allreduce_f16(const PtrTable t, int n, size_t count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    ifa::half_t acc = reinterpret_cast<const ifa::half_t *>(t.send[0])[i];
    for (int r = 1; r < n; r++) acc = ifa::f2h(ifa::h2f(acc) + ifa::h2f(reinterpret_cast<const ifa::half_t *>(t.send[r])[i]));
    for (int r = 0; r < n; r++) reinterpret_cast<ifa::half_t *>(t.recv[r])[i] = acc;
}

__global__ void __launch_bounds__(256) k_local_allgather(const PtrTable t, int n, size_t bytes)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bytes * (size_t)n) return;
    const int src = (int)(i / bytes);
    const uint8_t v = reinterpret_cast<const uint8_t *>(t.send[src])[i % bytes];
    for (int r = 0; r < n; r++) reinterpret_cast<uint8_t *>(t.recv[r])[i] = v;
}

// ---------------------------------------------------------------- one-shot all-reduce for a decode step's vectors
// A T = 1 step merges two [dim] vectors per layer (DistributeAndMergeTensors, inference_worker.cc:2148-2335): 64
// all-reduces of 8 KB per token on Llama-2-7B.  A ring collective is built for bandwidth; at this size the cost is
// latency, so ranks that can map each other's memory (one process driving several GPUs with peer access, or ranks
// sharing a device) exchange the vectors directly:
//   every rank PUSHES its vector into slot [epoch parity][my rank] of every rank's inbox (posted writes over xGMI),
//   then stores `epoch` into its flag word on every rank; a rank waits until its own flag words all show `epoch`,
//   adds the n slots of its inbox in RANK ORDER in half precision -- ((v0 + v1) + v2) ..., the MergeTensors order, the
//   same on every rank -- and writes the result.  One launch of one workgroup per rank, no host involvement: capturable.
// Two inbox halves alternate by epoch parity: a rank can only be one all-reduce ahead of the slowest (it needs that
// rank's flag of epoch e to finish e), so half (e & 1) is never rewritten while somebody still reads epoch e - 2 from it.
// Payload and flags travel as system-scope atomics (8-byte payload words: write-through stores, loads that bypass the
// caches), so no cache maintenance is assumed.  Every wait is bounded (ONESHOT_TIMEOUT_TICKS of the 100 MHz clock); a
// rank that gives up raises its status word (ifa_comm_status) and produces no result.
constexpr size_t ONESHOT_MAX_BYTES = 64 * 1024;            // per vector (dim <= 32768 halves)
constexpr long long ONESHOT_TIMEOUT_TICKS = 200000000;     // 2 s

struct OneShotPeers {
    unsigned long long *inbox[64];      // rank r's inbox: [2][n][ONESHOT_MAX_BYTES / 8] words, on r's device
    unsigned *flags[64];                // rank r's flags: [2][n] epochs
};

struct OneShot {
    int n = 0;
    OneShotPeers peers;                 // the same table on every rank
    std::vector<unsigned *> epoch;      // per rank: device counter of completed all-reduces
    std::vector<unsigned *> status;     // per rank: device word, non-zero = a wait gave up
    std::vector<int> device;
    bool ready = false;
    int ipc_self = -1;                  // >= 0: one process per rank -- only entry ipc_self is this process's allocation, the
                                        // other entries of `peers` are hipIpcOpenMemHandle mappings of the peers' buffers
    ~OneShot()
    {
        for (int r = 0; r < n; r++) {
            if (ipc_self >= 0 && r != ipc_self) {
                if (peers.inbox[r]) (void)hipIpcCloseMemHandle(peers.inbox[r]);
                if (peers.flags[r]) (void)hipIpcCloseMemHandle(peers.flags[r]);
                continue;
            }
            (void)hipSetDevice(device[(size_t)r]);
            if (peers.inbox[r]) (void)hipFree(peers.inbox[r]);
            if (peers.flags[r]) (void)hipFree(peers.flags[r]);
            if (epoch[(size_t)r]) (void)hipFree(epoch[(size_t)r]);
            if (status[(size_t)r]) (void)hipFree(status[(size_t)r]);
        }
    }
};

__global__ void __launch_bounds__(1024) k_oneshot_allreduce_f16(const OneShotPeers P, int me, int n, unsigned *epoch_ctr, unsigned *status,
                                                                const ifa::half_t *__restrict__ send, ifa::half_t *__restrict__ recv, size_t count)
{
    const unsigned e = *epoch_ctr + 1u;
    const unsigned par = e & 1u;
    const size_t words = (count + 3) / 4;                      // 8-byte words (4 halves); the tail word is zero-padded
    const size_t slot_words = ONESHOT_MAX_BYTES / 8;
    // push
    for (size_t i = threadIdx.x; i < words; i += blockDim.x) {
        unsigned long long w = 0;
        for (int k = 0; k < 4; k++) {
            const size_t idx = i * 4 + (size_t)k;
            const unsigned long long hb = idx < count ? (unsigned long long)__builtin_bit_cast(unsigned short, send[idx]) : 0ull;
            w |= hb << (16 * k);
        }
        for (int p = 0; p < n; p++)
            __hip_atomic_store(P.inbox[p] + ((size_t)par * n + (size_t)me) * slot_words + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if ((int)threadIdx.x < n) __hip_atomic_store(P.flags[threadIdx.x] + par * n + me, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // wait for every rank's vector of this epoch
    __shared__ int failed;
    if (threadIdx.x == 0) failed = 0;
    __syncthreads();
    if ((int)threadIdx.x < n) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(P.flags[me] + par * n + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > ONESHOT_TIMEOUT_TICKS) { failed = 1; break; }
        }
    }
    __syncthreads();
    if (failed) { if (threadIdx.x == 0) *status = e; return; }
    // sum in rank order, in half (MergeTensors, inference_worker.cc:2197-2260)
    for (size_t i = threadIdx.x; i < words; i += blockDim.x) {
        unsigned long long w = __hip_atomic_load(P.inbox[me] + ((size_t)par * n) * slot_words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ifa::half_t acc[4];
        for (int k = 0; k < 4; k++) acc[k] = __builtin_bit_cast(ifa::half_t, (unsigned short)(w >> (16 * k)));
        for (int r = 1; r < n; r++) {
            w = __hip_atomic_load(P.inbox[me] + ((size_t)par * n + (size_t)r) * slot_words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int k = 0; k < 4; k++)
                acc[k] = ifa::f2h(ifa::h2f(acc[k]) + ifa::h2f(__builtin_bit_cast(ifa::half_t, (unsigned short)(w >> (16 * k)))));
        }
        for (int k = 0; k < 4; k++) { const size_t idx = i * 4 + (size_t)k; if (idx < count) recv[idx] = acc[k]; }
    }
    if (threadIdx.x == 0) *epoch_ctr = e;
}

} // namespace

static std::atomic<unsigned long long> g_comm_serial{1};

struct ifa_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    unsigned long long serial = g_comm_serial++;      // identity of this communicator object (ifa_comm_serial)
    std::atomic<bool> aborted{false};
    bool no_oneshot = false;                // ifa_comm_set_oneshot(c, 0): keep RCCL / the rendezvous for every size
    bool no_rccl = false;                   // IFA_COMM_TEST_NO_RCCL: one-shot all-reduces only (tests)
    std::shared_ptr<LocalGroup> local;      // loopback group (ranks sharing a device); comm == nullptr then
    std::shared_ptr<OneShot> oneshot;       // peer-mapped exchange for small all-reduces (null: RCCL / rendezvous only)
};

#define IFA_NCCL_CHECK(expr)                                                                  \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess)                                                                \
            return ifa_fail(IFA_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(r_));     \
    } while (0)

extern "C" {

int ifa_comm_unique_id(void *id_out_128)
{
    IFA_REQUIRE(id_out_128, "ifa_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == IFA_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    IFA_NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(id_out_128, &id, sizeof(id));
    return IFA_OK;
}

int ifa_comm_init_rank(const void *id_128, int nranks, int rank, int device, ifa_comm **out)
{
    IFA_REQUIRE(id_128 && out, "ifa_comm_init_rank: null pointer");
    IFA_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "ifa_comm_init_rank: rank %d of %d", rank, nranks);
    IFA_HIP_CHECK(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id_128, sizeof(id));
    ifa_comm *c = new ifa_comm();
    c->rank = rank; c->nranks = nranks; c->device = device;
    if (getenv("IFA_COMM_TEST_NO_RCCL")) {      // tests of the cross-process one-shot exchange on ONE device (RCCL refuses two ranks on
        c->no_rccl = true;                      // a device): the communicator then only serves all-reduces that take the one-shot path
        *out = c;
        return IFA_OK;
    }
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return ifa_fail(IFA_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, nranks, device, ncclGetErrorString(r)); }
    *out = c;
    return IFA_OK;
}

// One process per rank (ifa_comm_init_rank): the one-shot exchange needs the peers' inboxes and flags mapped into this process.
// Inboxes, flags and counters are FINE-GRAINED device memory: kernels on different GPUs signal each other through them while
// they run, and the HSA memory model only promises cross-agent visibility of coarse-grained (plain hipMalloc) memory at
// dispatch boundaries -- a fresh flag next to a stale payload would be a silently wrong sum.
static hipError_t oneshot_alloc(void **p, size_t bytes)
{
    return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
}

// In-kernel visibility probe between two devices: the reader spins (bounded) on a word of ITS OWN device's flag area while the
// writer, a kernel running on the OTHER device at the same time, stores the tag with system scope.  A host-side hipMemcpy
// round trip (round 3's self-test) says nothing about what a running kernel sees.
static __global__ void k_oneshot_probe_wait(const unsigned *word, unsigned tag, unsigned *seen, long long timeout_ticks)
{
    const long long t_end = wall_clock64() + timeout_ticks;
    unsigned v = 0;
    while ((v = __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) != tag && wall_clock64() < t_end) __builtin_amdgcn_s_sleep(8);
    *seen = v == tag ? 1u : 0u;
}
static __global__ void k_oneshot_probe_post(unsigned *word, unsigned tag)
{
    __hip_atomic_store(word, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// export: allocate this rank's buffers and hand out their IPC handles (2 x 64 bytes); the caller gathers the handles of all ranks
// over whatever channel carried the communicator id; import: map them (hipIpcOpenMemHandle) -- from then on all-reduces of
// <= 64 KB take the one-shot kernel.  Nothing here proves cross-device visibility: the first exchanges have bounded waits
// (ifa_comm_status), and callers that can should compare a few steps against the RCCL path before relying on it (bench.py does).
int ifa_comm_oneshot_export(ifa_comm *c, void *handle_out_128)
{
    IFA_REQUIRE(c && handle_out_128 && !c->local, "ifa_comm_oneshot_export: a communicator made by ifa_comm_init_rank is needed");
    IFA_REQUIRE(c->nranks >= 2 && c->nranks <= 64, "ifa_comm_oneshot_export: %d ranks", c->nranks);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
    IFA_HIP_CHECK(hipSetDevice(c->device));
    auto os = std::make_shared<OneShot>();
    const int n = c->nranks;
    os->n = n; os->ipc_self = c->rank;
    os->epoch.assign((size_t)n, nullptr); os->status.assign((size_t)n, nullptr); os->device.assign((size_t)n, c->device);
    memset(&os->peers, 0, sizeof(os->peers));
    const size_t inbox_bytes = 2 * (size_t)n * ONESHOT_MAX_BYTES, flag_bytes = 2 * (size_t)n * sizeof(unsigned);
    IFA_HIP_CHECK(oneshot_alloc((void **)&os->peers.inbox[c->rank], inbox_bytes));
    IFA_HIP_CHECK(oneshot_alloc((void **)&os->peers.flags[c->rank], flag_bytes + 64));
    IFA_HIP_CHECK(oneshot_alloc((void **)&os->epoch[(size_t)c->rank], 4));
    IFA_HIP_CHECK(oneshot_alloc((void **)&os->status[(size_t)c->rank], 4));
    IFA_HIP_CHECK(hipMemset(os->peers.flags[c->rank], 0, flag_bytes + 64));
    IFA_HIP_CHECK(hipMemset(os->epoch[(size_t)c->rank], 0, 4));
    IFA_HIP_CHECK(hipMemset(os->status[(size_t)c->rank], 0, 4));
    hipIpcMemHandle_t h[2];
    IFA_HIP_CHECK(hipIpcGetMemHandle(&h[0], os->peers.inbox[c->rank]));
    IFA_HIP_CHECK(hipIpcGetMemHandle(&h[1], os->peers.flags[c->rank]));
    memcpy(handle_out_128, h, sizeof(h));
    c->oneshot = os;                   // not ready until the peers are mapped
    return IFA_OK;
}

int ifa_comm_oneshot_import(ifa_comm *c, const void *handles_all)
{
    IFA_REQUIRE(c && handles_all && c->oneshot && c->oneshot->ipc_self == c->rank && !c->oneshot->ready, "ifa_comm_oneshot_import: call ifa_comm_oneshot_export first");
    IFA_HIP_CHECK(hipSetDevice(c->device));
    OneShot &os = *c->oneshot;
    const hipIpcMemHandle_t *h = reinterpret_cast<const hipIpcMemHandle_t *>(handles_all);
    for (int r = 0; r < os.n; r++) {
        if (r == c->rank) continue;
        void *pi = nullptr, *pf = nullptr;
        IFA_HIP_CHECK(hipIpcOpenMemHandle(&pi, h[2 * r], hipIpcMemLazyEnablePeerAccess));
        os.peers.inbox[r] = (unsigned long long *)pi;
        IFA_HIP_CHECK(hipIpcOpenMemHandle(&pf, h[2 * r + 1], hipIpcMemLazyEnablePeerAccess));
        os.peers.flags[r] = (unsigned *)pf;
    }
    os.ready = true;
    return IFA_OK;
}

// Inboxes, flags and counters on every rank's device, peer access between the devices, and a visibility self-test: every
// rank's words must be writable AND readable from every device within 100 ms, otherwise the group keeps RCCL (or the
// rendezvous) for every size -- the one-shot path can be absent, never wrong.
static std::shared_ptr<OneShot> oneshot_setup(const int *device_ids, int n)
{
    if (n < 2 || getenv("IFA_NO_ONESHOT")) return nullptr;
    auto os = std::make_shared<OneShot>();
    os->n = n; os->epoch.assign((size_t)n, nullptr); os->status.assign((size_t)n, nullptr); os->device.assign(device_ids, device_ids + n);
    memset(&os->peers, 0, sizeof(os->peers));
    int prev = 0;
    (void)hipGetDevice(&prev);
    bool ok = true;
    for (int a = 0; a < n && ok; a++)
        for (int b = 0; b < n && ok; b++) {
            if (device_ids[a] == device_ids[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) != hipSuccess || !can) { ok = false; break; }
            (void)hipSetDevice(device_ids[a]);
            const hipError_t e = hipDeviceEnablePeerAccess(device_ids[b], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
            (void)hipGetLastError();
        }
    const size_t inbox_bytes = 2 * (size_t)n * ONESHOT_MAX_BYTES, flag_bytes = 2 * (size_t)n * sizeof(unsigned);
    for (int r = 0; r < n && ok; r++) {
        ok = hipSetDevice(device_ids[r]) == hipSuccess
            && oneshot_alloc((void **)&os->peers.inbox[r], inbox_bytes) == hipSuccess && oneshot_alloc((void **)&os->peers.flags[r], flag_bytes + 64) == hipSuccess
            && oneshot_alloc((void **)&os->epoch[(size_t)r], 4) == hipSuccess && oneshot_alloc((void **)&os->status[(size_t)r], 4) == hipSuccess
            && hipMemset(os->peers.flags[r], 0, flag_bytes + 64) == hipSuccess && hipMemset(os->epoch[(size_t)r], 0, 4) == hipSuccess
            && hipMemset(os->status[(size_t)r], 0, 4) == hipSuccess;
    }
    // self-test between every pair of DIFFERENT devices: a kernel on b's device waits (<= 100 ms) for the word a kernel on a's
    // device posts into b's spare flag area while it runs (ranks on one device share its L2: nothing to prove there)
    if (ok) {
        unsigned *seen = nullptr;
        for (int a = 0; a < n && ok; a++) {
            for (int b = 0; b < n && ok; b++) {
                if (device_ids[a] == device_ids[b]) continue;
                const unsigned tag = 0xA5000000u | (unsigned)(a * 64 + b);
                unsigned *word = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(os->peers.flags[b]) + flag_bytes);
                ok = hipSetDevice(device_ids[b]) == hipSuccess && oneshot_alloc((void **)&seen, 4) == hipSuccess && hipMemset(seen, 0, 4) == hipSuccess;
                if (!ok) break;
                k_oneshot_probe_wait<<<1, 1>>>(word, tag, seen, 10000000LL);                 // the reader first: it is running when the post arrives
                ok = hipGetLastError() == hipSuccess && hipSetDevice(device_ids[a]) == hipSuccess;
                if (ok) { k_oneshot_probe_post<<<1, 1>>>(word, tag); ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess; }
                unsigned got = 0;
                ok = ok && hipSetDevice(device_ids[b]) == hipSuccess && hipDeviceSynchronize() == hipSuccess
                    && hipMemcpy(&got, seen, 4, hipMemcpyDeviceToHost) == hipSuccess && got == 1u;
                (void)hipSetDevice(device_ids[b]);
                if (seen) { (void)hipFree(seen); seen = nullptr; }
            }
        }
        for (int r = 0; r < n && ok; r++) { (void)hipSetDevice(device_ids[r]); ok = hipMemset(reinterpret_cast<char *>(os->peers.flags[r]) + flag_bytes, 0, 64) == hipSuccess; }
    }
    (void)hipSetDevice(prev);
    (void)hipGetLastError();
    os->ready = ok;
    return ok ? os : nullptr;
}

int ifa_comm_init_all(const int *device_ids, int n, ifa_comm **comms_out)
{
    IFA_REQUIRE(device_ids && comms_out && n >= 1 && n <= 64, "ifa_comm_init_all: n %d", n);
    bool dup = false;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) dup = dup || device_ids[i] == device_ids[j];
    if (dup) {          // ranks sharing a device: in-process loopback group (see the header of this file)
        for (int i = 1; i < n; i++) IFA_REQUIRE(device_ids[i] == device_ids[0], "ifa_comm_init_all: a loopback group lives on ONE device");
        auto grp = std::make_shared<LocalGroup>();
        grp->n = n;
        std::shared_ptr<OneShot> os = oneshot_setup(device_ids, n);
        for (int i = 0; i < n; i++) {
            ifa_comm *c = new ifa_comm();
            c->rank = i; c->nranks = n; c->device = device_ids[i]; c->local = grp; c->oneshot = os;
            // Ranks that share a device are NOT guaranteed to run at the same time: their streams may sit on the same
            // hardware queue (the runtime has a handful), and a kernel that spins for a peer queued behind it never sees
            // that peer.  The exchange is therefore off by default for loopback groups (tests switch it on for <= 4 ranks
            // with ifa_comm_set_oneshot to check its arithmetic); one rank per GPU -- what it is for -- has no such limit.
            c->no_oneshot = true;
            comms_out[i] = c;
        }
        return IFA_OK;
    }
    std::vector<ncclComm_t> cs((size_t)n);
    IFA_NCCL_CHECK(ncclCommInitAll(cs.data(), n, device_ids));
    std::shared_ptr<OneShot> os = oneshot_setup(device_ids, n);
    for (int i = 0; i < n; i++) {
        ifa_comm *c = new ifa_comm();
        c->comm = cs[(size_t)i]; c->rank = i; c->nranks = n; c->device = device_ids[i]; c->oneshot = os;
        // The peer exchange has not run between two real devices yet (no multi-GPU box in four rounds): RCCL carries every
        // size until a caller switches it on (ifa_comm_set_oneshot) after checking a few steps against RCCL, as bench.py does
        c->no_oneshot = true;
        comms_out[i] = c;
    }
    return IFA_OK;
}

int ifa_comm_destroy(ifa_comm *c)
{
    if (!c) return IFA_OK;
    if (c->comm && !c->aborted.load()) (void)ncclCommDestroy(c->comm);
    delete c;
    return IFA_OK;
}

int ifa_comm_capturable(const ifa_comm *c) { return (c && c->local) ? 0 : 1; }
unsigned long long ifa_comm_serial(const ifa_comm *c) { return c ? c->serial : 0ull; }

// 1 if small all-reduces of this communicator take the one-shot peer exchange
int ifa_comm_oneshot(const ifa_comm *c) { return (c && c->oneshot && c->oneshot->ready && !c->no_oneshot) ? 1 : 0; }
int ifa_comm_set_oneshot(ifa_comm *c, int on) { if (c) c->no_oneshot = !on; return IFA_OK; }
// non-zero: a wait inside a one-shot all-reduce of this rank gave up (synchronises the device; call it after the stream
// was synchronised, e.g. at the end of a decode call)
int ifa_comm_status(ifa_comm *c)
{
    if (!c || !c->oneshot) return 0;
    unsigned v = 0;
    if (hipMemcpy(&v, c->oneshot->status[(size_t)c->rank], 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}

// A rank that fails between two collectives leaves its peers blocked in the next one.  Abort wakes them: RCCL
// communicators are aborted (ncclCommAbort: outstanding and later operations fail), loopback groups raise a flag that
// every rendezvous checks.  The communicator is unusable afterwards; destroy it.
int ifa_comm_abort(ifa_comm *c)
{
    if (!c) return IFA_OK;
    if (c->local) c->local->abort();
    else if (c->comm && !c->aborted.exchange(true)) (void)ncclCommAbort(c->comm);      // (frees the communicator: not destroyed again)
    return IFA_OK;
}
int ifa_comm_rank(const ifa_comm *c) { return c ? c->rank : -1; }
// ranks of the communicator: what RCCL itself counts for RCCL communicators (ncclCommCount -- bench.py reports it as ranks_seen),
// the constructor's figure for in-process loopback groups
int ifa_comm_size(const ifa_comm *c)
{
    if (!c) return 0;
    int n = 0;
    if (c->comm && !c->aborted.load() && ncclCommCount(c->comm, &n) == ncclSuccess && n > 0) return n;
    return c->nranks;
}

// one host thread issuing the calls of several ranks (ifa_comm_init_all communicators) brackets them with these
int ifa_comm_group_start(void) { IFA_NCCL_CHECK(ncclGroupStart()); return IFA_OK; }
int ifa_comm_group_end(void) { IFA_NCCL_CHECK(ncclGroupEnd()); return IFA_OK; }

int ifa_allreduce_sum_f16(ifa_comm *c, const void *send_f16, void *recv_f16, size_t count, ifa_stream stream)
{
    IFA_REQUIRE(c && (c->comm || c->local || c->no_rccl) && send_f16 && recv_f16, "ifa_allreduce_sum_f16: null pointer");
    if (count == 0) return IFA_OK;
    if (c->oneshot && c->oneshot->ready && count * 2 <= ONESHOT_MAX_BYTES && !c->no_oneshot) {
        OneShot &os = *c->oneshot;
        k_oneshot_allreduce_f16<<<dim3(1), dim3(1024), 0, ifa_s(stream)>>>(os.peers, c->rank, os.n, os.epoch[(size_t)c->rank], os.status[(size_t)c->rank],
                                                                          (const ifa::half_t *)send_f16, (ifa::half_t *)recv_f16, count);
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    if (c->local) {
        LocalGroup &g = *c->local;
        IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(stream)));
        { std::lock_guard<std::mutex> lk(g.mu); g.sendp[c->rank] = send_f16; g.recvp[c->rank] = recv_f16; }
        if (!g.barrier()) return ifa_fail(IFA_ERR_STATE, "ifa_allreduce_sum_f16: the group was aborted (a peer failed)");
        if (c->rank == 0) {
            PtrTable t; memset(&t, 0, sizeof(t));
            for (int r = 0; r < g.n; r++) { t.send[r] = g.sendp[r]; t.recv[r] = g.recvp[r]; }
            k_local_allreduce_f16<<<dim3(ifa_cdiv(count, 256)), dim3(256), 0, ifa_s(stream)>>>(t, g.n, count);
            IFA_LAUNCH_CHECK();
            IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(stream)));
        }
        if (!g.barrier()) return ifa_fail(IFA_ERR_STATE, "ifa_allreduce_sum_f16: the group was aborted (a peer failed)");
        return IFA_OK;
    }
    IFA_REQUIRE(c->comm, "ifa_allreduce_sum_f16: this communicator has no RCCL side (IFA_COMM_TEST_NO_RCCL) and the one-shot path does not apply (%zu halves)", count);
    IFA_NCCL_CHECK(ncclAllReduce(send_f16, recv_f16, count, ncclFloat16, ncclSum, c->comm, ifa_s(stream)));
    return IFA_OK;
}

int ifa_allgather(ifa_comm *c, const void *send, void *recv, size_t bytes_per_rank, ifa_stream stream)
{
    IFA_REQUIRE(c && (c->comm || c->local) && send && recv, "ifa_allgather: null pointer");
    if (bytes_per_rank == 0) return IFA_OK;
    if (c->local) {
        LocalGroup &g = *c->local;
        IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(stream)));
        { std::lock_guard<std::mutex> lk(g.mu); g.sendp[c->rank] = send; g.recvp[c->rank] = recv; }
        if (!g.barrier()) return ifa_fail(IFA_ERR_STATE, "ifa_allgather: the group was aborted (a peer failed)");
        if (c->rank == 0) {
            PtrTable t; memset(&t, 0, sizeof(t));
            for (int r = 0; r < g.n; r++) { t.send[r] = g.sendp[r]; t.recv[r] = g.recvp[r]; }
            k_local_allgather<<<dim3(ifa_cdiv(bytes_per_rank * (size_t)g.n, 256)), dim3(256), 0, ifa_s(stream)>>>(t, g.n, bytes_per_rank);
            IFA_LAUNCH_CHECK();
            IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(stream)));
        }
        if (!g.barrier()) return ifa_fail(IFA_ERR_STATE, "ifa_allgather: the group was aborted (a peer failed)");
        return IFA_OK;
    }
    IFA_NCCL_CHECK(ncclAllGather(send, recv, bytes_per_rank, ncclInt8, c->comm, ifa_s(stream)));
    return IFA_OK;
}

int ifa_broadcast(ifa_comm *c, void *buf, size_t bytes, int root, ifa_stream stream)
{
    IFA_REQUIRE(c && (c->comm || c->local) && buf, "ifa_broadcast: null pointer");
    IFA_REQUIRE(root >= 0 && root < c->nranks, "ifa_broadcast: root %d of %d", root, c->nranks);
    if (bytes == 0) return IFA_OK;
    if (c->local) {
        LocalGroup &g = *c->local;
        IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(stream)));
        { std::lock_guard<std::mutex> lk(g.mu); g.recvp[c->rank] = buf; }
        if (!g.barrier()) return ifa_fail(IFA_ERR_STATE, "ifa_broadcast: the group was aborted (a peer failed)");
        if (c->rank != root) {
            IFA_HIP_CHECK(hipMemcpyAsync(buf, g.recvp[root], bytes, hipMemcpyDeviceToDevice, ifa_s(stream)));
            IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(stream)));
        }
        if (!g.barrier()) return ifa_fail(IFA_ERR_STATE, "ifa_broadcast: the group was aborted (a peer failed)");
        return IFA_OK;
    }
    IFA_NCCL_CHECK(ncclBroadcast(buf, buf, bytes, ncclInt8, root, c->comm, ifa_s(stream)));
    return IFA_OK;
}

int ifa_send(ifa_comm *c, const void *buf, size_t bytes, int peer, ifa_stream stream)
{
    IFA_REQUIRE(c && (c->comm || c->local) && buf, "ifa_send: null pointer");
    IFA_REQUIRE(peer >= 0 && peer < c->nranks && peer != c->rank, "ifa_send: peer %d (rank %d of %d)", peer, c->rank, c->nranks);
    if (c->local) {      // rendezvous: the receiver copies, then releases the slot
        LocalGroup &g = *c->local;
        IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(stream)));
        std::unique_lock<std::mutex> lk(g.mu);
        LocalGroup::Slot &sl = g.slots[c->rank][peer];
        g.cv.wait(lk, [&] { return !sl.full || g.aborted; });
        if (g.aborted) return ifa_fail(IFA_ERR_STATE, "ifa_send: the group was aborted (a peer failed)");
        sl.src = buf; sl.bytes = bytes; sl.full = true; sl.refused = false;
        g.cv.notify_all();
        g.cv.wait(lk, [&] { return !sl.full || g.aborted; });
        if (g.aborted) { sl.full = false; return ifa_fail(IFA_ERR_STATE, "ifa_send: the group was aborted (a peer failed)"); }
        if (sl.refused) return ifa_fail(IFA_ERR_ARG, "ifa_send: the receiver expected another size than %zu bytes", bytes);
        return IFA_OK;
    }
    IFA_NCCL_CHECK(ncclSend(buf, bytes, ncclInt8, peer, c->comm, ifa_s(stream)));
    return IFA_OK;
}

int ifa_recv(ifa_comm *c, void *buf, size_t bytes, int peer, ifa_stream stream)
{
    IFA_REQUIRE(c && (c->comm || c->local) && buf, "ifa_recv: null pointer");
    IFA_REQUIRE(peer >= 0 && peer < c->nranks && peer != c->rank, "ifa_recv: peer %d (rank %d of %d)", peer, c->rank, c->nranks);
    if (c->local) {
        LocalGroup &g = *c->local;
        const void *src = nullptr;
        {
            std::unique_lock<std::mutex> lk(g.mu);
            LocalGroup::Slot &sl = g.slots[peer][c->rank];
            g.cv.wait(lk, [&] { return sl.full || g.aborted; });
            if (g.aborted) return ifa_fail(IFA_ERR_STATE, "ifa_recv: the group was aborted (a peer failed)");
            if (sl.bytes != bytes) {      // release the sender too (it would wait for the slot forever) and tell it why
                const size_t got = sl.bytes;
                sl.full = false; sl.refused = true;
                g.cv.notify_all();
                return ifa_fail(IFA_ERR_ARG, "ifa_recv: %zu bytes expected, the peer sends %zu", bytes, got);
            }
            src = sl.src;
        }
        IFA_HIP_CHECK(hipMemcpyAsync(buf, src, bytes, hipMemcpyDeviceToDevice, ifa_s(stream)));
        IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(stream)));
        { std::lock_guard<std::mutex> lk(g.mu); g.slots[peer][c->rank].full = false; }
        g.cv.notify_all();
        return IFA_OK;
    }
    IFA_NCCL_CHECK(ncclRecv(buf, bytes, ncclInt8, peer, c->comm, ifa_s(stream)));
    return IFA_OK;
}

} // extern "C"
