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

} // namespace

static std::atomic<unsigned long long> g_comm_serial{1};

struct ifa_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    unsigned long long serial = g_comm_serial++;      // identity of this communicator object (ifa_comm_serial)
    std::atomic<bool> aborted{false};
    std::shared_ptr<LocalGroup> local;      // loopback group (ranks sharing a device); comm == nullptr then
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
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return ifa_fail(IFA_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, nranks, device, ncclGetErrorString(r)); }
    *out = c;
    return IFA_OK;
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
        for (int i = 0; i < n; i++) {
            ifa_comm *c = new ifa_comm();
            c->rank = i; c->nranks = n; c->device = device_ids[i]; c->local = grp;
            comms_out[i] = c;
        }
        return IFA_OK;
    }
    std::vector<ncclComm_t> cs((size_t)n);
    IFA_NCCL_CHECK(ncclCommInitAll(cs.data(), n, device_ids));
    for (int i = 0; i < n; i++) {
        ifa_comm *c = new ifa_comm();
        c->comm = cs[(size_t)i]; c->rank = i; c->nranks = n; c->device = device_ids[i];
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
int ifa_comm_size(const ifa_comm *c) { return c ? c->nranks : 0; }

// one host thread issuing the calls of several ranks (ifa_comm_init_all communicators) brackets them with these
int ifa_comm_group_start(void) { IFA_NCCL_CHECK(ncclGroupStart()); return IFA_OK; }
int ifa_comm_group_end(void) { IFA_NCCL_CHECK(ncclGroupEnd()); return IFA_OK; }

int ifa_allreduce_sum_f16(ifa_comm *c, const void *send_f16, void *recv_f16, size_t count, ifa_stream stream)
{
    IFA_REQUIRE(c && (c->comm || c->local) && send_f16 && recv_f16, "ifa_allreduce_sum_f16: null pointer");
    if (count == 0) return IFA_OK;
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
