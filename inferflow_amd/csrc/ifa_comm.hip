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
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "ifa_host.h"

struct ifa_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
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
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
    return IFA_OK;
}

int ifa_comm_rank(const ifa_comm *c) { return c ? c->rank : -1; }
int ifa_comm_size(const ifa_comm *c) { return c ? c->nranks : 0; }

// one host thread issuing the calls of several ranks (ifa_comm_init_all communicators) brackets them with these
int ifa_comm_group_start(void) { IFA_NCCL_CHECK(ncclGroupStart()); return IFA_OK; }
int ifa_comm_group_end(void) { IFA_NCCL_CHECK(ncclGroupEnd()); return IFA_OK; }

int ifa_allreduce_sum_f16(ifa_comm *c, const void *send_f16, void *recv_f16, size_t count, ifa_stream stream)
{
    IFA_REQUIRE(c && c->comm && send_f16 && recv_f16, "ifa_allreduce_sum_f16: null pointer");
    if (count == 0) return IFA_OK;
    IFA_NCCL_CHECK(ncclAllReduce(send_f16, recv_f16, count, ncclFloat16, ncclSum, c->comm, ifa_s(stream)));
    return IFA_OK;
}

int ifa_allgather(ifa_comm *c, const void *send, void *recv, size_t bytes_per_rank, ifa_stream stream)
{
    IFA_REQUIRE(c && c->comm && send && recv, "ifa_allgather: null pointer");
    if (bytes_per_rank == 0) return IFA_OK;
    IFA_NCCL_CHECK(ncclAllGather(send, recv, bytes_per_rank, ncclInt8, c->comm, ifa_s(stream)));
    return IFA_OK;
}

int ifa_broadcast(ifa_comm *c, void *buf, size_t bytes, int root, ifa_stream stream)
{
    IFA_REQUIRE(c && c->comm && buf, "ifa_broadcast: null pointer");
    IFA_REQUIRE(root >= 0 && root < c->nranks, "ifa_broadcast: root %d of %d", root, c->nranks);
    if (bytes == 0) return IFA_OK;
    IFA_NCCL_CHECK(ncclBroadcast(buf, buf, bytes, ncclInt8, root, c->comm, ifa_s(stream)));
    return IFA_OK;
}

int ifa_send(ifa_comm *c, const void *buf, size_t bytes, int peer, ifa_stream stream)
{
    IFA_REQUIRE(c && c->comm && buf, "ifa_send: null pointer");
    IFA_REQUIRE(peer >= 0 && peer < c->nranks && peer != c->rank, "ifa_send: peer %d (rank %d of %d)", peer, c->rank, c->nranks);
    IFA_NCCL_CHECK(ncclSend(buf, bytes, ncclInt8, peer, c->comm, ifa_s(stream)));
    return IFA_OK;
}

int ifa_recv(ifa_comm *c, void *buf, size_t bytes, int peer, ifa_stream stream)
{
    IFA_REQUIRE(c && c->comm && buf, "ifa_recv: null pointer");
    IFA_REQUIRE(peer >= 0 && peer < c->nranks && peer != c->rank, "ifa_recv: peer %d (rank %d of %d)", peer, c->rank, c->nranks);
    IFA_NCCL_CHECK(ncclRecv(buf, bytes, ncclInt8, peer, c->comm, ifa_s(stream)));
    return IFA_OK;
}

} // extern "C"
