/*
 * mock_rccl.cpp -- TEST ONLY.  An in-process stand-in for librccl: the "ranks" are threads of one process that share
 * one GPU.  It implements exactly the entry points libmedpyhip resolves (MgcRccl, mgc_kernels.hip) with NCCL's
 * contract -- FIFO point-to-point channels per (source, destination), sizes of a Send and its Recv must match,
 * grouped calls complete at ncclGroupEnd, ncclAllReduce over int64 -- so the multi-rank protocol of mgc_halo_exchange /
 * mgc_allreduce_counts (two-stage compacted records, who sends what to whom, in which order) runs on a 1-GPU box.
 * Real RCCL refuses two ranks on one device, hence the mock; it is selected with MEDPY_HIP_RCCL=<this .so>.
 *
 * Stream semantics are made synchronous (hipStreamSynchronize + blocking copies): functionally what the
 * stream-ordered real calls do.  Data moves device -> host mailbox -> device.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string.h>
#include <utility>
#include <vector>

namespace {

struct World {
    std::mutex mu;
    std::condition_variable cv;
    int nranks = 0, joined = 0;
    std::map<std::pair<int, int>, std::deque<std::vector<char>>> box; /* (src, dst) -> messages in order */
    /* all-reduce rendezvous */
    int arrived = 0, generation = 0;
    std::vector<int64_t> acc, result;
};

struct Comm {
    World* w;
    int rank;
};

struct Op {
    bool send;
    const void* sbuf;
    void* rbuf;
    size_t bytes;
    int peer;
    Comm* c;
    hipStream_t stream;
};

std::mutex g_mu;
std::map<std::vector<char>, World*> g_worlds;
int g_next_id = 1;
thread_local int t_group = 0;
thread_local std::vector<Op> t_ops;

size_t dtype_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

ncclResult_t do_send(const Op& o)
{
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> msg(o.bytes);
    if (o.bytes && hipMemcpy(msg.data(), o.sbuf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    World* w = o.c->w;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->box[{o.c->rank, o.peer}].push_back(std::move(msg));
    }
    w->cv.notify_all();
    return ncclSuccess;
}

ncclResult_t do_recv(const Op& o)
{
    World* w = o.c->w;
    std::vector<char> msg;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        auto& q = w->box[{o.peer, o.c->rank}];
        w->cv.wait(lk, [&] { return !q.empty(); });
        msg = std::move(q.front());
        q.pop_front();
    }
    if (msg.size() != o.bytes) return ncclInvalidArgument; /* NCCL: undefined behaviour / hang; here: a loud error */
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpy(o.rbuf, msg.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    std::lock_guard<std::mutex> lk(g_mu);
    const int v = g_next_id++;
    memcpy(id->internal, &v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::vector<char> key(id.internal, id.internal + sizeof(id.internal));
    World* w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        World*& slot = g_worlds[key];
        if (!slot) { slot = new World(); slot->nranks = nranks; }
        w = slot;
    }
    if (w->nranks != nranks) return ncclInvalidArgument;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->joined++;
        w->cv.notify_all();
        w->cv.wait(lk, [&] { return w->joined >= w->nranks; });
    }
    *comm = (ncclComm_t) new Comm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete (Comm*)comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    t_group++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_group <= 0) return ncclInvalidUsage;
    if (--t_group > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    for (const Op& o : ops)
        if (o.send) { const ncclResult_t r = do_send(o); if (r != ncclSuccess) return r; }
    for (const Op& o : ops)
        if (!o.send) { const ncclResult_t r = do_recv(o); if (r != ncclSuccess) return r; }
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->w->nranks || peer == c->rank) return ncclInvalidArgument;
    Op o{true, buf, nullptr, count * dtype_size(dt), peer, c, stream};
    if (t_group) { t_ops.push_back(o); return ncclSuccess; }
    return do_send(o);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->w->nranks || peer == c->rank) return ncclInvalidArgument;
    Op o{false, nullptr, buf, count * dtype_size(dt), peer, c, stream};
    if (t_group) { t_ops.push_back(o); return ncclSuccess; }
    return do_recv(o);
}

ncclResult_t ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = (Comm*)comm;
    if (!c || dt != ncclInt64 || (op != ncclSum && op != ncclMin)) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<int64_t> mine(count);
    if (hipMemcpy(mine.data(), sendbuf, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    World* w = c->w;
    std::vector<int64_t> res;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        if (w->arrived == 0) w->acc = mine;
        else {
            if (w->acc.size() != count) return ncclInvalidArgument;
            for (size_t i = 0; i < count; ++i) w->acc[i] = op == ncclMin ? (mine[i] < w->acc[i] ? mine[i] : w->acc[i]) : w->acc[i] + mine[i];
        }
        const int gen = w->generation;
        if (++w->arrived == w->nranks) {
            w->result = w->acc;
            w->arrived = 0;
            w->generation++;
            w->cv.notify_all();
        } else {
            w->cv.wait(lk, [&] { return w->generation != gen; });
        }
        res = w->result;
    }
    if (hipMemcpy(recvbuf, res.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "success";
    case ncclInvalidArgument: return "invalid argument (mock: size mismatch between a Send and its Recv, or bad peer)";
    case ncclInvalidUsage: return "invalid usage";
    case ncclUnhandledCudaError: return "HIP error inside the mock";
    default: return "mock rccl error";
    }
}

} /* extern "C" */
