// TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl.so that lets SEVERAL ranks share ONE GPU.
// RCCL itself refuses two ranks on one device, so on the single-GPU boxes of this pool the library's own collective path
// (bdof_comm_create -> ncclCommInitRank, bdof_reduce_scatter_grad / bdof_allgather_volume in place at rank * count offsets,
// tickets on the communicator's stream; csrc/bdof_capi.hip, csrc/bdof_comm.h) never ran with nranks > 1.  This file implements
// the eight entry points the library binds, with NCCL's documented semantics, over a POSIX shared-memory segment named by the
// unique id: every collective drains the stream it was given, stages through the host and sums in rank order.  Loaded through
// BDOF_RCCL_LIB (csrc/bdof_comm.h); tests/test_gpu_dist.py builds it with hipcc.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {
struct Header {
    std::atomic<int> arrive, gen, attached;
};
struct StubComm {
    int nranks = 0, rank = 0;
    Header* hdr = nullptr;
    char* data = nullptr;        // nranks slots of `slot` bytes
    size_t slot = 0, total = 0;
    std::string name;
    std::vector<char> host;
};
constexpr size_t kHeader = 4096;

bool barrier(StubComm* c) {
    const int g = c->hdr->gen.load();
    if (c->hdr->arrive.fetch_add(1) + 1 == c->nranks) {
        c->hdr->arrive.store(0);
        c->hdr->gen.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->gen.load() == g) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;      // a rank died: fail, do not hang
    }
    return true;
}
char* slot_of(StubComm* c, int r) { return c->data + (size_t)r * c->slot; }
size_t elem(ncclDataType_t t) { return t == ncclFloat64 || t == ncclInt64 || t == ncclUint64 ? 8 : (t == ncclFloat16 || t == ncclBfloat16 ? 2 : (t == ncclInt8 || t == ncclUint8 ? 1 : 4)); }
#define HIPS(x) do { if ((x) != hipSuccess) return ncclUnhandledCudaError; } while (0)
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "bdofstub_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    StubComm* c = new StubComm;
    c->nranks = nranks;
    c->rank = rank;
    id.internal[sizeof(id.internal) - 1] = 0;
    c->name = std::string("/") + id.internal;
    const char* mb = getenv("BDOF_STUB_SLOT_MB");
    c->slot = (size_t)(mb ? atoi(mb) : 64) << 20;
    c->total = kHeader + c->slot * nranks;
    const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->total) != 0) { delete c; return ncclSystemError; }
    void* p = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->hdr = (Header*)p;               // a fresh segment is zero-filled: the counters start at 0
    c->data = (char*)p + kHeader;
    c->hdr->attached.fetch_add(1);
    if (!barrier(c)) return ncclSystemError;
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    StubComm* c = (StubComm*)comm;
    if (!c) return ncclSuccess;
    const bool last = c->hdr->attached.fetch_sub(1) == 1;
    munmap((void*)c->hdr, c->total);
    if (last) shm_unlink(c->name.c_str());
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
    StubComm* c = (StubComm*)comm;
    if (t != ncclFloat || op != ncclSum) return ncclInvalidArgument;
    const size_t bytes = count * 4;
    if (bytes > c->slot) return ncclInvalidArgument;
    HIPS(hipStreamSynchronize(s));
    HIPS(hipMemcpy(slot_of(c, c->rank), send, bytes, hipMemcpyDeviceToHost));
    if (!barrier(c)) return ncclSystemError;
    c->host.resize(bytes);
    float* o = (float*)c->host.data();
    for (size_t i = 0; i < count; ++i) {
        float v = ((const float*)slot_of(c, 0))[i];
        for (int r = 1; r < c->nranks; ++r) v += ((const float*)slot_of(c, r))[i];
        o[i] = v;
    }
    if (!barrier(c)) return ncclSystemError;
    HIPS(hipMemcpy(recv, o, bytes, hipMemcpyHostToDevice));
    return ncclSuccess;
}

// recv (recvcount elements) = sum over the ranks of their send[rank * recvcount ...]; send holds recvcount * nranks elements
ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
    StubComm* c = (StubComm*)comm;
    if (t != ncclFloat || op != ncclSum) return ncclInvalidArgument;
    const size_t bytes = recvcount * 4 * c->nranks;
    if (bytes > c->slot) return ncclInvalidArgument;
    HIPS(hipStreamSynchronize(s));
    HIPS(hipMemcpy(slot_of(c, c->rank), send, bytes, hipMemcpyDeviceToHost));
    if (!barrier(c)) return ncclSystemError;
    c->host.resize(recvcount * 4);
    float* o = (float*)c->host.data();
    for (size_t i = 0; i < recvcount; ++i) {
        float v = ((const float*)slot_of(c, 0))[(size_t)c->rank * recvcount + i];
        for (int r = 1; r < c->nranks; ++r) v += ((const float*)slot_of(c, r))[(size_t)c->rank * recvcount + i];
        o[i] = v;
    }
    if (!barrier(c)) return ncclSystemError;
    HIPS(hipMemcpy(recv, o, recvcount * 4, hipMemcpyHostToDevice));
    return ncclSuccess;
}

// recv (sendcount * nranks elements) = the ranks' send buffers in rank order
ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
    StubComm* c = (StubComm*)comm;
    const size_t bytes = sendcount * elem(t);
    if (bytes > c->slot) return ncclInvalidArgument;
    HIPS(hipStreamSynchronize(s));
    HIPS(hipMemcpy(slot_of(c, c->rank), send, bytes, hipMemcpyDeviceToHost));
    if (!barrier(c)) return ncclSystemError;
    c->host.resize(bytes * c->nranks);
    for (int r = 0; r < c->nranks; ++r) memcpy(c->host.data() + (size_t)r * bytes, slot_of(c, r), bytes);
    if (!barrier(c)) return ncclSystemError;
    HIPS(hipMemcpy(recv, c->host.data(), bytes * c->nranks, hipMemcpyHostToDevice));
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t s) {
    StubComm* c = (StubComm*)comm;
    const size_t bytes = count * elem(t);
    if (bytes > c->slot) return ncclInvalidArgument;
    HIPS(hipStreamSynchronize(s));
    if (c->rank == root) HIPS(hipMemcpy(slot_of(c, root), send, bytes, hipMemcpyDeviceToHost));
    if (!barrier(c)) return ncclSystemError;
    c->host.assign(slot_of(c, root), slot_of(c, root) + bytes);
    if (!barrier(c)) return ncclSystemError;
    HIPS(hipMemcpy(recv, c->host.data(), bytes, hipMemcpyHostToDevice));
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "success (stub)";
        case ncclUnhandledCudaError: return "HIP error inside the stub";
        case ncclSystemError: return "shared memory / barrier timeout inside the stub";
        case ncclInvalidArgument: return "invalid argument (stub: float sums only, message within BDOF_STUB_SLOT_MB)";
        default: return "error (stub)";
    }
}

}  // extern "C"
