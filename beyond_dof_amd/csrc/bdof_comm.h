// Collectives of the multi-GPU path behind the C ABI (include/bdof.h, bdof_comm_*): RCCL over xGMI, one process per GPU.
// Replaces comm.Allreduce(this_grads, grads) of cnn_propagator/fullfield.py:348-351 / ptychography.py:302-306.
//
// librccl.so is opened at run time (dlopen) the first time a communicator is asked for: single-GPU runs never map the
// 570-MB library, the CPU-only container can load libbdof.so, and a process that already holds an RCCL (e.g. one that
// imported torch) gets that copy by SONAME instead of a second one.
//
// Ordering model: a communicator owns one HIP stream.  Every collective is enqueued there behind an event recorded on
// the ctx stream at the call (so it sees everything the ctx has enqueued so far) and leaves a "ticket" — an event on the
// communicator's stream.  bdof_comm_wait(ctx, ticket) makes the ctx stream wait for it.  No host synchronisation: the
// producer of the next slab and the consumer of the previous one keep running while a slab is on the wire.
#pragma once
#include <cstdlib>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string>
#include <vector>

#define BDOF_COMM_TICKETS 256

struct RcclApi {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

static RcclApi g_rccl;

static bool rccl_load() {
    if (g_rccl.dl) return true;
    // BDOF_RCCL_LIB names the library to bind instead (tests/rccl_stub: a stand-in that lets several ranks share one GPU, so
    // that this file's callers run with nranks > 1 on a single-GPU box); RTLD_LOCAL there: its nccl* symbols stay private
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* dl = nullptr;
    const char* over = getenv("BDOF_RCCL_LIB");
    if (over && over[0]) {
        dl = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!dl) {
            const char* e = dlerror();
            g_rccl.err = std::string("dlopen(BDOF_RCCL_LIB=") + over + ") failed: " + (e ? e : "?");
            return false;
        }
    }
    for (const char* n : names) {
        if (dl) break;
        dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!dl) {
        const char* e = dlerror();
        g_rccl.err = std::string("dlopen(librccl.so.1) failed: ") + (e ? e : "?");
        return false;
    }
#define BDOF_RCCL_SYM(field, name)                                                          \
    *(void**)(&g_rccl.field) = dlsym(dl, name);                                             \
    if (!g_rccl.field) { g_rccl.err = std::string("librccl lacks ") + name; dlclose(dl); return false; }
    BDOF_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    BDOF_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    BDOF_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    BDOF_RCCL_SYM(AllReduce, "ncclAllReduce")
    BDOF_RCCL_SYM(ReduceScatter, "ncclReduceScatter")
    BDOF_RCCL_SYM(AllGather, "ncclAllGather")
    BDOF_RCCL_SYM(Broadcast, "ncclBroadcast")
    BDOF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef BDOF_RCCL_SYM
    g_rccl.dl = dl;
    return true;
}

struct bdof_comm {
    int device = 0, nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr;                    // "the ctx stream has reached this point"
    hipEvent_t ticket[BDOF_COMM_TICKETS] = {};
    unsigned next = 0;
    std::string err;
};
