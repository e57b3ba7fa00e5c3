// pm_rccl_abi.h -- the part of rccl.h (RCCL 2.x ABI) pm_comm.hip binds at run time, declared by hand: the library
// carries no build- or link-time dependency on RCCL. tests/rccl_abi_check.cpp includes this header NEXT TO the
// installed <rccl/rccl.h> and static_asserts that every type, constant and prototype below is the installed one's
// (tests/test_capi_symbols.py compiles it) - nothing else may declare these.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace pm_rccl {

typedef struct ncclComm *Comm;  // ncclComm_t: a pointer to an opaque struct
struct UniqueId {               // ncclUniqueId: NCCL_UNIQUE_ID_BYTES chars, passed BY VALUE to ncclCommInitRank
    char internal[128];
};
// ncclResult_t / ncclDataType_t / ncclRedOp_t are C enums (int-sized): the values used
enum { Success = 0, Int32 = 2, Float64 = 8, Sum = 0 };

using GetUniqueId_t = int (*)(UniqueId *);
using CommInitRank_t = int (*)(Comm *, int, UniqueId, int);
using CommDestroy_t = int (*)(Comm);
using CommAbort_t = int (*)(Comm);
using AllGather_t = int (*)(const void *, void *, size_t, int, Comm, hipStream_t);
using AllReduce_t = int (*)(const void *, void *, size_t, int, int, Comm, hipStream_t);
using Send_t = int (*)(const void *, size_t, int, int, Comm, hipStream_t);
using Recv_t = int (*)(void *, size_t, int, int, Comm, hipStream_t);
using GroupStart_t = int (*)();
using GroupEnd_t = int (*)();
using GetErrorString_t = const char *(*)(int);

// X(member, soname symbol)
#define PM_RCCL_SYMBOLS(X)                 \
    X(GetUniqueId, "ncclGetUniqueId")      \
    X(CommInitRank, "ncclCommInitRank")    \
    X(CommDestroy, "ncclCommDestroy")      \
    X(CommAbort, "ncclCommAbort")          \
    X(AllGather, "ncclAllGather")          \
    X(AllReduce, "ncclAllReduce")          \
    X(Send, "ncclSend")                    \
    X(Recv, "ncclRecv")                    \
    X(GroupStart, "ncclGroupStart")        \
    X(GroupEnd, "ncclGroupEnd")            \
    X(GetErrorString, "ncclGetErrorString")

}  // namespace pm_rccl
