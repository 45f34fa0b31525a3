// rf_rccl_abi.h -- the eight RCCL entry points librfhip.so binds with dlsym, as plain-C function
// pointer types (enums as int, ncclComm_t as void*, the 128-byte unique id as a struct by value).
// One definition, three users: rf_graph.cpp (the product), tests/native/fake_rccl.cpp (the
// shared-memory test double) and tests/native/rccl_abi_check.cpp, which is compiled against the
// REAL <rccl/rccl.h> and static_asserts that every type here is ABI-equivalent to RCCL's own
// prototype (tests/test_rccl_abi.py, CPU).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace rf {

struct NcclId { char internal[128]; };   // ncclUniqueId

typedef int (*NcclGetUniqueIdFn)(NcclId*);
typedef int (*NcclCommInitRankFn)(void**, int, NcclId, int);
typedef int (*NcclCommDestroyFn)(void*);
typedef int (*NcclSendFn)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*NcclRecvFn)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*NcclGroupFn)();
typedef const char* (*NcclGetErrorStringFn)(int);

constexpr int kNcclChar = 0;      // ncclInt8 / ncclChar
constexpr int kNcclSuccess = 0;   // ncclSuccess

}  // namespace rf
