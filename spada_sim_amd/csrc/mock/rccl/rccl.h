// TEST DOUBLE of the RCCL entry points spada_comm.hip uses (see mock/hip/hip_runtime.h): every call is appended to a log that the test
// reads back; ncclAllGather returns what the test has scripted as the gathered words of all ranks; ncclBroadcast moves nothing (the
// test replays the logged broadcasts between the ranks' buffers itself).
#pragma once
#include <cstddef>
#include <cstdint>

#include "hip/hip_runtime.h"

typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
typedef struct mock_comm_t *ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum ncclDataType_t { ncclUint32 = 3, ncclUint64 = 5, ncclFloat64 = 8 };

extern "C" {
const char *ncclGetErrorString(ncclResult_t);
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int nranks, ncclUniqueId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t, int root, ncclComm_t, hipStream_t);
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t, ncclComm_t, hipStream_t);
}
