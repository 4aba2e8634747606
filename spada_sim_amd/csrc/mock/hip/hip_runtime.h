// TEST DOUBLE of the few HIP runtime entry points spada_comm.hip uses -- NOT a HIP implementation and never shipped.
// `make -C spada_sim_amd/csrc asan` compiles the host half of spada_comm.hip against this header and mock/rccl/rccl.h with g++ under
// -fsanitize=address,undefined (lib_asan/libspada_comm_mock.so): "device" memory is host memory, copies run at once, streams and
// events are tokens, and every RCCL call is logged (mock/comm_mock.cpp) so that tests/test_host_cpu.py can check that all ranks of
// an N = 8 exchange post the same collectives in the same order -- what no one-GPU box can show on the real library.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0, hipErrorOutOfMemory = 2;
typedef struct mock_stream_t *hipStream_t;
typedef struct mock_event_t *hipEvent_t;
constexpr unsigned hipStreamNonBlocking = 1;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

inline const char *hipGetErrorString(hipError_t) { return "mock hip error"; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)std::malloc(1); return *s ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
