// hip_runtime.h -- TEST INFRASTRUCTURE, CPU only: just enough of the HIP host API for hexl-fpga_amd/csrc/capi.hip (the C-ABI's host half:
// staging pipeline, copy-thread pool, plan set-up) to compile with g++ and RUN without a GPU. "Device" memory is host memory, copies are
// memcpy on the calling thread, streams and events are ordering no-ops (everything is synchronous), the kernel launchers are the stubs
// of tests/cpp/stage_model_stubs.cpp. What is left is exactly the HOST side of the host-pointer entry points -- packing, unpacking /
// accumulating, the runner threads of hexl_fpga_api.cpp above it -- which tests/test_host_staging_model.py times for NUM_DEV = 1 ... 8
// under this pod's CPU quota (the host side of NUM_DEV = 8 was unmodelled). Never part of the product.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999, hipErrorNotReady = 600, hipErrorInvalidValue = 1 };
typedef struct hipShimStream* hipStream_t;
typedef struct hipShimEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocCoherent = 0x40000000, hipHostMallocMapped = 2 };
enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipPointerAttribute_t { int type; };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hip shim error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { const char* e = getenv("FAKE_DEVICES"); *n = e ? atoi(e) : 8; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceGetPCIBusId(char* b, int len, int) { if (len > 0) b[0] = 0; return hipSuccess; }   // (no sysfs node: no pinning)
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "CPU staging model (tests/cpp/hip_shim)"); strcpy(p->gcnArchName, "none");
    p->multiProcessorCount = 256; p->totalGlobalMem = size_t(288) << 30;
    return hipSuccess;
}
static inline hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 4096, n ? n : 1) ? hipErrorUnknown : hipSuccess; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
// (a copy engine moves these bytes in the product, not a host core: the model charges nothing for them -- HEXL_MODEL_COPY=1 does a memcpy)
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    static const bool real = [] { const char* e = getenv("HEXL_MODEL_COPY"); return e && atoi(e) == 1; }();
    if (real || n <= 4096) memcpy(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { if (n <= 4096) memset(p, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(1); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeHost; return hipErrorInvalidValue; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
