// Stand-in for libamdhip64 in the HOST-SIDE sanitizer build of libgsdd (tools/host_asan/build.sh): the C-ABI wrappers of csrc/*.hip are
// compiled host-only (hipcc --cuda-host-only -fsanitize=address,undefined), so every argument check, descriptor copy, grid computation
// and workspace carve-up runs under ASan / UBSan on the CPU, and the kernels themselves are never executed: a launch is validated
// (non-empty grid and block, block <= 1024 threads, dynamic LDS <= 160 KB) and counted.  Test infrastructure only; never shipped.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace {
std::atomic<long> g_launches{0};
thread_local dim3 g_grid, g_block;
thread_local size_t g_shmem = 0;
thread_local hipStream_t g_stream = nullptr;
thread_local hipError_t g_last = hipSuccess;
int g_capturing = 0;
long g_fail_set_attribute = 0;     // > 0: the next n hipFuncSetAttribute calls fail (drives the retry path of GSDD_ONCE_PER_DEVICE)
}  // namespace

extern "C" {
long gsdd_stub_launches() { return g_launches.load(); }
void gsdd_stub_fail_next_set_attribute(long n) { g_fail_set_attribute = n; }

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    g_grid = grid; g_block = block; g_shmem = shmem; g_stream = stream;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    *grid = g_grid; *block = g_block; *shmem = g_shmem; *stream = g_stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void* fn, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t) {
    const unsigned long long threads = (unsigned long long)block.x * block.y * block.z;
    if (fn == nullptr || args == nullptr || grid.x == 0 || grid.y == 0 || grid.z == 0 || threads == 0 || threads > 1024 || shmem > 160 * 1024 ||
        grid.y > 65535 || grid.z > 65535) {
        std::fprintf(stderr, "hip_stub: invalid launch grid (%u,%u,%u) block (%u,%u,%u) lds %zu\n", grid.x, grid.y, grid.z, block.x, block.y,
                     block.z, shmem);
        g_last = hipErrorInvalidConfiguration;
        return g_last;
    }
    g_launches.fetch_add(1);
    return hipSuccess;
}
hipError_t hipGetLastError() { const hipError_t e = g_last; g_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 160 * 1024; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int value) {
    if (g_fail_set_attribute > 0) { --g_fail_set_attribute; return hipErrorInvalidValue; }
    return value <= 160 * 1024 ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipMemsetAsync(void* p, int, size_t, hipStream_t) { return p != nullptr ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipMemset2DAsync(void* p, size_t pitch, int, size_t width, size_t height, hipStream_t) {
    return (p != nullptr && width <= pitch && height > 0) ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { if (g_capturing) return hipErrorIllegalState; g_capturing = 1; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
    if (!g_capturing) { *g = nullptr; return hipErrorIllegalState; }
    g_capturing = 0;
    *g = reinterpret_cast<hipGraph_t>(std::malloc(8));
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = reinterpret_cast<hipGraphExec_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { std::free(g); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
// fat-binary registration of the host-only objects: nothing to register
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}
}
