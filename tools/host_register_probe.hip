// host_register_probe.hip -- what would registering the caller's own buffers (hipHostRegister, cached by pointer) buy the host-pointer
// KeySwitch path? Measures: (1) the cost of hipHostRegister / hipHostUnregister per buffer of a keyswitch object
// (0.79 MB t_target, 1.57 MB result, malloc'ed), (2) the rate at which a kernel reads / the copy engine copies registered pageable memory
// against a hipHostMalloc slab, (3) whether a registration FOLLOWS the virtual address when the caller frees and re-allocates the
// buffer at the same address (munmap + mmap MAP_FIXED): a cache keyed on the pointer is only safe if it does.
//   hipcc --offload-arch=gfx950 -O2 tools/host_register_probe.hip -o tools/host_register_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_sum(const unsigned long long* __restrict__ p, size_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) s += p[i];
    atomicAdd(out, s);
}

int main() {
    unsigned long long* d_out;
    CK(hipMalloc(&d_out, 8));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // (1) registration cost
    for (size_t bytes : {size_t(6) * 16384 * 8, size_t(12) * 16384 * 8}) {
        std::vector<void*> bufs(64);
        for (auto& b : bufs) { b = malloc(bytes); memset(b, 1, bytes); }
        double t0 = now();
        for (auto b : bufs) CK(hipHostRegister(b, bytes, hipHostRegisterDefault));
        double t1 = now();
        for (auto b : bufs) CK(hipHostUnregister(b));
        double t2 = now();
        printf("hipHostRegister %7zu B (malloc): %.1f us, hipHostUnregister %.1f us per buffer\n", bytes, (t1 - t0) / 64 * 1e6, (t2 - t1) / 64 * 1e6);
        for (auto b : bufs) free(b);
    }
    // (2) read rates: 96 MB
    const size_t bytes = size_t(96) << 20, n = bytes / 8;
    void *reg = malloc(bytes), *pin = nullptr, *dev = nullptr;
    memset(reg, 3, bytes);
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    memset(pin, 3, bytes);
    CK(hipMalloc(&dev, bytes));
    CK(hipHostRegister(reg, bytes, hipHostRegisterDefault));
    void *dreg = nullptr, *dpin = nullptr;
    CK(hipHostGetDevicePointer(&dreg, reg, 0));
    CK(hipHostGetDevicePointer(&dpin, pin, 0));
    for (int which = 0; which < 2; ++which) {
        const void* src = which ? dpin : dreg;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(d_out, 0, 8, st));
            double t0 = now();
            hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, st, (const unsigned long long*)src, n, d_out);
            CK(hipStreamSynchronize(st));
            double dt = now() - t0;
            if (rep) printf("kernel reads 96 MB of %s host memory: %.1f GB/s\n", which ? "hipHostMalloc" : "REGISTERED malloc", bytes / dt / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(dev, which ? pin : reg, bytes, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            double dt = now() - t0;
            if (rep) printf("hipMemcpyAsync H2D 96 MB from %s: %.1f GB/s\n", which ? "hipHostMalloc" : "REGISTERED malloc", bytes / dt / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(which ? pin : reg, dev, bytes, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double dt = now() - t0;
            if (rep) printf("hipMemcpyAsync D2H 96 MB into %s: %.1f GB/s\n", which ? "hipHostMalloc" : "REGISTERED malloc", bytes / dt / 1e9);
        }
    }
    {   // un-registered pageable memory through hipMemcpyAsync (what a caller's plain std::vector costs the runtime)
        void* plain = malloc(bytes);
        memset(plain, 3, bytes);
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(dev, plain, bytes, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            double dt = now() - t0;
            if (rep) printf("hipMemcpyAsync H2D 96 MB from PLAIN malloc: %.1f GB/s\n", bytes / dt / 1e9);
        }
        free(plain);
    }
    CK(hipHostUnregister(reg));
    // (3) does a registration follow the virtual address?
    const size_t mb = size_t(2) << 20;
    void* at = mmap(nullptr, mb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (at == MAP_FAILED) { printf("mmap failed\n"); return 1; }
    for (size_t i = 0; i < mb / 8; ++i) ((unsigned long long*)at)[i] = 1;
    CK(hipHostRegister(at, mb, hipHostRegisterDefault));
    void* dat = nullptr;
    CK(hipHostGetDevicePointer(&dat, at, 0));
    unsigned long long h = 0;
    CK(hipMemsetAsync(d_out, 0, 8, st));
    hipLaunchKernelGGL(k_sum, dim3(64), dim3(256), 0, st, (const unsigned long long*)dat, mb / 8, d_out);
    CK(hipMemcpyAsync(&h, d_out, 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("registered region, pattern 1: kernel sums %llu (want %zu)\n", h, mb / 8);
    munmap(at, mb);
    void* again = mmap(at, mb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED, -1, 0);
    printf("re-mapped at the same address: %s\n", again == at ? "yes" : "no");
    for (size_t i = 0; i < mb / 8; ++i) ((unsigned long long*)again)[i] = 2;
    fflush(stdout);
    CK(hipMemsetAsync(d_out, 0, 8, st));
    hipLaunchKernelGGL(k_sum, dim3(64), dim3(256), 0, st, (const unsigned long long*)dat, mb / 8, d_out);
    CK(hipMemcpyAsync(&h, d_out, 8, hipMemcpyDeviceToHost, st));
    hipError_t e = hipStreamSynchronize(st);
    printf("after munmap + mmap(MAP_FIXED) + pattern 2, through the OLD registration: %s, kernel sums %llu (pattern 2 would be %zu, stale pattern 1 %zu)\n",
           hipGetErrorString(e), h, 2 * (mb / 8), mb / 8);
    e = hipHostRegister(again, mb, hipHostRegisterDefault);
    printf("registering the new mapping again: %s\n", hipGetErrorString(e));
    return 0;
}
