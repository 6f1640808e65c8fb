// probe_waitvalue.hip — does a kernel on stream A release work queued on stream B through a
// memory flag + hipStreamWaitValue64 (command-processor wait, no polling kernel), and what does the
// hand-off cost?  Used to decide the design of the merged frame+bulk launch.
//   hipcc --offload-arch=gfx950 -O2 -o probe_waitvalue probe_waitvalue.hip && ./probe_waitvalue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::printf("FAIL %s: %s\n", #x, hipGetErrorString(e_));                       \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

// producer: many blocks; each bumps a counter when done; the last one publishes `pass` in the flag
__global__ void producer(unsigned* counter, unsigned long long* flag, unsigned long long pass, unsigned nblocks,
                         unsigned long long* t_flag, int spin) {
    for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        const unsigned done = atomicAdd(counter, 1u);
        if (done == nblocks - 1) {
            *counter = 0;  // ready for the next pass
            *t_flag = wall_clock64();
            __hip_atomic_store(flag, pass, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void consumer(unsigned long long* t_start, const unsigned long long* flag, unsigned long long* seen) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *t_start = wall_clock64();
        *seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main() {
    int dev = 0, ok = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, dev));
    std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", ok);
    unsigned long long* flag = nullptr;
    CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&flag), 8, hipMallocSignalMemory));
    *flag = 0;  // signal memory is host-visible
    unsigned* counter;
    unsigned long long *t_flag, *t_start, *seen;
    CK(hipMalloc(&counter, 4));
    CK(hipMemset(counter, 0, 4));
    CK(hipHostMalloc(&t_flag, 8));
    CK(hipHostMalloc(&t_start, 8));
    CK(hipHostMalloc(&seen, 8));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const unsigned nblocks = 300;
    double worst = 0, sum = 0;
    int bad = 0;
    const int passes = 200;
    for (int p = 1; p <= passes; ++p) {
        // consumer side is queued FIRST (it must not run before the producer of this pass has finished)
        CK(hipStreamWaitValue64(b, flag, static_cast<uint64_t>(p), hipStreamWaitValueGte, ~0ull));
        hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, t_start, flag, seen);
        hipLaunchKernelGGL(producer, dim3(nblocks), dim3(256), 0, a, counter, flag, static_cast<unsigned long long>(p),
                           nblocks, t_flag, 200 + (p % 7) * 100);
        CK(hipStreamSynchronize(b));
        CK(hipStreamSynchronize(a));
        const double us = (static_cast<double>(*t_start) - static_cast<double>(*t_flag)) / 100.0;  // 100 MHz clock
        if (*seen < static_cast<unsigned long long>(p) || us < 0) ++bad;
        if (p > 5) {
            sum += us;
            if (us > worst) worst = us;
        }
    }
    std::printf("passes %d  ordering violations %d  hand-off flag->consumer start: mean %.2f us  worst %.2f us\n", passes, bad,
                sum / (passes - 5), worst);
    // the same hand-off through an event, for comparison
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    sum = worst = 0;
    for (int p = 1; p <= passes; ++p) {
        hipLaunchKernelGGL(producer, dim3(nblocks), dim3(256), 0, a, counter, flag, static_cast<unsigned long long>(passes + p),
                           nblocks, t_flag, 200);
        CK(hipEventRecord(ev, a));
        CK(hipStreamWaitEvent(b, ev, 0));
        hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, t_start, flag, seen);
        CK(hipStreamSynchronize(b));
        const double us = (static_cast<double>(*t_start) - static_cast<double>(*t_flag)) / 100.0;
        if (p > 5) {
            sum += us;
            if (us > worst) worst = us;
        }
    }
    std::printf("event record + stream wait event: mean %.2f us  worst %.2f us\n", sum / (passes - 5), worst);
    std::printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad ? 2 : 0;
}
