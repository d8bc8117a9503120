// How long does the HOST spend inside hipMemcpyAsync (pinned -> device, non-blocking stream) when 16 copies of a given size are queued back to
// back?  (round 4: the CLI's flush of a 16.8-MB batch in 16 pieces of 1.05 MB sometimes blocked for 8 ms inside these calls.)
//   hipcc --offload-arch=gfx950 -O2 tools/memcpy_probe.hip -o tools/memcpy_probe && tools/memcpy_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t total = 96u << 20;
    char *h = nullptr, *d = nullptr;
    hipStream_t st;
    if (hipHostMalloc((void**)&h, total, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&d, total) != hipSuccess ||
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
        return 1;
    memset(h, 1, total);
    (void)hipMemcpyAsync(d, h, 1 << 20, hipMemcpyHostToDevice, st);
    (void)hipStreamSynchronize(st);
    const size_t sizes[] = {262144, 524288, 1000000, 1048576, 1050315, 1100000, 2000000, 4000000};
    for (size_t sz : sizes) {
        for (int rep = 0; rep < 4; ++rep) {
            double worst = 0, sum = 0;
            const double t0 = now_ms();
            for (int i = 0; i < 16; ++i) {
                const double a = now_ms();
                (void)hipMemcpyAsync(d + (size_t)i * sz, h + (size_t)i * sz + 100 * i, sz, hipMemcpyHostToDevice, st);
                const double b = now_ms() - a;
                worst = b > worst ? b : worst;
                sum += b;
            }
            const double queued = now_ms() - t0;
            (void)hipStreamSynchronize(st);
            printf("size %8zu x16: host time in the calls %.3f ms (worst call %.3f), queued after %.3f ms, done after %.3f ms\n", sz, sum, worst, queued, now_ms() - t0);
        }
    }
    // which calls block: 8192 copies of 64 KB on a fresh stream, every call that takes > 0.5 ms of host time is reported with its index
    for (int trial = 0; trial < 3; ++trial) {
        hipStream_t s2;
        (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        for (int i = 0; i < 8192; ++i) {
            const double a = now_ms();
            (void)hipMemcpyAsync(d + (size_t)(i % 1024) * 65536, h + (size_t)(i % 1024) * 65536, 65536, hipMemcpyHostToDevice, s2);
            const double b = now_ms() - a;
            if (b > 0.5) printf("trial %d: copy #%d blocked the host for %.3f ms\n", trial, i, b);
            if (i % 64 == 63) (void)hipStreamSynchronize(s2);
        }
        (void)hipStreamSynchronize(s2);
        (void)hipStreamDestroy(s2);
    }
    return 0;
}
