// micro-benchmark: device -> pageable host container, 3.9 MB (108 k pair records), by route
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 108000 * 36;
    void *d, *pin;
    hipMalloc(&d, bytes), hipMemset(d, 1, bytes), hipHostMalloc(&pin, bytes, hipHostMallocDefault);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    std::vector<char> v(bytes);
    auto wait = [&]() { while (hipStreamQuery(st) == hipErrorNotReady) {} };
    for (int rep = 0; rep < 3; rep++)
    {
        double t0 = now();
        for (int i = 0; i < 20; i++) { hipMemcpyAsync(v.data(), d, bytes, hipMemcpyDeviceToHost, st); wait(); }
        double a = (now() - t0) / 20;
        t0 = now();
        for (int i = 0; i < 20; i++) { hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, st); wait(); }
        double dma = (now() - t0) / 20;
        t0 = now();
        for (int i = 0; i < 20; i++) { hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, st); wait(); memcpy(v.data(), pin, bytes); }
        double b = (now() - t0) / 20;
        double c[3];
        int    nts[3] = {2, 4, 8};
        for (int k = 0; k < 3; k++)
        {
            const int nt = nts[k];
            t0 = now();
            for (int i = 0; i < 20; i++)
            {
                hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, st); wait();
                std::vector<std::thread> th;
                for (int t = 0; t < nt; t++) th.emplace_back([&, t]() { size_t b0 = bytes * t / nt, e = bytes * (t + 1) / nt; memcpy(v.data() + b0, (char*)pin + b0, e - b0); });
                for (auto& x : th) x.join();
            }
            c[k] = (now() - t0) / 20;
        }
        t0 = now();
        for (int i = 0; i < 20; i++) { hipHostRegister(v.data(), bytes, hipHostRegisterDefault); hipMemcpyAsync(v.data(), d, bytes, hipMemcpyDeviceToHost, st); wait(); hipHostUnregister(v.data()); }
        double r = (now() - t0) / 20;
        // two halves pipelined: DMA of half 2 overlaps the host copy of half 1
        t0 = now();
        for (int i = 0; i < 20; i++)
        {
            const size_t h = bytes / 2;
            hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
            hipMemcpyAsync(pin, d, h, hipMemcpyDeviceToHost, st); hipEventRecord(e, st);
            hipMemcpyAsync((char*)pin + h, (char*)d + h, bytes - h, hipMemcpyDeviceToHost, st);
            while (hipEventQuery(e) == hipErrorNotReady) {}
            memcpy(v.data(), pin, h); wait(); memcpy(v.data() + h, (char*)pin + h, bytes - h);
            hipEventDestroy(e);
        }
        double p2 = (now() - t0) / 20;
        printf("rep %d: pageable direct %.3f | DMA to pinned only %.3f | pinned+memcpy %.3f | +2thr %.3f +4thr %.3f +8thr %.3f | register+copy %.3f | 2-stage pipelined %.3f ms\n",
               rep, a, dma, b, c[0], c[1], c[2], r, p2);
    }
    return 0;
}
