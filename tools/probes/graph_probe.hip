// Does replaying a step as a hipGraph shorten the gaps between DEPENDENT kernels on gfx950?  A chain of 12 kernels
// (each spins ~20 us, like the step's: prologue, two search kernels, compaction x3, Gauss-Newton x6) launched
//   (1) one by one on a stream,  (2) as one captured graph,  (3) as one kernel (the bound),
// each followed by a 96-byte read-back and a host wait, like one ICP step.  Prints the span per step.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/graph_probe.hip -o gpurun_out/graph_probe && gpurun_out/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(unsigned long long ticks, double* out, int blocks_work)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1.0;
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double* d;
    CK(hipMalloc(&d, 1024));
    CK(hipMemset(d, 0, 1024));
    double* h;
    CK(hipHostMalloc(&h, 1024));
    const int K = 12;
    const unsigned long long ticks = 2000;  // 100 MHz clock: 20 us
    const int grids[K] = {2048, 4096, 4096, 1024, 1, 1024, 256, 1, 256, 1, 256, 1};
    auto chain = [&]() {
        for (int k = 0; k < K; k++) hipLaunchKernelGGL(spin, dim3(grids[k]), dim3(64), 0, s, ticks, d, 0);
        (void)hipMemcpyAsync(h, d, 96, hipMemcpyDeviceToHost, s);
    };
    auto wait = [&]() { while (hipStreamQuery(s) == hipErrorNotReady) {} };
    auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int steps = 200;
    // (1) stream
    for (int i = 0; i < 20; i++) chain(), wait();
    double t0 = now();
    for (int i = 0; i < steps; i++) chain(), wait();
    const double us_stream = (now() - t0) / steps;
    // (2) graph
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; i++) { CK(hipGraphLaunch(ge, s)); wait(); }
    t0 = now();
    for (int i = 0; i < steps; i++) { (void)hipGraphLaunch(ge, s); wait(); }
    const double us_graph = (now() - t0) / steps;
    // (3) one kernel of the same total spin
    for (int i = 0; i < 20; i++) { hipLaunchKernelGGL(spin, dim3(2048), dim3(64), 0, s, ticks * K, d, 0); (void)hipMemcpyAsync(h, d, 96, hipMemcpyDeviceToHost, s); wait(); }
    t0 = now();
    for (int i = 0; i < steps; i++) { hipLaunchKernelGGL(spin, dim3(2048), dim3(64), 0, s, ticks * K, d, 0); (void)hipMemcpyAsync(h, d, 96, hipMemcpyDeviceToHost, s); wait(); }
    const double us_one = (now() - t0) / steps;
    // (4) 5 kernels on a stream (what fusing the compaction and the solver's kernels would leave)
    auto chain5 = [&]() {
        const unsigned long long tk[5] = {ticks, ticks * 2, ticks * 2, ticks * 3, ticks * 4};
        for (int k = 0; k < 5; k++) hipLaunchKernelGGL(spin, dim3(grids[k]), dim3(64), 0, s, tk[k], d, 0);
        (void)hipMemcpyAsync(h, d, 96, hipMemcpyDeviceToHost, s);
    };
    for (int i = 0; i < 20; i++) chain5(), wait();
    t0 = now();
    for (int i = 0; i < steps; i++) chain5(), wait();
    const double us_five = (now() - t0) / steps;
    printf("spin total %d us per step\n", (int)(K * ticks / 100));
    printf("12 kernels on a stream : %.1f us per step (overhead %.1f)\n", us_stream, us_stream - K * ticks / 100.0);
    printf("12 kernels as a graph  : %.1f us per step (overhead %.1f)\n", us_graph, us_graph - K * ticks / 100.0);
    printf(" 5 kernels on a stream : %.1f us per step (overhead %.1f)\n", us_five, us_five - K * ticks / 100.0);
    printf(" 1 kernel              : %.1f us per step (overhead %.1f)\n", us_one, us_one - K * ticks / 100.0);
    return 0;
}
