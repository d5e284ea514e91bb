// sync_stall_probe -- does a host wait for a sub-millisecond kernel sometimes take tens of
// milliseconds on this box, whatever the program?  (mc3_amd.sample() on C5 with 1e5 samples
// shows ONE ~75 ms wait per call when calls follow each other, in a different HIP call every
// time -- tools/stall_probe.py, profiles/r06_e2e_sample.txt.)  A stream of its own, a kernel of
// ~0.5 ms, hipStreamSynchronize after each; waits beyond 5 ms are listed with their time.
//   sync_stall_probe [seconds = 3] [pause_us between launches = 0] [burst_ms = 0, idle_ms = 0]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ void spin_kernel(unsigned long long ticks, unsigned long long *out)
{
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    do {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    } while (t1 - t0 < ticks);
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *out = t1 - t0;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int pause_us = argc > 2 ? atoi(argv[2]) : 0;
    const int burst_ms = argc > 3 ? atoi(argv[3]) : 0, idle_ms = argc > 4 ? atoi(argv[4]) : 0;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
        return 1;
    unsigned long long *d = nullptr;
    (void)hipMalloc(&d, 8);
    using clk = std::chrono::steady_clock;
    const auto t_start = clk::now();
    auto since = [&](clk::time_point t) {
        return std::chrono::duration<double, std::milli>(t - t_start).count();
    };
    long n = 0, slow = 0;
    double worst = 0.0, sum = 0.0;
    auto burst_start = clk::now();
    while (since(clk::now()) < seconds * 1e3) {
        const auto t0 = clk::now();
        hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, s, 50000ull, d);   // 0.5 ms
        (void)hipStreamSynchronize(s);
        const auto t1 = clk::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        ++n;
        sum += ms;
        worst = ms > worst ? ms : worst;
        if (ms > 5.0) {
            ++slow;
            printf("  wait of %7.2f ms at t = %8.1f ms\n", ms, since(t0));
        }
        if (pause_us)
            std::this_thread::sleep_for(std::chrono::microseconds(pause_us));
        if (burst_ms && std::chrono::duration<double, std::milli>(clk::now() - burst_start).count() >
                            burst_ms) {
            std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
            burst_start = clk::now();
        }
    }
    printf("%ld launches of a 0.5 ms kernel in %.1f s (pause %d us, bursts %d ms / idle %d ms): "
           "mean wait %.3f ms, worst %.2f ms, %ld waits beyond 5 ms\n",
           n, seconds, pause_us, burst_ms, idle_ms, sum / n, worst, slow);
    return 0;
}
