// host_visible_vram_probe.hip -- can the HOST store into device memory that kernels then poll locally?  (a mailbox in VRAM: every block of a
// persistent kernel could poll it without a PCIe read per poll.)  Tries fine-grained device memory (hipExtMallocWithFlags), managed memory
// and plain hipMalloc memory, each in a forked child (a refused store is a SIGSEGV), and times host store -> kernel sees it.
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>

__global__ void k_wait(const uint64_t *mail, uint64_t *seen_clk, uint64_t want, uint32_t max_spins) {
    for (uint32_t s = 0; s < max_spins; ++s) {
        if (__hip_atomic_load(mail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == want) {
            seen_clk[0] = wall_clock64();
            seen_clk[1] = s;
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    seen_clk[0] = 0;
    seen_clk[1] = max_spins;
}

static int run(int kind) {
    uint64_t *mail = nullptr, *seen = nullptr, *h_seen = nullptr;
    hipError_t e = hipSuccess;
    if (kind == 0) e = hipExtMallocWithFlags(reinterpret_cast<void **>(&mail), 4096, hipDeviceMallocFinegrained);
    else if (kind == 1) e = hipMallocManaged(reinterpret_cast<void **>(&mail), 4096);
    else if (kind == 2) e = hipMalloc(reinterpret_cast<void **>(&mail), 4096);
    else e = hipHostMalloc(reinterpret_cast<void **>(&mail), 4096, hipHostMallocMapped | hipHostMallocCoherent); // the baseline: host memory polled over PCIe
    if (e != hipSuccess) { std::printf("{\"kind\": %d, \"alloc\": \"%s\"}\n", kind, hipGetErrorString(e)); return 1; }
    if (kind == 1) (void)hipMemAdvise(mail, 4096, hipMemAdviseSetPreferredLocation, 0);
    hipMemset(mail, 0, 4096);
    hipMalloc(reinterpret_cast<void **>(&seen), 64);
    hipHostMalloc(reinterpret_cast<void **>(&h_seen), 64, hipHostMallocMapped);
    hipDeviceSynchronize();
    double best_us = 1e30, sum_us = 0;
    uint64_t spins_sum = 0;
    const int reps = 50;
    for (int r = 1; r <= reps; ++r) {
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, nullptr, mail, h_seen, (uint64_t)r, 1u << 22);
        usleep(300); // the kernel is polling by now
        const auto t0 = std::chrono::steady_clock::now();
        *reinterpret_cast<volatile uint64_t *>(mail) = (uint64_t)r; // THE STORE (SIGSEGV if the host may not)
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        if (hipDeviceSynchronize() != hipSuccess) { std::printf("{\"kind\": %d, \"sync failed\": true}\n", kind); return 1; }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (h_seen[0] == 0) { std::printf("{\"kind\": %d, \"seen\": false, \"rep\": %d}\n", kind, r); return 1; }
        best_us = us < best_us ? us : best_us;
        sum_us += us;
        spins_sum += h_seen[1];
    }
    std::printf("{\"kind\": %d, \"what\": \"%s\", \"host_store_to_kernel_exit_seen_by_host_us_min\": %.2f, \"mean\": %.2f, \"polls_before_seen_mean\": %.0f}\n", kind,
                kind == 0 ? "fine-grained device memory" : kind == 1 ? "managed memory, preferred on device" : kind == 2 ? "hipMalloc memory" : "host-mapped pinned memory (baseline)",
                best_us, sum_us / reps, (double)spins_sum / reps);
    return 0;
}

int main() {
    for (int kind = 0; kind < 4; ++kind) {
        std::fflush(stdout);
        pid_t pid = fork();
        if (pid == 0) { const int rc = run(kind); std::fflush(stdout); _exit(rc); }
        int st = 0;
        waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) std::printf("{\"kind\": %d, \"signal\": %d}\n", kind, WTERMSIG(st));
    }
    return 0;
}
