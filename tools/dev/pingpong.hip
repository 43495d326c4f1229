// Developer microbenchmark (round 4): latency of one agent-scope hand-off between two workgroups on gfx950, the floor of every
// chain in this library (tile Cholesky, triangular solves).  Workgroup 0 and workgroup B (B chosen so that they share / do not
// share an XCD under round-robin placement) bounce a counter N times; reported: ns per one-way hand-off.
//   variants: polling by 1 lane or by 256 lanes of the consumer (256 granules = 16 cache lines, as trsv does), with / without s_sleep,
//             payload through the same 8-byte word (data is the flag)
//   hipcc --offload-arch=gfx950 -O3 tools/dev/pingpong.hip -o tools/dev/bin/pingpong && tools/dev/bin/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <int LANES, bool SLEEP>
__global__ __launch_bounds__(256) void pingpong(u64* a, u64* b, int partner, int rounds, unsigned* xcc, long long* cycles) {
    const int me = blockIdx.x;
    if (me != 0 && me != partner) return;
    const int tid = threadIdx.x;
    if (tid == 0) xcc[me == 0 ? 0 : 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
    u64* mine = me == 0 ? a : b;      // I write here
    u64* theirs = me == 0 ? b : a;    // I poll here
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {
            if (tid < LANES) __hip_atomic_store(mine + tid, (u64)r, RLX_AGENT);
        }
        if (tid < LANES) {
            for (unsigned spins = 0; spins < (1u << 24); ++spins) {
                if (__hip_atomic_load(theirs + tid, RLX_AGENT) == (u64)r) break;
                if (SLEEP) __builtin_amdgcn_s_sleep(1);
            }
        }
        if (LANES > 64) __syncthreads();
        if (me != 0) {
            if (tid < LANES) __hip_atomic_store(mine + tid, (u64)r, RLX_AGENT);
        }
    }
    if (tid == 0 && me == 0) *cycles = __builtin_readcyclecounter() - t0;
}

template <int LANES, bool SLEEP>
static void run(const char* name, int partner, int rounds) {
    u64 *a, *b; unsigned* xcc; long long* cyc;
    hipMalloc(&a, 256 * 8); hipMalloc(&b, 256 * 8); hipMalloc(&xcc, 8); hipMalloc(&cyc, 8);
    hipMemset(a, 0, 256 * 8); hipMemset(b, 0, 256 * 8); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((pingpong<LANES, SLEEP>), dim3(256), dim3(256), 0, 0, a, b, partner, 10, xcc, cyc);   // warm-up (values 1..10)
    hipMemset(a, 0, 256 * 8); hipMemset(b, 0, 256 * 8); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((pingpong<LANES, SLEEP>), dim3(256), dim3(256), 0, 0, a, b, partner, rounds, xcc, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned hx[2]; hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
    printf("%-34s partner wg %3d (xcc %u vs %u): %.0f ns per one-way hand-off\n", name, partner, hx[0] & 15, hx[1] & 15, 1e6 * ms / (2.0 * rounds));
    hipFree(a); hipFree(b); hipFree(xcc); hipFree(cyc);
}

int main() {
    const int rounds = 2000;
    for (int partner : {8, 1, 4, 128}) {
        run<1, false>("1 lane, busy poll", partner, rounds);
        run<1, true>("1 lane, s_sleep(1)", partner, rounds);
        run<64, true>("64 lanes, s_sleep(1)", partner, rounds);
        run<256, true>("256 lanes + barrier, s_sleep(1)", partner, rounds);
        run<256, false>("256 lanes + barrier, busy poll", partner, rounds);
    }
    return 0;
}
