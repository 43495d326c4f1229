// Device allocations of the library: dev_alloc / dev_free (declared in kkt_common.h, used through DEV_ALLOC).
//
// Product behaviour: hipMalloc, then the block is cleared (zero bytes) on a private non-blocking stream and that stream is
// waited for -- the state of a handle must not depend on what a recycled block held before, and the clear must not
// wait on the legacy stream (which would wait for every blocking stream of the process, a torch user's included).
//
// Test modes, selected by knobs (include/mi355kkt_test.h; never read from the environment):
//   MI355KKT_ALLOC_POISON   every byte starts as 0xff (NaN in every double, -1 in every int): a read of memory nobody wrote
//                           changes a result instead of being hidden by the zeros
//   MI355KKT_ALLOC_RAW      no clear at all (the block's history shows through: the conditions of the round-4 abort)
//   MI355KKT_ALLOC_GUARD    "electric fence" for the GPU: every allocation is its own mapping of a whole number of 2 MB (which
//                           bypasses the runtime's sub-allocator for small blocks), filled with 0xff, and the caller's block is
//                           placed at the END of it: the first byte past the block is either unmapped or the poisoned front of the
//                           next guarded mapping.  An out-of-bounds access by a kernel -- also the "harmless" over-read of a
//                           vector load whose tail is masked -- becomes a memory fault (reported by the runtime as an error of
//                           the next synchronisation) or a NaN at once, in the test that performs it, instead of once in 2000
//                           handles when a block happens to end where a mapping ends.  (8-byte granularity: blocks whose size
//                           is not a multiple of 8 keep up to 7 bytes of slack.)
// Every allocation and release is also written to a ring of the last 65536 events; mi355kkt_test_install_abort_dump(path)
// installs a SIGABRT handler that writes the ring to `path` before the previous handler runs: the HIP runtime reports a GPU
// memory fault with the faulting address and abort()s on one of its own threads, and the ring maps that address to its owner
// (tools/alloc_owner.py).
#include <hip/hip_runtime.h>

#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <unistd.h>
#include <unordered_map>

#include "knobs.h"

namespace mi355kkt {

namespace {

struct Event { const void* p; size_t bytes; const char* file; int line; int op; };   // op 1 alloc, 2 free, 3 guarded alloc
constexpr unsigned RING = 1u << 16;
Event g_ring[RING];
std::atomic<unsigned long long> g_nev{0};
char g_dump_path[512] = "";
struct sigaction g_old_abrt;

void record(const void* p, size_t bytes, const char* file, int line, int op) {
    const unsigned long long k = g_nev.fetch_add(1, std::memory_order_relaxed);
    g_ring[k % RING] = Event{p, bytes, file, line, op};
}

std::mutex g_mu;
std::unordered_map<const void*, void*>& guard_bases() {          // guarded block -> base of its mapping
    static std::unordered_map<const void*, void*> m;
    return m;
}
hipStream_t g_clear[64] = {};

hipStream_t clear_stream() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_clear[dev] && hipStreamCreateWithFlags(&g_clear[dev], hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        g_clear[dev] = nullptr;
    }
    return g_clear[dev];
}

void drop_clear_stream(hipStream_t st) {          // the cached handle went stale (hipDeviceReset by the host application)
    std::lock_guard<std::mutex> lk(g_mu);
    for (hipStream_t& s : g_clear)
        if (s == st) s = nullptr;                 // (not destroyed: the handle belongs to a context that no longer exists)
}

hipError_t clear(void* p, int value, size_t bytes) {
    hipStream_t st = clear_stream();           // (nullptr: the legacy stream, as a last resort)
    hipError_t e = hipMemsetAsync(p, value, bytes, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess && st) {
        // a cached private stream does not survive a device reset.  Forget it, clear the error and retry ONCE on a
        // fresh stream (or on the legacy stream if none can be made); a second failure is the caller's error
        (void)hipGetLastError();
        drop_clear_stream(st);
        st = clear_stream();
        e = hipMemsetAsync(p, value, bytes, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    return e;
}

// (a crash handler: snprintf is not on the async-signal-safe list, but the process is about to die and the ring is all it reads)
void abort_dump(int sig) {
    if (g_dump_path[0]) {
        const int fd = open(g_dump_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd >= 0) {
            const unsigned long long n = g_nev.load(std::memory_order_relaxed);
            const unsigned long long first = n > RING ? n - RING : 0;
            char line[384];
            int len = snprintf(line, sizeof(line), "# mi355kkt allocation events %llu..%llu (op 1 alloc, 2 free, 3 guarded alloc)\n", first, n);
            if (len > 0) (void)!write(fd, line, (size_t)len);
            for (unsigned long long k = first; k < n; ++k) {
                const Event& e = g_ring[k % RING];
                len = snprintf(line, sizeof(line), "%llu %d %p %zu %s:%d\n", k, e.op, e.p, e.bytes, e.file ? e.file : "?", e.line);
                if (len > 0) (void)!write(fd, line, (size_t)len);
            }
            close(fd);
        }
    }
    // hand over to whoever was there before (faulthandler's traceback dump), then die with the default action
    if (g_old_abrt.sa_handler != SIG_DFL && g_old_abrt.sa_handler != SIG_IGN && g_old_abrt.sa_handler != nullptr) {
        sigaction(SIGABRT, &g_old_abrt, nullptr);
        raise(sig);
        return;
    }
    signal(SIGABRT, SIG_DFL);
    raise(sig);
}

}  // namespace

int install_abort_dump(const char* path) {
    if (!path || !*path || strlen(path) >= sizeof(g_dump_path)) return -1;
    const bool first = g_dump_path[0] == 0;
    strcpy(g_dump_path, path);
    if (!first) return 0;
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = abort_dump;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_NODEFER;
    return sigaction(SIGABRT, &sa, &g_old_abrt);
}

hipError_t dev_alloc(void** p, size_t bytes, const char* file, int line) {
    if (!p) return hipErrorInvalidValue;
    *p = nullptr;
    const bool guard = dev_knob("MI355KKT_ALLOC_GUARD") != nullptr;
    const int fill = dev_knob("MI355KKT_ALLOC_POISON") ? 0xff : 0;
    const bool raw = dev_knob("MI355KKT_ALLOC_RAW") != nullptr;
    if (guard) {
        // (measured, call r5c01: the runtime rounds a >= 2 MB allocation up to a multiple of 2 MB and maps all of it -- a block placed
        //  at the end of a request that is NOT such a multiple is followed by up to 2 MB of readable padding.  So the request is a
        //  multiple of 2 MB with at least 4 KB of poison in front of the block: what follows the block is either unmapped or the
        //  0xff-poisoned front of the neighbouring guarded mapping -- a fault or a NaN, never a plausible number.)
        constexpr size_t PAGE = 4096, HUGE = (size_t)2 << 20;
        const size_t user = (bytes + 7) & ~(size_t)7;
        const size_t total = (user + PAGE + HUGE - 1) / HUGE * HUGE;
        void* base = nullptr;
        hipError_t e = hipMalloc(&base, total);
        if (e != hipSuccess) return e;
        void* blk = static_cast<char*>(base) + (total - user);
        if ((e = clear(base, 0xff, total - user)) != hipSuccess ||        // what lies in front of the block is poison
            (user && (e = clear(blk, fill, user)) != hipSuccess)) {
            (void)hipFree(base);
            return e;
        }
        {
            std::lock_guard<std::mutex> lk(g_mu);
            guard_bases()[blk] = base;
        }
        record(blk, bytes, file, line, 3);
        *p = blk;
        return hipSuccess;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return e;
    record(*p, bytes, file, line, 1);
    if (bytes && !raw && (e = clear(*p, fill, bytes)) != hipSuccess) {
        record(*p, 0, nullptr, 0, 2);          // the ring pairs every alloc with a free
        (void)hipFree(*p);
        *p = nullptr;
    }
    return e;
}

std::atomic<int> g_violations{0};
int guard_violations() { return g_violations.load(); }

hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    record(p, 0, nullptr, 0, 2);
    void* base = p;
    bool guarded = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = guard_bases().find(p);
        if (it != guard_bases().end()) {
            base = it->second;
            guard_bases().erase(it);
            guarded = true;
        }
    }
    if (guarded) {
        // an out-of-bounds WRITE behind the block that precedes this mapping lands in the first bytes of its poisoned front
        // (at least 4 KB of 0xff): look at them before the mapping goes
        unsigned char front[4096];
        if (hipMemcpy(front, base, sizeof(front), hipMemcpyDeviceToHost) == hipSuccess) {
            for (size_t i = 0; i < sizeof(front); ++i)
                if (front[i] != 0xff) {
                    g_violations.fetch_add(1);
                    fprintf(stderr, "mi355kkt guard: byte %zu in front of block %p is 0x%02x, not poison: something wrote past the end of "
                                    "the block mapped before it\n", i, p, front[i]);
                    break;
                }
        } else {
            (void)hipGetLastError();
        }
    }
    return hipFree(base);
}

}  // namespace mi355kkt
