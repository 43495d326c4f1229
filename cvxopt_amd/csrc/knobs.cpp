// see knobs.h
#include "knobs.h"

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace mi355kkt {

namespace {
std::mutex g_mu;
std::map<std::string, std::string>& table() {
    static std::map<std::string, std::string> t;
    return t;
}
}  // namespace

std::atomic<int> g_count{0};        // number of knobs set: the common case (none) costs one relaxed load, no lock, no allocation

const char* dev_knob(const char* name) {
    if (!name) return nullptr;
    if (g_count.load(std::memory_order_acquire) != 0) {
        // the value is COPIED under the lock into storage of the calling thread: the pointer stays valid until this thread's next
        // dev_knob() call, whatever other threads set in the meantime
        static thread_local std::string mine;
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = table().find(name);
        if (it != table().end()) {
            mine = it->second;
            return mine.c_str();
        }
    }
#ifdef MI355KKT_DEBUG
    return std::getenv(name);
#else
    return nullptr;
#endif
}

int set_dev_knob(const char* name, const char* value) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!name) table().clear();
    else if (!value) table().erase(name);
    else table()[name] = value;
    g_count.store((int)table().size(), std::memory_order_release);
    return 0;
}

}  // namespace mi355kkt
