// see knobs.h
#include "knobs.h"

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

const char* dev_knob(const char* name) {
    if (!name) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = table().find(name);
        if (it != table().end()) return it->second.c_str();    // (stable until the knob is set again: std::map nodes do not move)
    }
#ifdef MI355KKT_DEBUG
    return std::getenv(name);
#else
    return nullptr;
#endif
}

int set_dev_knob(const char* name, const char* value) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!name) { table().clear(); return 0; }
    if (!value) { table().erase(name); return 0; }
    table()[name] = value;
    return 0;
}

}  // namespace mi355kkt
