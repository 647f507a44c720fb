#include "options.h"

#include <cstdlib>

namespace nufft {

static thread_local const Options* g_current = nullptr;

void Options::parse(const char* text) {
    if (!text) return;
    std::string s(text);
    size_t i = 0;
    while (i < s.size()) {
        size_t j = i;
        while (j < s.size() && s[j] != ';' && s[j] != ' ' && s[j] != '\n' && s[j] != '\t') ++j;
        const std::string item = s.substr(i, j - i);
        const size_t eq = item.find('=');
        if (eq != std::string::npos && eq > 0) kv[item.substr(0, eq)] = item.substr(eq + 1);
        i = j + 1;
    }
}

std::string Options::str() const {
    std::string out;
    for (const auto& e : kv) {
        if (!out.empty()) out += ';';
        out += e.first + "=" + e.second;
    }
    return out;
}

void set_current_options(const Options* o) { g_current = o; }
const Options* current_options() { return g_current; }

const char* option_str(const char* name) {
    if (g_current) {
        auto it = g_current->kv.find(name);
        if (it != g_current->kv.end()) return it->second.c_str();
    }
#if defined(NUFFT_ENV_SWITCHES) && NUFFT_ENV_SWITCHES
    const char* v = std::getenv(name);
    if (v && *v) return v;
#endif
    return nullptr;
}

bool option_present(const char* name) { return option_str(name) != nullptr; }

int option_int(const char* name, int fallback) {
    const char* v = option_str(name);
    return (v && *v) ? std::atoi(v) : fallback;
}

double option_double(const char* name, double fallback) {
    const char* v = option_str(name);
    return (v && *v) ? std::atof(v) : fallback;
}

}  // namespace nufft
