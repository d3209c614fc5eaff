// rk_policy.cpp -- the hashing policy as text: what `--hash-policy` / RKMH_POLICY of both command lines parse, and what a sketch file
// records.  The arithmetic it selects lives in the un-vendored mkmh submodule (/root/reference/.gitmodules:1-3; the include at
// src/rkmh.cpp:17), so every choice the reference's tree does not fix (SURVEY.md section 8c, U1 ... U12) is a run-time switch: a user
// holding the real rkmh can match it without rebuilding, and `mash` selects the variant the reference's README claims compatibility
// with (README.md:12; hash type and seed of the schema, src/rkmh.cpp:493-497): the first 64 bits of MurmurHash3_x64_128, seed 42,
// every one of the len - k + 1 windows.  Host code only.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/rkmh_amd.h"

extern "C" void rk__set_error(const char* msg);

namespace {

struct Choice { const char* key; const char* name; int value; };
// key=value pairs; the first name of a value is the one rk_policy_describe prints
const Choice CHOICES[] = {
    {"fold", "swap32", RK_FOLD_SWAP32}, {"fold", "h1", RK_FOLD_H1}, {"fold", "w2w1", RK_FOLD_W2W1},
    {"windows", "len-k", 1}, {"windows", "len-k+1", 0},
    {"zero", "count", 1}, {"zero", "skip", 0},     // U12: does the 6-argument calc_hashes count the 0 sentinel
    {"mask", "lt", 1}, {"mask", "le", 0},          // U9: mask_by_frequency zeroes count < min (lt) or count <= min (le)
    {"freqmax", "incl", 1}, {"freqmax", "excl", 0} // U10: minhashes_frequency_filter keeps count <= max (incl) or count < max
};
int32_t* field(rk_policy* p, const char* key) {
    if (!strcmp(key, "fold")) return &p->fold;
    if (!strcmp(key, "windows")) return &p->drop_last_window;
    if (!strcmp(key, "zero")) return &p->counter_counts_zero;
    if (!strcmp(key, "mask")) return &p->mask_strict_less;
    if (!strcmp(key, "freqmax")) return &p->freq_max_inclusive;
    return nullptr;
}
int bad(const std::string& msg) { rk__set_error(msg.c_str()); return RK_ERR_ARG; }

} // namespace

// spec: comma-separated items applied left to right onto *p (which the caller initialised, e.g. rk_default_policy): a preset
// (`default` = the build's defaults, `mash`) or key=value with the keys above, or seed=<n>.  Empty / NULL: nothing changes.
extern "C" int rk_policy_parse(const char* spec, rk_policy* p) {
    if (!p) return bad("policy is NULL");
    if (!spec) return RK_OK;
    std::string s(spec);
    size_t at = 0;
    while (at <= s.size()) {
        size_t end = s.find(',', at);
        if (end == std::string::npos) end = s.size();
        std::string item = s.substr(at, end - at);
        at = end + 1;
        while (!item.empty() && (item.front() == ' ' || item.front() == '\t')) item.erase(item.begin());
        while (!item.empty() && (item.back() == ' ' || item.back() == '\t')) item.pop_back();
        if (item.empty()) continue;
        if (item == "default") { rk_default_policy(p); continue; }
        if (item == "mash") { p->fold = RK_FOLD_H1; p->drop_last_window = 0; p->seed = 42; continue; }
        const size_t eq = item.find('=');
        if (eq == std::string::npos) return bad("hash policy: '" + item + "' is neither a preset (default, mash) nor key=value");
        const std::string key = item.substr(0, eq), val = item.substr(eq + 1);
        if (key == "seed") {
            char* e = nullptr;
            const unsigned long long v = strtoull(val.c_str(), &e, 0);
            if (val.empty() || *e || v > 0xFFFFFFFFull) return bad("hash policy: seed=" + val + " is not a 32-bit number");
            p->seed = (uint32_t)v;
            continue;
        }
        int32_t* f = field(p, key.c_str());
        if (!f) return bad("hash policy: unknown key '" + key + "' (fold, windows, zero, mask, freqmax, seed)");
        bool found = false;
        std::string names;
        for (const Choice& c : CHOICES) {
            if (key != c.key) continue;
            if (val == c.name) { *f = c.value; found = true; }
            names += names.empty() ? "" : "|";
            names += c.name;
        }
        if (!found) return bad("hash policy: " + key + "=" + val + " (expected " + key + "=" + names + ")");
    }
    return RK_OK;
}

// the canonical text of a policy, every key spelled out: "fold=swap32,windows=len-k,zero=count,mask=lt,freqmax=incl,seed=42".
// Returns the length written (excluding the NUL) or RK_ERR_ARG (a value outside the known ones, or cap too small).
extern "C" int rk_policy_describe(const rk_policy* p, char* dst, size_t cap) {
    if (!p || !dst) return bad("bad arguments");
    rk_policy q = *p;
    std::string out;
    for (const char* key : {"fold", "windows", "zero", "mask", "freqmax"}) {
        const int32_t v = *field(&q, key);
        const char* name = nullptr;
        for (const Choice& c : CHOICES)
            if (!strcmp(c.key, key) && c.value == v && !name) name = c.name;
        if (!name) return bad(std::string("hash policy: field '") + key + "' holds an unknown value");
        out += key; out += '='; out += name; out += ',';
    }
    out += "seed=" + std::to_string(p->seed);
    if (out.size() + 1 > cap) return bad("hash policy: buffer too small");
    memcpy(dst, out.c_str(), out.size() + 1);
    return (int)out.size();
}

// Do two policies give the same hash values and sketches (fold, window rule, seed)?  The other fields only act on depth counters.
extern "C" int rk_policy_same_hashes(const rk_policy* a, const rk_policy* b) {
    return a && b && a->fold == b->fold && (a->drop_last_window != 0) == (b->drop_last_window != 0) && a->seed == b->seed ? 1 : 0;
}
