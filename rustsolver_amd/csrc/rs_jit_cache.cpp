// rs_jit_cache.cpp -- the other half of the tree-specialised kernels (rs_jit.cpp writes their source): hipRTC loaded at run time, sources compiled for gfx950 --
// one by one or, for the kernels of one plan, as groups that share the prelude's parse --, code objects cached in the process and on disk (keyed by source + compiler version +
// options; a blob that does not load is thrown away and rebuilt), functions handed out per device.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "rs_internal.hpp"

namespace rs {

namespace {

typedef struct _hiprtcProgram *hiprtcProgram;
struct Rtc {
    void *handle = nullptr;
    int (*CreateProgram)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*CompileProgram)(hiprtcProgram, int, const char **) = nullptr;
    int (*GetProgramLogSize)(hiprtcProgram, size_t *) = nullptr;
    int (*GetProgramLog)(hiprtcProgram, char *) = nullptr;
    int (*GetCodeSize)(hiprtcProgram, size_t *) = nullptr;
    int (*GetCode)(hiprtcProgram, char *) = nullptr;
    int (*DestroyProgram)(hiprtcProgram *) = nullptr;
};

Rtc *rtc() {
    static Rtc r;
    static bool tried = false;
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) return nullptr;
#define RS_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.handle, sym)
    RS_SYM(CreateProgram, "hiprtcCreateProgram");
    RS_SYM(CompileProgram, "hiprtcCompileProgram");
    RS_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
    RS_SYM(GetProgramLog, "hiprtcGetProgramLog");
    RS_SYM(GetCodeSize, "hiprtcGetCodeSize");
    RS_SYM(GetCode, "hiprtcGetCode");
    RS_SYM(DestroyProgram, "hiprtcDestroyProgram");
#undef RS_SYM
    if (!r.CreateProgram || !r.CompileProgram || !r.GetProgramLogSize || !r.GetProgramLog || !r.GetCodeSize || !r.GetCode ||
        !r.DestroyProgram) {
        dlclose(r.handle);
        r.handle = nullptr;
        return nullptr;
    }
    return &r;
}

uint64_t fnv1a(const std::string &s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

bool read_file(const std::string &path, std::vector<char> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? size_t(n) : 0);
    const bool ok = n > 0 && fread(out.data(), 1, size_t(n), f) == size_t(n);
    fclose(f);
    return ok;
}

void write_file_atomic(const std::string &dir, const std::string &path, const std::vector<char> &data) {
    (void)mkdir(dir.substr(0, dir.rfind('/')).c_str(), 0755);
    (void)mkdir(dir.c_str(), 0755);
    const std::string tmp = path + "." + std::to_string(getpid()) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    fclose(f);
    if (ok) (void)rename(tmp.c_str(), path.c_str());
    else (void)unlink(tmp.c_str());
}

struct Loaded {
    hipModule_t mod;
    hipFunction_t fn;
};
std::mutex g_mu;
std::map<std::pair<int, uint64_t>, Loaded> g_loaded;   // (device, source hash)
std::map<uint64_t, std::vector<char>> g_code;          // source hash -> code object
// Kernels that were compiled together (jit_get_kernels: one hipRTC program per group of sources with the same prelude, the prelude parsed once) live in ONE code object:
// group_<H>.hsaco on disk, and per member a tree_<h>.ref file naming the group and the member's entry point inside it
struct GroupRef {
    uint64_t group;
    std::string entry;
};
std::map<uint64_t, GroupRef> g_ref;                              // source hash -> where its kernel lives
std::map<std::pair<int, uint64_t>, hipModule_t> g_group_mod;    // (device, group hash) -> loaded module

}  // namespace

bool jit_available() { return rtc() != nullptr; }

static const char *const kRtcOpts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"};
static constexpr int kNRtcOpts = int(sizeof(kRtcOpts) / sizeof(kRtcOpts[0]));

// what a cached code object depends on besides the source: the compiler (hipRTC version) and the options
static const std::string &cache_salt() {
    static std::string salt;
    static bool done = false;
    if (done) return salt;
    done = true;
    int major = 0, minor = 0;
    if (Rtc *r = rtc()) {
        auto ver = (int (*)(int *, int *))dlsym(r->handle, "hiprtcVersion");
        if (ver) (void)ver(&major, &minor);
    }
    salt = "hiprtc " + std::to_string(major) + "." + std::to_string(minor);
    for (const char *o : kRtcOpts) salt += std::string(" ") + o;
    return salt;
}

static int compile_source(const std::string &source, std::vector<char> &buf, bool dump) {
    Rtc *r = rtc();
    if (!r) return fail(RS_ERR_UNSUPPORTED, "tree-specialised kernels need libhiprtc.so, which could not be loaded");
    hiprtcProgram prog = nullptr;
    if (r->CreateProgram(&prog, source.c_str(), "rs_tree_kernel.hip", 0, nullptr, nullptr) != 0)
        return fail(RS_ERR_HIP, "hiprtcCreateProgram failed");
    const char *opts[kNRtcOpts];
    for (int i = 0; i < kNRtcOpts; ++i) opts[i] = kRtcOpts[i];
    const int rc = r->CompileProgram(prog, kNRtcOpts, opts);
    if (rc != 0) {
        size_t n = 0;
        r->GetProgramLogSize(prog, &n);
        std::string log(n + 1, '\0');
        if (n) r->GetProgramLog(prog, &log[0]);
        r->DestroyProgram(&prog);
        if (dump) {
            FILE *f = fopen("/tmp/rs_tree_kernel_failed.hip", "w");
            if (f) {
                fputs(source.c_str(), f);
                fclose(f);
            }
        }
        return fail(RS_ERR_HIP, "hipRTC compile failed: " + log.substr(0, 4000));
    }
    size_t n = 0;
    r->GetCodeSize(prog, &n);
    buf.resize(n);
    r->GetCode(prog, buf.data());
    r->DestroyProgram(&prog);
    return RS_OK;
}

// Groups of sources that can share one hipRTC program: the same text in front of the prelude (the #defines that shape it), at most 24 members (a bound on the size of
// one program), every member's own part inside its own namespace; entry points that occur more than once in a group (one form, several subtree shapes) become `<entry>__s<i>`.  Sources that fit no group of two are left out.
struct KernelGroup {
    std::string combined;
    std::vector<std::string> names;     // per member: its entry point inside the group
    std::vector<size_t> members;        // indices into the input
};
static std::vector<KernelGroup> group_sources(const std::vector<std::pair<const std::string *, const std::string *>> &items /* (source, entry) */) {
    const std::string prelude(jit_device_source());
    std::map<std::string, std::vector<size_t>> by_defs;
    for (size_t i = 0; i < items.size(); ++i) {
        const size_t at = items[i].first->find(prelude);
        if (at != std::string::npos) by_defs[items[i].first->substr(0, at)].push_back(i);
    }
    std::vector<KernelGroup> out;
    for (auto &kv : by_defs)
        for (size_t lo = 0; lo < kv.second.size(); lo += 24) {
            const size_t hi = std::min(kv.second.size(), lo + 24);
            if (hi - lo < 2) continue;
            KernelGroup G;
            G.combined = kv.first + prelude;
            bool ok = true;
            std::map<std::string, int> total, seen_n;   // an entry point keeps its name unless the group holds several kernels of that name (same form, different subtree shapes)
            for (size_t k = lo; k < hi; ++k) total[*items[kv.second[k]].second] += 1;
            for (size_t k = lo; k < hi && ok; ++k) {
                const std::string &src = *items[kv.second[k]].first, &entry = *items[kv.second[k]].second;
                std::string body = src.substr(kv.first.size() + prelude.size());
                const std::string from = "void " + entry + "(", name = total[entry] > 1 ? entry + "__s" + std::to_string(seen_n[entry]++) : entry;
                const size_t at = body.find(from);
                ok = at != std::string::npos && body.find(from, at + 1) == std::string::npos;
                if (!ok) break;
                body.replace(at, from.size(), "void " + name + "(");
                G.combined += "\nnamespace rs_g" + std::to_string(k - lo) + " {\n" + body + "\n}\n";
                G.names.push_back(name);
                G.members.push_back(kv.second[k]);
            }
            if (ok) out.push_back(std::move(G));
        }
    return out;
}

static std::string cache_path(const std::string &dir, const char *prefix, uint64_t h, const char *ext) {
    char name[80];
    snprintf(name, sizeof(name), "/%s_%016llx.%s", prefix, (unsigned long long)h, ext);
    return dir + name;
}

// the kernel of source hash `h` from its group's code object (memory, else disk); false = no usable group (a stale reference or blob is removed)
static bool load_from_group(int device, uint64_t h, Loaded *out) {
    const std::string dir = jit_cache_dir();
    auto rit = g_ref.find(h);
    if (rit == g_ref.end()) {
        if (dir.empty()) return false;
        std::vector<char> txt;
        if (!read_file(cache_path(dir, "tree", h, "ref"), txt) || txt.empty()) return false;
        unsigned long long H = 0;
        char entry[256];
        entry[0] = 0;
        txt.push_back(0);
        if (sscanf(txt.data(), "%llx %255s", &H, entry) != 2) {
            (void)unlink(cache_path(dir, "tree", h, "ref").c_str());
            return false;
        }
        rit = g_ref.emplace(h, GroupRef{uint64_t(H), std::string(entry)}).first;
    }
    const uint64_t H = rit->second.group;
    auto mit = g_group_mod.find({device, H});
    if (mit == g_group_mod.end()) {
        std::vector<char> blob;
        hipModule_t mod = nullptr;
        const std::string gpath = dir.empty() ? std::string() : cache_path(dir, "group", H, "hsaco");
        const bool have = !gpath.empty() && read_file(gpath, blob) && !blob.empty();
        if (!have || hipSetDevice(device) != hipSuccess || hipModuleLoadData(&mod, blob.data()) != hipSuccess) {
            (void)hipGetLastError();
            if (!gpath.empty()) (void)unlink(gpath.c_str());
            if (!dir.empty()) (void)unlink(cache_path(dir, "tree", h, "ref").c_str());
            g_ref.erase(h);
            return false;
        }
        mit = g_group_mod.emplace(std::make_pair(device, H), mod).first;
    }
    hipFunction_t fn = nullptr;
    if (hipModuleGetFunction(&fn, mit->second, rit->second.entry.c_str()) != hipSuccess) {
        (void)hipGetLastError();
        if (!dir.empty()) (void)unlink(cache_path(dir, "tree", h, "ref").c_str());
        g_ref.erase(h);
        return false;
    }
    *out = Loaded{mit->second, fn};
    return true;
}

// Compiles `source` (entry point `entry`) for gfx950, or fetches it from the caches, and returns a function handle valid on `device`.  The disk cache
// is keyed by source + hipRTC version + options; a cached blob that does not load (stale, truncated, foreign) is deleted and compiled again once.
int jit_get_kernel(const std::string &source, const std::string &entry, int device, hipFunction_t *fn, bool dump) {
    std::lock_guard<std::mutex> lock(g_mu);
    const uint64_t h = jit_source_key(source);
    auto it = g_loaded.find({device, h});
    if (it != g_loaded.end()) {
        *fn = it->second.fn;
        return RS_OK;
    }
    char name[64];
    snprintf(name, sizeof(name), "/tree_%016llx.hsaco", (unsigned long long)h);
    const std::string dir = jit_cache_dir(), path = dir + name;
    bool from_disk = false;
    if (g_code.find(h) == g_code.end()) {   // a kernel compiled as a member of a group (jit_get_kernels) lives in the group's code object
        std::vector<char> own;
        Loaded G{};
        if (!(!dir.empty() && read_file(path, own) && !own.empty()) && load_from_group(device, h, &G)) {
            g_loaded[{device, h}] = G;
            *fn = G.fn;
            return RS_OK;
        }
    }
    if (g_code.find(h) == g_code.end()) {
        std::vector<char> buf;
        if (!dir.empty() && read_file(path, buf) && !buf.empty()) from_disk = true;
        else {
            if (int rc = compile_source(source, buf, dump)) return rc;
            if (!dir.empty()) write_file_atomic(dir, path, buf);
        }
        g_code[h] = std::move(buf);
    }
    if (dump) {
        char p[96];
        snprintf(p, sizeof(p), "/tmp/rs_tree_kernel_%016llx.hip", (unsigned long long)h);
        FILE *f = fopen(p, "w");
        if (f) {
            fputs(source.c_str(), f);
            fclose(f);
        }
    }
    Loaded L{};
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    for (int attempt = 0;; ++attempt) {
        L = Loaded{};
        e = hipModuleLoadData(&L.mod, g_code[h].data());
        if (e == hipSuccess) {
            e = hipModuleGetFunction(&L.fn, L.mod, entry.c_str());
            if (e != hipSuccess) (void)hipModuleUnload(L.mod);
        }
        if (e == hipSuccess) break;
        if (!from_disk || attempt > 0) {
            g_code.erase(h);
            return hip_fail(e, "loading a tree-specialised kernel");
        }
        (void)hipGetLastError();
        (void)unlink(path.c_str());        // the cached blob is unusable: compile it afresh and rewrite the file
        std::vector<char> buf;
        if (int rc = compile_source(source, buf, dump)) {
            g_code.erase(h);
            return rc;
        }
        write_file_atomic(dir, path, buf);
        g_code[h] = std::move(buf);
    }
    g_loaded[{device, h}] = L;
    *fn = L.fn;
    return RS_OK;
}

uint64_t jit_source_key(const std::string &source) { return fnv1a(source + "\n// " + cache_salt()); }

// Many kernels at once (a solver's whole plan).  hipRTC compiles one program at a time inside a process, and a third of a kernel's 1.2 s goes into parsing the shared
// prelude (rs_device.hpp): the sources no cache holds are therefore compiled TOGETHER, one program per group of sources with the same prelude -- every member's own part inside
// its own namespace -- and the group's code object serves all of them.
int jit_get_kernels(std::vector<JitRequest> &reqs, int device, bool dump) {
    struct Todo {
        uint64_t h;
        const std::string *source, *entry;
    };
    std::vector<Todo> todo;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        const std::string dir = jit_cache_dir();
        std::set<uint64_t> seen;
        for (const JitRequest &r : reqs) {
            const uint64_t h = jit_source_key(*r.source);
            if (g_loaded.count({device, h}) || g_code.count(h) || !seen.insert(h).second) continue;
            std::vector<char> buf;
            if (!dir.empty() && read_file(cache_path(dir, "tree", h, "hsaco"), buf) && !buf.empty()) continue;   // the one-kernel path will pick the file up
            Loaded L{};
            if (load_from_group(device, h, &L)) {
                g_loaded[{device, h}] = L;
                continue;
            }
            todo.push_back(Todo{h, r.source, r.entry});
        }
        std::vector<std::pair<const std::string *, const std::string *>> items;
        for (const Todo &t : todo) items.emplace_back(t.source, t.entry);
        for (const KernelGroup &G : group_sources(items)) {
            std::vector<char> blob;
            if (int rc = compile_source(G.combined, blob, dump)) return rc;
            const uint64_t H = fnv1a(G.combined + "\n// " + cache_salt());
            hipModule_t mod = nullptr;
            hipError_t e = hipSetDevice(device);
            if (e == hipSuccess) e = hipModuleLoadData(&mod, blob.data());
            if (e != hipSuccess) return hip_fail(e, "loading a group of tree-specialised kernels");
            g_group_mod[{device, H}] = mod;
            if (!dir.empty()) write_file_atomic(dir, cache_path(dir, "group", H, "hsaco"), blob);
            for (size_t k = 0; k < G.members.size(); ++k) {
                const Todo &t = todo[G.members[k]];
                Loaded L{mod, nullptr};
                e = hipModuleGetFunction(&L.fn, mod, G.names[k].c_str());
                if (e != hipSuccess) return hip_fail(e, "a kernel missing from its group's code object");
                g_loaded[{device, t.h}] = L;
                g_ref[t.h] = GroupRef{H, G.names[k]};
                if (!dir.empty()) {
                    char line[320];
                    const int n = snprintf(line, sizeof(line), "%016llx %s\n", (unsigned long long)H, G.names[k].c_str());
                    write_file_atomic(dir, cache_path(dir, "tree", t.h, "ref"), std::vector<char>(line, line + n));
                }
                if (dump) {
                    char p[96];
                    snprintf(p, sizeof(p), "/tmp/rs_tree_kernel_%016llx.hip", (unsigned long long)t.h);
                    if (FILE *f = fopen(p, "w")) {
                        fputs(t.source->c_str(), f);
                        fclose(f);
                    }
                }
            }
        }
    }
    for (JitRequest &r : reqs)
        if (int rc = jit_get_kernel(*r.source, *r.entry, device, &r.fn, dump)) return rc;
    return RS_OK;
}

// compile only (no device needed): used by the CPU test that checks every generated source builds
int jit_compile_only(const std::string &source, bool dump) {
    if (dump) {
        char p[96];
        snprintf(p, sizeof(p), "/tmp/rs_tree_kernel_%016llx.hip", (unsigned long long)fnv1a(source));
        FILE *f = fopen(p, "w");
        if (f) {
            fputs(source.c_str(), f);
            fclose(f);
        }
    }
    Rtc *r = rtc();
    if (!r) return fail(RS_ERR_UNSUPPORTED, "libhiprtc.so could not be loaded");
    hiprtcProgram prog = nullptr;
    if (r->CreateProgram(&prog, source.c_str(), "rs_tree_kernel.hip", 0, nullptr, nullptr) != 0)
        return fail(RS_ERR_HIP, "hiprtcCreateProgram failed");
    const char *opts[kNRtcOpts];
    for (int i = 0; i < kNRtcOpts; ++i) opts[i] = kRtcOpts[i];
    const int rc = r->CompileProgram(prog, kNRtcOpts, opts);
    std::string log;
    if (rc != 0) {
        size_t n = 0;
        r->GetProgramLogSize(prog, &n);
        log.assign(n + 1, '\0');
        if (n) r->GetProgramLog(prog, &log[0]);
    }
    r->DestroyProgram(&prog);
    if (rc != 0) return fail(RS_ERR_HIP, "hipRTC compile failed: " + log.substr(0, 4000));
    return RS_OK;
}

int jit_compile_many(const std::map<std::string, int> &sources, bool dump) {
    // the way jit_get_kernels builds them: sources with the same prelude share one program (hipRTC compiles one program at a time inside a process; a third of a
    // kernel's compile is the prelude's parse), the rest one by one
    std::vector<const std::string *> todo;
    std::vector<std::string> entries;
    for (const auto &kv : sources) {
        todo.push_back(&kv.first);
        std::string entry;
        const size_t g = kv.first.find("extern \"C\" __global__");
        const size_t v = g == std::string::npos ? g : kv.first.find(") void ", g);
        if (v != std::string::npos) {
            const size_t b = v + 7, e = kv.first.find('(', b);
            if (e != std::string::npos) entry = kv.first.substr(b, e - b);
        }
        entries.push_back(entry);
    }
    if (todo.empty()) return RS_OK;
    if (!rtc()) return fail(RS_ERR_UNSUPPORTED, "libhiprtc.so could not be loaded");
    std::vector<std::pair<const std::string *, const std::string *>> items;
    for (size_t i = 0; i < todo.size(); ++i) items.emplace_back(todo[i], &entries[i]);
    std::vector<char> grouped(todo.size(), 0);
    for (const KernelGroup &G : group_sources(items)) {
        if (int rc = jit_compile_only(G.combined, false)) return rc;
        for (size_t m : G.members) grouped[m] = 1;
        if (dump)
            for (size_t m : G.members) {
                char p[96];
                snprintf(p, sizeof(p), "/tmp/rs_tree_kernel_%016llx.hip", (unsigned long long)fnv1a(*todo[m]));
                if (FILE *f = fopen(p, "w")) {
                    fputs(todo[m]->c_str(), f);
                    fclose(f);
                }
            }
    }
    for (size_t i = 0; i < todo.size(); ++i)
        if (!grouped[i])
            if (int rc = jit_compile_only(*todo[i], dump)) return rc;
    return RS_OK;
}
}  // namespace rs
