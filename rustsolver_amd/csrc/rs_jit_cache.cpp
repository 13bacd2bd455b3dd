// rs_jit_cache.cpp -- the other half of the tree-specialised kernels (rs_jit.cpp writes their source): sources compiled for gfx950 with hipRTC -- the kernels of one plan that no
// cache holds by a handful of helper PROCESSES side by side (rs_jitc: hipRTC serialises compiles inside a process), or in this process when the helper is not there or fails --,
// code objects cached in the process and on disk (keyed by source + compiler version + options; a blob that does not load is thrown away and rebuilt), functions handed out per device.
#include <dirent.h>
#include <dlfcn.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "rs_internal.hpp"

extern char **environ;

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

Rtc *rtc_load() {
    static Rtc r;
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
Rtc *rtc() {   // called with and without g_mu held (jit_available, jit_compile_only): the load happens once
    static std::once_flag once;
    static Rtc *r = nullptr;
    std::call_once(once, [] { r = rtc_load(); });
    return r;
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
}  // namespace

bool jit_available() { return rtc() != nullptr; }

static const char *const kRtcOpts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"};
static constexpr int kNRtcOpts = int(sizeof(kRtcOpts) / sizeof(kRtcOpts[0]));

// what a cached code object depends on besides the source: the compiler (hipRTC version) and the options
static const std::string &cache_salt() {
    static std::string salt;
    static std::once_flag once;
    std::call_once(once, [] {
        int major = 0, minor = 0;
        if (Rtc *r = rtc()) {
            auto ver = (int (*)(int *, int *))dlsym(r->handle, "hiprtcVersion");
            if (ver) (void)ver(&major, &minor);
        }
        salt = "hiprtc " + std::to_string(major) + "." + std::to_string(minor);
        for (const char *o : kRtcOpts) salt += std::string(" ") + o;
    });
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

static std::string cache_path(const std::string &dir, const char *prefix, uint64_t h, const char *ext) {
    char name[80];
    snprintf(name, sizeof(name), "/%s_%016llx.%s", prefix, (unsigned long long)h, ext);
    return dir + name;
}

// ---- compiles side by side: helper processes ------------------------------------------------------------------------------------------------------------------
// rs_jitc sits next to this library (rustsolver_amd/build.py builds both).  Every helper gets its share of (source file, output file) pairs; whatever a helper did not
// produce -- it is missing, it crashed, a source does not compile -- is simply not there afterwards and the caller compiles it in this process, where the error gets reported.
static std::string helper_path() {
    Dl_info info;
    if (!dladdr((const void *)&helper_path, &info) || !info.dli_fname) return std::string();
    std::string p(info.dli_fname);
    const size_t slash = p.rfind('/');
    p = (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/rs_jitc";
    return access(p.c_str(), X_OK) == 0 ? p : std::string();
}
// a scratch directory made for one hand-over: whatever a helper left behind (a .tmp of a compile that was killed, a .log) goes with it
static void remove_scratch_dir(const std::string &dir) {
    if (DIR *d = opendir(dir.c_str())) {
        while (const dirent *e = readdir(d)) {
            const std::string name = e->d_name;
            if (name != "." && name != "..") (void)unlink((dir + "/" + name).c_str());
        }
        closedir(d);
    }
    (void)rmdir(dir.c_str());
}
static void compile_in_processes(const std::vector<std::pair<std::string, std::string>> &jobs /* (source file, output file) */) {
    const std::string exe = helper_path();
    if (exe.empty() || jobs.empty()) return;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t n_proc = std::min<size_t>({jobs.size(), size_t(hw), size_t(16)});
    std::vector<pid_t> pids;
    for (size_t k = 0; k < n_proc; ++k) {
        std::vector<std::string> args{exe};   // the compiler options and the hipRTC version this library keys its cache with travel along (rs_jitc.cpp)
        {
            int major = 0, minor = 0;
            if (Rtc *r = rtc()) {
                auto ver = (int (*)(int *, int *))dlsym(r->handle, "hiprtcVersion");
                if (ver) (void)ver(&major, &minor);
            }
            args.push_back("--rtc");
            args.push_back(std::to_string(major) + "." + std::to_string(minor));
            for (const char *o : kRtcOpts) {
                args.push_back("--opt");
                args.push_back(o);
            }
            args.push_back("--");
        }
        for (size_t i = k; i < jobs.size(); i += n_proc) {
            args.push_back(jobs[i].first);
            args.push_back(jobs[i].second);
        }
        std::vector<char *> argv;
        for (std::string &a : args) argv.push_back(&a[0]);
        argv.push_back(nullptr);
        pid_t pid = 0;
        if (posix_spawn(&pid, exe.c_str(), nullptr, nullptr, argv.data(), environ) == 0) pids.push_back(pid);
    }
    for (pid_t pid : pids) {
        int status = 0;
        while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
    }
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

// Many kernels at once (a solver's whole plan).  hipRTC compiles one program at a time inside a process (36 kernels of a three-street deal plan: 44 s of CPU in 49 s of wall
// clock on a pool of threads), so the sources that no cache holds are written out and compiled by helper processes side by side (rs_jitc, above); jit_get_kernel then finds
// their code objects on disk.  What the helpers did not deliver is compiled right here by the same loop, one by one.
int jit_get_kernels(std::vector<JitRequest> &reqs, int device, bool dump) {
    {
        std::lock_guard<std::mutex> lock(g_mu);
        std::string dir = jit_cache_dir();
        std::vector<std::pair<uint64_t, const std::string *>> todo;
        std::set<uint64_t> seen;
        for (const JitRequest &r : reqs) {
            const uint64_t h = jit_source_key(*r.source);
            if (g_loaded.count({device, h}) || g_code.count(h) || !seen.insert(h).second) continue;
            if (!dir.empty() && access(cache_path(dir, "tree", h, "hsaco").c_str(), R_OK) == 0) continue;
            todo.emplace_back(h, r.source);
        }
        if (todo.size() >= 2 && !knobs_resolve(nullptr).jit_no_procs) {
            const bool scratch = dir.empty();   // no disk cache wanted: the helpers still need a place to hand their code objects over
            if (scratch) {
                char tmpl[] = "/tmp/rs_jit_XXXXXX";
                if (const char *d = mkdtemp(tmpl)) dir = d;
            } else {
                (void)mkdir(dir.substr(0, dir.rfind('/')).c_str(), 0755);
                (void)mkdir(dir.c_str(), 0755);
            }
            if (!dir.empty()) {
                std::vector<std::pair<std::string, std::string>> jobs;
                for (auto &t : todo) {
                    const std::string src = cache_path(dir, "tree", t.first, ("src." + std::to_string(getpid())).c_str());
                    write_file_atomic(dir, src, std::vector<char>(t.second->begin(), t.second->end()));
                    jobs.emplace_back(src, cache_path(dir, "tree", t.first, "hsaco"));
                }
                compile_in_processes(jobs);
                for (size_t i = 0; i < jobs.size(); ++i) {
                    (void)unlink(jobs[i].first.c_str());
                    (void)unlink((jobs[i].second + ".log").c_str());   // the in-process compile below reports the error
                    if (scratch) {
                        std::vector<char> buf;
                        if (read_file(jobs[i].second, buf) && !buf.empty()) g_code[todo[i].first] = std::move(buf);
                        (void)unlink(jobs[i].second.c_str());
                    }
                }
                if (scratch) remove_scratch_dir(dir);
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
    // the CPU suite's "does every generated form compile": helper processes side by side like jit_get_kernels; a source a helper did not deliver is compiled in this process,
    // which reports the compiler's words
    if (sources.empty()) return RS_OK;
    if (!rtc()) return fail(RS_ERR_UNSUPPORTED, "libhiprtc.so could not be loaded");
    std::vector<const std::string *> todo;
    for (const auto &kv : sources) todo.push_back(&kv.first);
    std::vector<char> done(todo.size(), 0);
    if (todo.size() >= 2 && !knobs_resolve(nullptr).jit_no_procs) {
        char tmpl[] = "/tmp/rs_jit_XXXXXX";
        if (const char *d = mkdtemp(tmpl)) {
            const std::string dir(d);
            std::vector<std::pair<std::string, std::string>> jobs;
            for (size_t i = 0; i < todo.size(); ++i) {
                const std::string src = cache_path(dir, "chk", uint64_t(i), "hip");
                write_file_atomic(dir, src, std::vector<char>(todo[i]->begin(), todo[i]->end()));
                jobs.emplace_back(src, cache_path(dir, "chk", uint64_t(i), "hsaco"));
            }
            compile_in_processes(jobs);
            for (size_t i = 0; i < jobs.size(); ++i) {
                done[i] = access(jobs[i].second.c_str(), R_OK) == 0;
                (void)unlink(jobs[i].first.c_str());
                (void)unlink(jobs[i].second.c_str());
                (void)unlink((jobs[i].second + ".log").c_str());
            }
            remove_scratch_dir(dir);
        }
    }
    for (size_t i = 0; i < todo.size(); ++i) {
        if (dump) {
            char p[96];
            snprintf(p, sizeof(p), "/tmp/rs_tree_kernel_%016llx.hip", (unsigned long long)fnv1a(*todo[i]));
            if (FILE *f = fopen(p, "w")) {
                fputs(todo[i]->c_str(), f);
                fclose(f);
            }
        }
        if (!done[i])
            if (int rc = jit_compile_only(*todo[i], false)) return rc;
    }
    return RS_OK;
}
}  // namespace rs
