// rs_jitc.cpp -- the compile helper of rs_jit_cache.cpp: a small program that compiles HIP sources for gfx950 with hipRTC and writes the code objects to files.
//
//     rs_jitc [--rtc <major.minor>] [--opt <compiler option>]... [--] <source file> <output file> [<source file> <output file> ...]
//
// The library hands over ITS option list (--opt, one each; part of its cache key) and the hipRTC version it keyed the cache with (--rtc): a helper that finds another hipRTC
// through its own dlopen search refuses (exit 104) instead of filing another compiler's code under the library's key -- the sweeps are bit-exact contracts
// (-ffp-contract=off, -fno-fast-math).  Without --opt (the CPU test calls the helper by hand) the defaults below apply.
//
// hipRTC serialises compiles inside one process (36 kernels of a three-street deal plan: 44 s of CPU in 49 s of wall clock, whatever the number of threads), separate processes do
// not: the library writes the sources no cache holds to files and starts a handful of these, each with its share.  No GPU is touched (hipRTC cross-compiles), nothing of the
// library is linked.  An output is written under a temporary name and renamed; a failed compile leaves `<output>.log` with the compiler's words and counts in the exit code.
#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef struct _hiprtcProgram *hiprtcProgram;

static bool read_text(const char *path, std::string &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

int main(int argc, char **argv) {
    std::vector<const char *> given;
    std::string want_rtc;
    int first = 1;
    while (first + 1 < argc && argv[first][0] == '-' && argv[first][1] == '-') {
        const std::string a = argv[first];
        if (a == "--") { ++first; break; }
        if (a == "--opt") given.push_back(argv[first + 1]);
        else if (a == "--rtc") want_rtc = argv[first + 1];
        else break;
        first += 2;
    }
    if (first < argc && std::string(argv[first]) == "--") ++first;
    if (argc - first < 2 || (argc - first) % 2) {
        fprintf(stderr, "usage: rs_jitc [--rtc <major.minor>] [--opt <option>]... [--] <source> <output> [<source> <output> ...]\n");
        return 100;
    }
    if (getenv("RS_JITC_SELFTEST_FAIL")) return 103;   // test hook (tests/test_jit_cpu.py): a helper that delivers nothing -- the library must then compile in its own process
    void *h = nullptr;
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        fprintf(stderr, "rs_jitc: libhiprtc.so could not be loaded\n");
        return 101;
    }
    auto Create = (int (*)(hiprtcProgram *, const char *, const char *, int, const char **, const char **))dlsym(h, "hiprtcCreateProgram");
    auto Compile = (int (*)(hiprtcProgram, int, const char **))dlsym(h, "hiprtcCompileProgram");
    auto LogSize = (int (*)(hiprtcProgram, size_t *))dlsym(h, "hiprtcGetProgramLogSize");
    auto Log = (int (*)(hiprtcProgram, char *))dlsym(h, "hiprtcGetProgramLog");
    auto CodeSize = (int (*)(hiprtcProgram, size_t *))dlsym(h, "hiprtcGetCodeSize");
    auto Code = (int (*)(hiprtcProgram, char *))dlsym(h, "hiprtcGetCode");
    auto Destroy = (int (*)(hiprtcProgram *))dlsym(h, "hiprtcDestroyProgram");
    if (!Create || !Compile || !LogSize || !Log || !CodeSize || !Code || !Destroy) return 102;
    if (!want_rtc.empty()) {   // the library's cache key names a hipRTC version: this process must have found the same one
        auto Version = (int (*)(int *, int *))dlsym(h, "hiprtcVersion");
        int major = 0, minor = 0;
        if (!Version || Version(&major, &minor) != 0 || want_rtc != std::to_string(major) + "." + std::to_string(minor)) {
            fprintf(stderr, "rs_jitc: hipRTC %d.%d here, the caller keyed its cache with %s\n", major, minor, want_rtc.c_str());
            return 104;
        }
    }
    static const char *const defaults[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"};   // a call by hand (tests); the library passes its own
    std::vector<const char *> opts_v = given;
    if (opts_v.empty()) opts_v.assign(defaults, defaults + sizeof(defaults) / sizeof(defaults[0]));
    const char **opts = opts_v.data();
    const int n_opts = int(opts_v.size());
    int failed = 0;
    for (int i = first; i + 1 < argc; i += 2) {
        std::string src;
        const std::string out = argv[i + 1];
        if (!read_text(argv[i], src)) {
            ++failed;
            continue;
        }
        hiprtcProgram prog = nullptr;
        bool ok = Create(&prog, src.c_str(), "rs_tree_kernel.hip", 0, nullptr, nullptr) == 0 && Compile(prog, n_opts, opts) == 0;
        if (ok) {
            size_t n = 0;
            CodeSize(prog, &n);
            std::vector<char> code(n);
            Code(prog, code.data());
            const std::string tmp = out + "." + std::to_string(getpid()) + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            ok = f && n > 0 && fwrite(code.data(), 1, n, f) == n;
            if (f) fclose(f);
            if (ok) ok = rename(tmp.c_str(), out.c_str()) == 0;
            else (void)unlink(tmp.c_str());
        } else if (prog) {
            size_t n = 0;
            LogSize(prog, &n);
            std::string log(n + 1, '\0');
            if (n) Log(prog, &log[0]);
            if (FILE *f = fopen((out + ".log").c_str(), "w")) {
                fputs(log.c_str(), f);
                fclose(f);
            }
        }
        if (prog) Destroy(&prog);
        if (!ok) ++failed;
    }
    return failed > 99 ? 99 : failed;
}
