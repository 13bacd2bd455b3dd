// rs_jitc.cpp -- the compile helper of rs_jit_cache.cpp: a small program that compiles HIP sources for gfx950 with hipRTC and writes the code objects to files.
//
//     rs_jitc <source file> <output file> [<source file> <output file> ...]
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
    if (argc < 3 || (argc - 1) % 2) {
        fprintf(stderr, "usage: rs_jitc <source> <output> [<source> <output> ...]\n");
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
    // the options of rs_jit_cache.cpp kRtcOpts (part of the cache key there): keep the two lists the same
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"};
    int failed = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string src;
        const std::string out = argv[i + 1];
        if (!read_text(argv[i], src)) {
            ++failed;
            continue;
        }
        hiprtcProgram prog = nullptr;
        bool ok = Create(&prog, src.c_str(), "rs_tree_kernel.hip", 0, nullptr, nullptr) == 0 && Compile(prog, int(sizeof(opts) / sizeof(opts[0])), opts) == 0;
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
