/* tests/stub_rccl.c -- TEST INFRASTRUCTURE, not part of the product.
 *
 * A stand-in for the five RCCL entry points csrc/rs_comm.cpp binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclAllGather, ncclGetErrorString),
 * for N PROCESSES THAT SHARE ONE GPU.  RCCL itself refuses two ranks on one device, and the GPU boxes of this pool hold one card, so the multi-rank code paths of
 * rs_comm.cpp / rs_solver.cpp / rs_trainer.cpp (slot offsets of the in-place all-gather, which stream a phase-1 kernel waits on, the wrapping ncclInt32 sums) could only ever
 * run with one rank, where every collective is a no-op.  With RS_RCCL_LIB pointing at this library (read in csrc/rs_knobs.cpp, test-only) the same calls run with 2, 3, ... real
 * processes: tests/test_gpu_multiproc.py compares their tables with a single process's.
 *
 * Transport: one POSIX shared-memory segment per communicator (its name is the "unique id"): a header with a sense-reversing barrier and one slot of kSlotBytes per rank.
 * A collective = for every chunk: hipMemcpyAsync device -> own slot on the CALLER'S stream, hipStreamSynchronize, barrier, every rank combines all slots in rank order on the
 * host (sum as wrapping 32-bit integers or as floats; gather = copy), barrier, hipMemcpyAsync host -> device on the caller's stream.  The call returns with the result
 * enqueued on the stream, like RCCL: whatever the caller launches next on that stream sees it; work the caller left on OTHER streams is NOT waited for -- exactly the
 * ordering contract of the real library, which is what the test is there to check.
 *
 *   gcc -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stub_rccl.c -o tests/libstub_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -lpthread
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

enum { kSlotBytes = 8 << 20, kMaxRanks = 16, kIdBytes = 128 };

typedef struct {
    volatile int ready;          /* set by the creator once the header is initialised */
    volatile int arrived;        /* barrier: ranks that reached it in this generation */
    volatile int generation;
    volatile int n_ranks;
    volatile int attached;
} Header;

typedef struct ncclComm {
    Header *hdr;
    char *slots;                 /* [n_ranks][kSlotBytes] */
    char *result;                /* process-local pinned staging for the combined chunk: [n_ranks][kSlotBytes] (an all-gather stages every rank's slot) */
    hipEvent_t ev;               /* recorded behind the last host -> device copy out of `result`: the staging is not rewritten before it has been read */
    int ev_set;
    size_t map_bytes;
    int rank, n_ranks;
    char name[64];
} ncclComm;
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[kIdBytes]; } ncclUniqueId;

static const char *g_err = "stub_rccl: ok";

const char *ncclGetErrorString(int code) { return code == 0 ? "no error" : g_err; }

int ncclGetUniqueId(ncclUniqueId *id) {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id->internal, 0, kIdBytes);
    snprintf(id->internal, 48, "/rs_stub_%d_%lx%lx", (int)getpid(), (unsigned long)ts.tv_sec, (unsigned long)ts.tv_nsec);
    return 0;
}

static void barrier(ncclComm *c) {
    Header *h = c->hdr;
    const int gen = __atomic_load_n(&h->generation, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&h->arrived, 1, __ATOMIC_ACQ_REL) == c->n_ranks) {
        __atomic_store_n(&h->arrived, 0, __ATOMIC_RELAXED);
        __atomic_add_fetch(&h->generation, 1, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n(&h->generation, __ATOMIC_ACQUIRE) == gen) sched_yield();
    }
}

int ncclCommInitRank(ncclComm_t *out, int n_ranks, ncclUniqueId id, int rank) {
    if (n_ranks < 1 || n_ranks > kMaxRanks || rank < 0 || rank >= n_ranks) { g_err = "stub_rccl: bad rank / n_ranks"; return 1; }
    ncclComm *c = (ncclComm *)calloc(1, sizeof(ncclComm));
    if (!c) { g_err = "stub_rccl: out of memory"; return 1; }
    memcpy(c->name, id.internal, sizeof(c->name) - 1);
    if (c->name[0] != '/') { g_err = "stub_rccl: the unique id is not a shared-memory name"; free(c); return 1; }
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->map_bytes = 4096 + (size_t)n_ranks * kSlotBytes;
    int creator = 1;
    int fd = shm_open(c->name, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 && errno == EEXIST) {
        creator = 0;
        fd = shm_open(c->name, O_RDWR, 0600);
    }
    if (fd < 0) { g_err = "stub_rccl: shm_open failed"; free(c); return 1; }
    if (creator && ftruncate(fd, (off_t)c->map_bytes) != 0) { g_err = "stub_rccl: ftruncate failed"; close(fd); free(c); return 1; }
    if (!creator) {   /* wait until the creator has sized the segment */
        struct stat st;
        for (int spin = 0; spin < 200000; ++spin) {
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) break;
            usleep(100);
        }
    }
    void *p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { g_err = "stub_rccl: mmap failed"; free(c); return 1; }
    c->hdr = (Header *)p;
    c->slots = (char *)p + 4096;
    if (creator) {
        c->hdr->arrived = 0;
        c->hdr->generation = 0;
        c->hdr->n_ranks = n_ranks;
        c->hdr->attached = 0;
        __atomic_store_n(&c->hdr->ready, 1, __ATOMIC_RELEASE);
    } else {
        while (!__atomic_load_n(&c->hdr->ready, __ATOMIC_ACQUIRE)) sched_yield();
        if (c->hdr->n_ranks != n_ranks) { g_err = "stub_rccl: ranks disagree on n_ranks"; munmap(p, c->map_bytes); free(c); return 1; }
    }
    if (hipHostMalloc((void **)&c->result, (size_t)n_ranks * kSlotBytes, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) { g_err = "stub_rccl: hipHostMalloc failed"; munmap(p, c->map_bytes); free(c); return 1; }
    __atomic_add_fetch(&c->hdr->attached, 1, __ATOMIC_ACQ_REL);
    barrier(c);                  /* like ncclCommInitRank: returns when every rank has joined */
    if (creator) shm_unlink(c->name);   /* everybody holds a mapping now: the name can go */
    *out = c;
    return 0;
}

int ncclCommDestroy(ncclComm_t c) {
    if (!c) return 0;
    if (c->ev_set) (void)hipEventSynchronize(c->ev);
    if (c->ev) (void)hipEventDestroy(c->ev);
    if (c->result) (void)hipHostFree(c->result);
    if (c->hdr) munmap((void *)c->hdr, c->map_bytes);
    free(c);
    return 0;
}

/* dtype: 2 = ncclInt32, 7 = ncclFloat32 (rccl.h); op 0 = ncclSum */
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t c, hipStream_t stream) {
    if (!c || op != 0 || (dtype != 2 && dtype != 7)) { g_err = "stub_rccl: unsupported all-reduce"; return 1; }
    const size_t per = kSlotBytes / 4;
    for (size_t at = 0; at < count; at += per) {
        const size_t n = count - at < per ? count - at : per;
        if (hipMemcpyAsync(c->slots + (size_t)c->rank * kSlotBytes, (const char *)send + at * 4, n * 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { g_err = "stub_rccl: device -> slot copy failed"; return 1; }
        barrier(c);
        if (c->ev_set && hipEventSynchronize(c->ev) != hipSuccess) { g_err = "stub_rccl: event wait failed"; return 1; }
        if (dtype == 2) {
            uint32_t *acc = (uint32_t *)c->result;
            memcpy(acc, c->slots, n * 4);
            for (int r = 1; r < c->n_ranks; ++r) {
                const uint32_t *x = (const uint32_t *)(c->slots + (size_t)r * kSlotBytes);
                for (size_t i = 0; i < n; ++i) acc[i] += x[i];   /* wrapping, like ncclInt32 sums */
            }
        } else {
            float *acc = (float *)c->result;
            memcpy(acc, c->slots, n * 4);
            for (int r = 1; r < c->n_ranks; ++r) {
                const float *x = (const float *)(c->slots + (size_t)r * kSlotBytes);
                for (size_t i = 0; i < n; ++i) acc[i] += x[i];   /* rank order: the same bits on every rank */
            }
        }
        barrier(c);              /* nobody overwrites a slot before everybody has read it */
        /* the result is ENQUEUED on the caller's stream, not waited for: like RCCL, the call orders nothing on any other stream */
        if (hipMemcpyAsync((char *)recv + at * 4, c->result, n * 4, hipMemcpyHostToDevice, stream) != hipSuccess || hipEventRecord(c->ev, stream) != hipSuccess) {
            g_err = "stub_rccl: result -> device copy failed";
            return 1;
        }
        c->ev_set = 1;
    }
    return 0;
}

/* every rank contributes `count` elements of 4 bytes; recv = [n_ranks][count]; send may alias recv + rank * count (in place) */
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm_t c, hipStream_t stream) {
    if (!c || (dtype != 2 && dtype != 7)) { g_err = "stub_rccl: unsupported all-gather"; return 1; }
    const size_t per = kSlotBytes / 4;
    for (size_t at = 0; at < count; at += per) {
        const size_t n = count - at < per ? count - at : per;
        if (hipMemcpyAsync(c->slots + (size_t)c->rank * kSlotBytes, (const char *)send + at * 4, n * 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { g_err = "stub_rccl: device -> slot copy failed"; return 1; }
        barrier(c);
        if (c->ev_set && hipEventSynchronize(c->ev) != hipSuccess) { g_err = "stub_rccl: event wait failed"; return 1; }
        for (int r = 0; r < c->n_ranks; ++r) {
            memcpy(c->result + (size_t)r * kSlotBytes, c->slots + (size_t)r * kSlotBytes, n * 4);
            if (hipMemcpyAsync((char *)recv + ((size_t)r * count + at) * 4, c->result + (size_t)r * kSlotBytes, n * 4, hipMemcpyHostToDevice, stream) != hipSuccess) {
                g_err = "stub_rccl: slot -> device copy failed";
                return 1;
            }
        }
        if (hipEventRecord(c->ev, stream) != hipSuccess) { g_err = "stub_rccl: event record failed"; return 1; }
        c->ev_set = 1;
        barrier(c);
    }
    return 0;
}
