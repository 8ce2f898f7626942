// stub_rccl.cpp -- a stand-in for librccl that lets TWO processes sharing ONE GPU drive the engine's own RCCL path
// (pe_dist_init_ex: two communicators, ncclCommAbort on a timeout, the collectives the engine issues between its kernels).
// Real RCCL refuses two ranks on one device, so until round 6 that path had only ever run with world size 1.
//
// TEST INFRASTRUCTURE: loaded through POSEVO_RCCL_PATH by tests/dist_worker.py (mode "stubrccl"); never part of the product.
//
// Semantics: every collective is SYNCHRONOUS on the calling host thread -- wait for the stream, copy out, meet the other ranks in
// a file-backed shared segment, copy the result back.  That is stricter than RCCL (whose calls return at once): two ranks
// that issue the collectives of two communicators in DIFFERENT host order pass with RCCL and dead-lock here -- which is the
// point: the engine promises the same order on every rank, and a rank that drains early must keep it.  A wait
// that exceeds STUB_RCCL_TIMEOUT_MS (default 20 000) returns an error instead of hanging.  Every call is appended to
// $STUB_RCCL_LOG.<rank> ("<communicator ordinal> <op> <count>"): the test compares the ranks' logs.
//
// Build: hipcc -shared -fPIC -O2 -o libstub_rccl.so stub_rccl.cpp   (tests/test_gpu_dist_custom.py does it)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct StubComm* ncclComm_t;
typedef int ncclResult_t;   // 0 = ncclSuccess
typedef int ncclDataType_t; // nccl.h: ncclUint32 = 3, ncclUint64 = 5
typedef int ncclRedOp_t;    // ncclSum = 0
}

namespace {
constexpr int MAX_RANKS = 8;
constexpr size_t SLOT_BYTES = 4u << 20;  // per rank and parity: the engine's exchanges are <= a few hundred KB
struct Segment {
    std::atomic<uint64_t> seq[MAX_RANKS];  // collectives rank r has PUBLISHED on this communicator
    std::atomic<uint32_t> joined;
    std::atomic<uint32_t> aborted;
    uint32_t pad[14];
    unsigned char buf[MAX_RANKS][2][SLOT_BYTES];
};
}  // namespace
struct StubComm {
    int rank, world, ordinal;
    Segment* seg;
    uint64_t ops;
    std::string path;
    std::vector<unsigned char> stage;
};
namespace {
std::atomic<int> g_comms{0};
long timeout_ms()
{
    const char* e = getenv("STUB_RCCL_TIMEOUT_MS");
    return e ? atol(e) : 20000;
}
void log_call(const StubComm* c, const char* op, size_t count)
{
    const char* base = getenv("STUB_RCCL_LOG");
    if (!base) return;
    const std::string path = std::string(base) + "." + std::to_string(c->rank);
    if (FILE* f = fopen(path.c_str(), "a")) {
        fprintf(f, "%d %s %zu\n", c->ordinal, op, count);
        fclose(f);
    }
}
// true when every rank has published `want` collectives; false on timeout / abort
bool wait_all(StubComm* c, uint64_t want)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t it = 0;; ++it) {
        bool all = true;
        for (int r = 0; r < c->world; ++r) all = all && c->seg->seq[r].load(std::memory_order_acquire) >= want;
        if (all) return true;
        if (c->seg->aborted.load(std::memory_order_acquire)) return false;
        if ((it & 1023) == 1023) {
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (ms > timeout_ms()) return false;
            std::this_thread::yield();
        }
    }
}
// one collective: publish `bytes` of this rank, meet the others, hand every rank's bytes to `combine`
template <typename F>
ncclResult_t exchange(StubComm* c, const void* dev_send, size_t bytes, hipStream_t s, F combine)
{
    if (bytes > SLOT_BYTES) return 5;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    const uint64_t k = c->ops++;
    unsigned char* mine = c->seg->buf[c->rank][k & 1];
    if (bytes && hipMemcpy(mine, dev_send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    c->seg->seq[c->rank].store(k + 1, std::memory_order_release);
    if (!wait_all(c, k + 1)) return 6;  // "remote error": a peer never arrived
    return combine(k & 1);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/dev/shm/posevo_stub_rccl_%d_%d_%ld", (int)getpid(), g_comms.load(),
             (long)std::chrono::steady_clock::now().time_since_epoch().count());
    g_comms.fetch_add(1000);  // unique ids of one process differ
    return 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank)
{
    if (!out || world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return 4;
    id.internal[sizeof(id.internal) - 1] = 0;
    const int fd = open(id.internal, O_RDWR | O_CREAT, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, sizeof(Segment)) != 0) { close(fd); return 2; }  // a new file reads as zeros: every counter starts at 0
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    StubComm* c = new StubComm();
    c->rank = rank;
    c->world = world;
    c->seg = static_cast<Segment*>(p);
    c->ops = 0;
    c->path = id.internal;
    static std::atomic<int> ordinal{0};
    c->ordinal = ordinal.fetch_add(1);
    c->seg->joined.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();  // ncclCommInitRank is collective
    while ((int)c->seg->joined.load() < world) {
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) {
            munmap(p, sizeof(Segment));
            delete c;
            return 6;
        }
        std::this_thread::yield();
    }
    log_call(c, "init", (size_t)world);
    *out = c;
    return 0;
}
static ncclResult_t comm_free(ncclComm_t c, const char* how)
{
    if (!c) return 4;
    log_call(c, how, 0);
    if (c->rank == 0) unlink(c->path.c_str());
    munmap(c->seg, sizeof(Segment));
    delete c;
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { return comm_free(c, "destroy"); }
ncclResult_t ncclCommAbort(ncclComm_t c)
{
    if (c) c->seg->aborted.store(1);
    return comm_free(c, "abort");
}
ncclResult_t ncclGroupStart() { return 0; }
ncclResult_t ncclGroupEnd() { return 0; }
const char* ncclGetErrorString(ncclResult_t r)
{
    return r == 0 ? "no error" : r == 6 ? "stub rccl: a peer did not arrive (collectives issued in different order?)" : "stub rccl error";
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t s)
{
    if (!c || type != 5 || op != 0) return 4;  // the engine reduces u64 sums only
    log_call(c, "allreduce", count);
    return exchange(c, send, count * 8, s, [&](int par) -> ncclResult_t {
        std::vector<uint64_t> sum(count, 0);
        for (int r = 0; r < c->world; ++r) {
            const uint64_t* v = reinterpret_cast<const uint64_t*>(c->seg->buf[r][par]);
            for (size_t i = 0; i < count; ++i) sum[i] += v[i];
        }
        return count && hipMemcpy(recv, sum.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
    });
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t s)
{
    if (!c || type != 3) return 4;  // the engine gathers u32 words only
    log_call(c, "allgather", count);
    const size_t bytes = count * 4;
    return exchange(c, send, bytes, s, [&](int par) -> ncclResult_t {
        for (int r = 0; r < c->world; ++r)
            if (bytes && hipMemcpy(static_cast<unsigned char*>(recv) + (size_t)r * bytes, c->seg->buf[r][par], bytes,
                                   hipMemcpyHostToDevice) != hipSuccess)
                return 1;
        return 0;
    });
}
}
