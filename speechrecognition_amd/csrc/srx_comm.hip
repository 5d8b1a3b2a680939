// The exchange step of embedded training behind the C ABI: all-reduce of sufficient statistics across the ranks
// (one process per GPU) that hold shards of the corpus.
//
// The reference is single-process: what is exchanged are the sums its sequential `for i < ut.size()` loops build
// (src/hmm.cpp:275, 298, 319: zeta sums, zeta*x sums, zeta*(x-mean)^2 sums; :350-356 the state-model sums).
//
//   SRX_COMM_RCCL  one GPU per rank: RCCL all-reduce over xGMI on the context's stream (ncclAllReduce in place).
//                  librccl is opened on first use (dlopen), so libsrx.so loads where RCCL is absent and a process that
//                  already holds an RCCL (torch's) keeps using that one.
//   SRX_COMM_HOST  rehearsal transport for ranks that SHARE a GPU -- RCCL refuses two ranks on one device ("Duplicate
//                  GPU detected"), and the development boxes have one.  Buffers cross a POSIX shared-memory segment and
//                  are reduced in rank order on the host (deterministic, identical on every rank).  It exists so that
//                  the multi-rank control flow of the C++ layer can be tested on one GPU; it is not a data path.
//   world == 1     every collective is a no-op: the single-GPU run goes through the same calls.
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdlib>
#include <chrono>
#include <thread>

#include "srx_common.h"

namespace {
// the slice of RCCL's ABI that is used (rccl.h: ncclUniqueId is 128 opaque bytes passed by value)
struct nccl_uid { char internal[SRX_COMM_ID_BYTES]; };
typedef void *nccl_comm_t;
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_INT32 = 2, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8 };
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

int rccl_load(srx_ctx *ctx) {
    if (g_rccl.lib) return SRX_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "srx_comm: cannot open librccl (%s)", dlerror());
    RcclApi a;
    a.lib = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString)
        return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "srx_comm: librccl lacks an expected symbol");
    g_rccl = a;
    return SRX_OK;
}

// ---- host rehearsal transport ----
struct ShmHeader {
    std::atomic<int> ready;
    std::atomic<int> abort;   // a rank failed inside an all-reduce: every rank leaves the call with an error, no rank skips a barrier
    int world;
    size_t slot_bytes;
    pthread_barrier_t barrier;
};
constexpr size_t kSlotBytes = (size_t)16 << 20;   // reductions larger than a slot go in pieces
}  // namespace

struct srx_comm {
    srx_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    int transport = 0;   // 0 none (world 1), 1 RCCL, 2 host
    nccl_comm_t nccl = nullptr;
    // host transport
    std::string shm_name;
    ShmHeader *hdr = nullptr;
    unsigned char *slots = nullptr;
    size_t map_bytes = 0;
    std::vector<unsigned char> stage;
    int64_t bytes_reduced = 0;
};

extern "C" int srx_comm_unique_id(void *id_out) {
    if (!id_out) return srx_fail(nullptr, SRX_ERR_INVALID, "srx_comm_unique_id: null argument");
    int rc = rccl_load(nullptr);
    if (rc != SRX_OK) return rc;
    nccl_uid id;
    int e = g_rccl.GetUniqueId(&id);
    if (e != 0) return srx_fail(nullptr, SRX_ERR_DEVICE, "ncclGetUniqueId: %s", g_rccl.GetErrorString(e));
    std::memcpy(id_out, &id, SRX_COMM_ID_BYTES);
    return SRX_OK;
}

extern "C" int srx_comm_create(srx_ctx *ctx, int rank, int world, const void *id, srx_comm **out) {
    SRX_ENTER(ctx, "srx_comm_create");
    SRX_REQUIRE(ctx, out && world >= 1 && rank >= 0 && rank < world, "srx_comm_create: rank %d of world %d", rank, world);
    *out = nullptr;
    srx_comm *c = new srx_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    // SRX_COMM_FORCE_RCCL=1 with an id: a one-rank RCCL communicator instead of the no-op one, so that the RCCL calls
    // themselves (library lookup, ABI, stream ordering) can be exercised on a one-GPU box
    const bool force = id && std::getenv("SRX_COMM_FORCE_RCCL") && std::getenv("SRX_COMM_FORCE_RCCL")[0] == '1';
    if (world > 1 || force) {
        if (!id) {
            delete c;
            return srx_fail(ctx, SRX_ERR_INVALID, "srx_comm_create: world %d needs the id rank 0 got from srx_comm_unique_id", world);
        }
        int rc = rccl_load(ctx);
        if (rc != SRX_OK) {
            delete c;
            return rc;
        }
        nccl_uid uid;
        std::memcpy(&uid, id, SRX_COMM_ID_BYTES);
        int e = g_rccl.CommInitRank(&c->nccl, world, uid, rank);
        if (e != 0) {
            delete c;
            return srx_fail(ctx, SRX_ERR_DEVICE, "ncclCommInitRank(rank %d of %d, device %d): %s (RCCL needs one GPU per rank; ranks "
                            "sharing a device can only use srx_comm_create_host)", rank, world, ctx->device, g_rccl.GetErrorString(e));
        }
        c->transport = 1;
    }
    *out = c;
    return SRX_OK;
}

extern "C" int srx_comm_create_host(srx_ctx *ctx, int rank, int world, const char *shm_name, srx_comm **out) {
    SRX_ENTER(ctx, "srx_comm_create_host");
    SRX_REQUIRE(ctx, out && world >= 1 && rank >= 0 && rank < world && shm_name && shm_name[0] == '/',
                "srx_comm_create_host: rank %d of world %d, name must start with '/'", rank, world);
    *out = nullptr;
    srx_comm *c = new srx_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    if (world == 1) {
        *out = c;
        return SRX_OK;
    }
    c->transport = 2;
    c->shm_name = shm_name;
    c->map_bytes = sizeof(ShmHeader) + 256 + (size_t)world * kSlotBytes;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(shm_name);
        fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)c->map_bytes) != 0) {
            close(fd);
            fd = -1;
        }
    } else {
        for (int tries = 0; tries < 3000 && fd < 0; tries++) {   // rank 0 may not have created it yet: up to 30 s
            fd = shm_open(shm_name, O_RDWR, 0600);
            if (fd >= 0) {
                off_t sz = lseek(fd, 0, SEEK_END);
                if (sz < (off_t)c->map_bytes) {
                    close(fd);
                    fd = -1;
                }
            }
            if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    if (fd < 0) {
        delete c;
        return srx_fail(ctx, SRX_ERR_IO, "srx_comm_create_host: cannot open shared memory %s", shm_name);
    }
    void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        delete c;
        return srx_fail(ctx, SRX_ERR_NOMEM, "srx_comm_create_host: mmap failed");
    }
    c->hdr = reinterpret_cast<ShmHeader *>(m);
    c->slots = reinterpret_cast<unsigned char *>(m) + ((sizeof(ShmHeader) + 255) / 256) * 256;
    if (rank == 0) {
        pthread_barrierattr_t at;
        pthread_barrierattr_init(&at);
        pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->hdr->barrier, &at, (unsigned)world);
        pthread_barrierattr_destroy(&at);
        c->hdr->world = world;
        c->hdr->slot_bytes = kSlotBytes;
        c->hdr->abort.store(0, std::memory_order_relaxed);
        c->hdr->ready.store(1, std::memory_order_release);
    } else {
        for (int tries = 0; tries < 3000 && c->hdr->ready.load(std::memory_order_acquire) != 1; tries++)
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (c->hdr->ready.load(std::memory_order_acquire) != 1 || c->hdr->world != world) {
            munmap(m, c->map_bytes);
            delete c;
            return srx_fail(ctx, SRX_ERR_IO, "srx_comm_create_host: segment %s was not initialised for world %d", shm_name, world);
        }
    }
    c->stage.resize(kSlotBytes);
    pthread_barrier_wait(&c->hdr->barrier);   // everybody has mapped it: rank 0 may unlink the name
    if (rank == 0) shm_unlink(shm_name);
    *out = c;
    return SRX_OK;
}

extern "C" void srx_comm_destroy(srx_comm *c) {
    if (!c) return;
    if (c->transport == 1 && c->nccl) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        (void)g_rccl.CommDestroy(c->nccl);
    }
    if (c->transport == 2 && c->hdr) munmap(c->hdr, c->map_bytes);
    delete c;
}

extern "C" int srx_comm_rank(const srx_comm *c) { return c ? c->rank : -1; }
extern "C" int srx_comm_world(const srx_comm *c) { return c ? c->world : -1; }
extern "C" int64_t srx_comm_bytes_reduced(const srx_comm *c) { return c ? c->bytes_reduced : -1; }

namespace {
template <class T, bool MAX>
void reduce_slots(const unsigned char *slots, int world, size_t slot_bytes, size_t n, T *out) {
    const T *s0 = reinterpret_cast<const T *>(slots);
    for (size_t i = 0; i < n; i++) out[i] = s0[i];
    for (int r = 1; r < world; r++) {
        const T *s = reinterpret_cast<const T *>(slots + (size_t)r * slot_bytes);
        for (size_t i = 0; i < n; i++) {
            if (MAX) out[i] = s[i] > out[i] ? s[i] : out[i];   // NaN-free by construction (shift maxima)
            else out[i] = out[i] + s[i];
        }
    }
}

template <class T, bool MAX>
int allreduce(srx_comm *c, T *buf, int64_t n, int nccl_type, const char *what) {
    if (!c) return srx_fail(nullptr, SRX_ERR_INVALID, "%s: null communicator", what);
    srx_ctx *ctx = c->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    SRX_REQUIRE(ctx, n >= 0 && (buf || n == 0), "%s: bad argument", what);
    if (c->transport == 0 || n == 0) return SRX_OK;
    c->bytes_reduced += n * (int64_t)sizeof(T);
    if (c->transport == 1) {
        SrxProf prof_(ctx, MAX ? "rccl_allreduce_max" : "rccl_allreduce_sum");
        int e = g_rccl.AllReduce(buf, buf, (size_t)n, nccl_type, MAX ? NCCL_MAX : NCCL_SUM, c->nccl, ctx->stream);
        if (e != 0) return srx_fail(ctx, SRX_ERR_DEVICE, "%s: ncclAllReduce: %s", what, g_rccl.GetErrorString(e));
        return SRX_OK;
    }
    // Every rank passes the same barriers whatever happens to it (n is the same on every rank): a failed copy raises the
    // segment's abort flag instead of returning, so that no peer is left waiting in a barrier; the flag is seen by everyone
    // behind the next barrier and the call fails on every rank.
    const size_t per = kSlotBytes / sizeof(T);
    hipError_t first_err = hipSuccess;
    auto note = [&](hipError_t e) {
        if (e != hipSuccess) {
            if (first_err == hipSuccess) first_err = e;
            c->hdr->abort.store(1, std::memory_order_release);
        }
    };
    for (int64_t o = 0; o < n; o += (int64_t)per) {
        const size_t cnt = (size_t)((n - o) < (int64_t)per ? (n - o) : (int64_t)per);
        unsigned char *mine = c->slots + (size_t)c->rank * kSlotBytes;
        const bool live = c->hdr->abort.load(std::memory_order_acquire) == 0;
        if (live) {
            note(hipMemcpyAsync(mine, buf + o, cnt * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            note(hipStreamSynchronize(ctx->stream));
        }
        pthread_barrier_wait(&c->hdr->barrier);
        const bool ok = c->hdr->abort.load(std::memory_order_acquire) == 0;
        if (ok) reduce_slots<T, MAX>(c->slots, c->world, kSlotBytes, cnt, reinterpret_cast<T *>(c->stage.data()));
        pthread_barrier_wait(&c->hdr->barrier);   // everybody has read every slot before anyone overwrites its own
        if (ok) {
            note(hipMemcpyAsync(buf + o, c->stage.data(), cnt * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
            note(hipStreamSynchronize(ctx->stream));
        }
    }
    if (first_err != hipSuccess) return srx_fail(ctx, SRX_ERR_DEVICE, "%s: %s", what, hipGetErrorString(first_err));
    if (c->hdr->abort.load(std::memory_order_acquire) != 0)
        return srx_fail(ctx, SRX_ERR_DEVICE, "%s: another rank failed inside an all-reduce; the communicator is unusable", what);
    return SRX_OK;
}
}  // namespace

extern "C" int srx_allreduce_max_f32_dev(srx_comm *c, float *buf_dev, int64_t n) {
    return allreduce<float, true>(c, buf_dev, n, NCCL_FLOAT32, "srx_allreduce_max_f32_dev");
}
extern "C" int srx_allreduce_sum_f64_dev(srx_comm *c, double *buf_dev, int64_t n) {
    return allreduce<double, false>(c, buf_dev, n, NCCL_FLOAT64, "srx_allreduce_sum_f64_dev");
}
extern "C" int srx_allreduce_sum_i32_dev(srx_comm *c, int32_t *buf_dev, int64_t n) {
    return allreduce<int32_t, false>(c, buf_dev, n, NCCL_INT32, "srx_allreduce_sum_i32_dev");
}
