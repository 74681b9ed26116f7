// extern "C" weight broadcast over RCCL (include/m0_engine.h, SURVEY 8b: "RCCL over xGMI only for the optional weight
// broadcast") for callers without torch: one process per GPU, rank 0 makes a 128-byte id and ships it to the others by any
// out-of-band means, every rank creates its communicator, and m0_net_broadcast_weights sends every packed device buffer of a
// FINALIZED network from the root to all ranks (the ranks finalize a network of the same configuration first -- with any
// weights of the right shapes -- so the buffer list and sizes agree; a mismatch is detected before anything is overwritten).
// librccl is loaded on first use (dlopen), so a process that never broadcasts does not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/m0_engine.h"
#include "net.h"
#include "capi_common.h"

namespace {
// the part of rccl.h this file needs (stable NCCL 2 API)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0, ncclUint64 = 5 };      // ncclDataType_t
enum { ncclMax = 2, ncclMin = 3 };           // ncclRedOp_t
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
        }
        if (!r.h) { r.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.h, n); if (!p) r.why = std::string("librccl has no ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r.why.empty() ? &r : nullptr;
}
std::string rccl_err(const char* what, int rc) {
    Rccl* r = rccl();
    return std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(rc) : "rccl error");
}
}  // namespace

struct m0_dist {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    uint64_t* scratch = nullptr;        // [4] device words for the agreement check
};

extern "C" {

int m0_dist_unique_id(void* id128) {
    if (!id128) { m0_set_error("id128 is null"); return M0_ERR_INVALID; }
    Rccl* r = rccl();
    if (!r) { m0_set_error("RCCL unavailable"); return M0_ERR_UNSUPPORTED; }
    ncclUniqueId id;
    const int rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) { m0_set_error(rccl_err("ncclGetUniqueId", rc)); return M0_ERR_HIP; }
    memcpy(id128, id.internal, 128);
    return M0_OK;
}

m0_dist* m0_dist_create(int rank, int world, const void* id128, int hip_device) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) { m0_set_error("invalid argument"); return nullptr; }
    Rccl* r = rccl();
    if (!r) { m0_set_error("RCCL unavailable"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { m0_set_error("no HIP device available"); return nullptr; }
    if (hip_device < 0 || hip_device >= ndev || hipSetDevice(hip_device) != hipSuccess) { m0_set_error("hip_device out of range"); return nullptr; }
    m0_dist* d = new m0_dist();
    d->rank = rank; d->world = world; d->device = hip_device;
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    const int rc = r->CommInitRank(&d->comm, world, id, rank);
    if (rc != ncclSuccess) { m0_set_error(rccl_err("ncclCommInitRank", rc)); delete d; return nullptr; }
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d->scratch), 4 * sizeof(uint64_t)) != hipSuccess) {
        m0_set_error("hipStreamCreate / hipMalloc failed");
        if (d->stream) (void)hipStreamDestroy(d->stream);
        (void)r->CommDestroy(d->comm);
        delete d;
        return nullptr;
    }
    return d;
}

void m0_dist_destroy(m0_dist* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->stream) { (void)hipStreamSynchronize(d->stream); (void)hipStreamDestroy(d->stream); }
    if (d->scratch) (void)hipFree(d->scratch);
    if (Rccl* r = rccl()) if (d->comm) (void)r->CommDestroy(d->comm);
    delete d;
}

int m0_dist_rank(const m0_dist* d) { return d ? d->rank : -1; }
int m0_dist_world(const m0_dist* d) { return d ? d->world : 0; }

int m0_net_broadcast_weights(m0_net* n, m0_dist* d, int root) {
    if (!n || !d || root < 0 || root >= d->world) { m0_set_error("invalid argument"); return M0_ERR_INVALID; }
    Net* net = m0_net_impl(n);
    if (!net || !net->ready()) { m0_set_error("broadcast needs a finalized network on every rank (same configuration)"); return M0_ERR_STATE; }
    if (m0_net_device(n) != d->device) { m0_set_error("network and communicator are on different devices"); return M0_ERR_INVALID; }
    Rccl* r = rccl();
    if (!r) { m0_set_error("RCCL unavailable"); return M0_ERR_UNSUPPORTED; }
    if (hipSetDevice(d->device) != hipSuccess) { m0_set_error("hipSetDevice failed"); return M0_ERR_HIP; }
    m0_net_lock(n);                                     // no forward of this network while its weights change
    (void)hipStreamSynchronize(m0_net_stream(n));
    const std::vector<void*>& bufs = net->device_buffers();
    const std::vector<size_t>& bytes = net->device_buffer_bytes();
    int rc = M0_OK;
    do {
        if (bufs.size() != bytes.size()) { m0_set_error("internal: buffer list"); rc = M0_ERR_STATE; break; }
        // every rank must hold the same list: (count, total bytes, size checksum) agree <=> their max and min over the ranks agree
        uint64_t sig[4] = {bufs.size(), 0, 0x9E3779B97F4A7C15ull, 0};
        for (size_t i = 0; i < bytes.size(); ++i) { sig[1] += bytes[i]; sig[2] = (sig[2] ^ bytes[i]) * 0x100000001B3ull + i; }
        uint64_t mx[4], mn[4];
        (void)hipMemcpyAsync(d->scratch, sig, sizeof(sig), hipMemcpyHostToDevice, d->stream);
        int e = r->AllReduce(d->scratch, d->scratch, 4, ncclUint64, ncclMax, d->comm, d->stream);
        if (e == ncclSuccess) (void)hipMemcpyAsync(mx, d->scratch, sizeof(mx), hipMemcpyDeviceToHost, d->stream);
        if (e == ncclSuccess) (void)hipMemcpyAsync(d->scratch, sig, sizeof(sig), hipMemcpyHostToDevice, d->stream);
        if (e == ncclSuccess) e = r->AllReduce(d->scratch, d->scratch, 4, ncclUint64, ncclMin, d->comm, d->stream);
        if (e == ncclSuccess) (void)hipMemcpyAsync(mn, d->scratch, sizeof(mn), hipMemcpyDeviceToHost, d->stream);
        if (e != ncclSuccess) { m0_set_error(rccl_err("ncclAllReduce", e)); rc = M0_ERR_HIP; break; }
        if (hipStreamSynchronize(d->stream) != hipSuccess) { m0_set_error("broadcast: stream failed"); rc = M0_ERR_HIP; break; }
        if (memcmp(mx, mn, sizeof(mx)) != 0) {
            m0_set_error("broadcast: the ranks hold networks of different configurations (buffer lists differ)");
            rc = M0_ERR_INVALID;
            break;
        }
        for (size_t i = 0; i < bufs.size() && rc == M0_OK; ++i) {
            e = r->Broadcast(bufs[i], bufs[i], bytes[i], ncclInt8, root, d->comm, d->stream);
            if (e != ncclSuccess) { m0_set_error(rccl_err("ncclBroadcast", e)); rc = M0_ERR_HIP; }
        }
        if (rc == M0_OK && hipStreamSynchronize(d->stream) != hipSuccess) { m0_set_error("broadcast: stream failed"); rc = M0_ERR_HIP; }
    } while (0);
    m0_net_unlock(n);
    return rc;
}

}  // extern "C"
