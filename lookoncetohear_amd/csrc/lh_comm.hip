// The path's one exchange step for hosts without Python (SURVEY.md §8b "later": rccl_allreduce_f64; §8e): an
// all-reduce (sum) of the fp64 metric sums over one process per GPU — RCCL over xGMI.  The reference does this with
// Lightning's `sync_dist` all-reduce (src/ts_hear_embed_pl_module.py:82-107); the Python host of this repo uses
// torch.distributed (backend "nccl" = RCCL) and never calls these.
//
// RCCL is bound at run time with dlopen/dlsym, not at link time: a process that already carries an RCCL (PyTorch-ROCm
// bundles its own librccl.so) must not get a second copy, and a single-GPU user of the library needs no RCCL at all.
// LOOKONCE_RCCL_LIB overrides the library name.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "lh_common.h"

namespace {

struct UniqueId { char internal[128]; };            // ncclUniqueId: 128 opaque bytes (rccl.h NCCL_UNIQUE_ID_BYTES)
using comm_t = void*;                                // ncclComm_t
enum { kNcclFloat64 = 8, kNcclSum = 0 };             // rccl.h: ncclDataType_t ncclFloat64 = 8, ncclRedOp_t ncclSum = 0

struct Rccl {
    void* so = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    bool ok = false;
};

void bind(Rccl& r) {
    const char* names[] = {getenv("LOOKONCE_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // First pass over ALL names with RTLD_NOLOAD: reuse the copy the process already mapped (PyTorch-ROCm bundles its own
    // librccl.so; /opt/rocm ships librccl.so.1 — trying to LOAD the first name before LOOKING for the second would map a
    // second RCCL next to torch's).  Only when none is mapped does the second pass really load one.
    for (int pass = 0; pass < 2 && !r.so; ++pass)
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.so = dlopen(n, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_LOCAL));
            if (r.so) break;
        }
    if (!r.so) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.so, "ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
}

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { bind(r); });          // thread-safe: several host threads may create communicators
    return r;
}

}  // namespace

extern "C" int lh_comm_unique_id(void* id128) {
    if (!id128) return LH_ERR_ARG;
    Rccl& r = rccl();
    if (!r.ok) return LH_ERR_UNSUPPORTED;
    UniqueId id;
    if (r.GetUniqueId(&id) != 0) return LH_ERR_LAUNCH;
    memcpy(id128, id.internal, sizeof(id.internal));
    return LH_OK;
}

extern "C" int lh_comm_init(const void* id128, int n_ranks, int rank, void** comm) {
    if (!id128 || !comm || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return LH_ERR_ARG;
    Rccl& r = rccl();
    if (!r.ok) return LH_ERR_UNSUPPORTED;
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    comm_t c = nullptr;
    if (r.CommInitRank(&c, n_ranks, id, rank) != 0) return LH_ERR_LAUNCH;
    *comm = c;
    return LH_OK;
}

extern "C" int lh_allreduce_f64(void* comm, double* buf, int count, lh_stream_t stream) {
    if (!comm || !buf || count <= 0) return LH_ERR_ARG;
    Rccl& r = rccl();
    if (!r.ok) return LH_ERR_UNSUPPORTED;
    return r.AllReduce(buf, buf, (size_t)count, kNcclFloat64, kNcclSum, comm, (hipStream_t)stream) == 0 ? LH_OK
                                                                                                      : LH_ERR_LAUNCH;
}

extern "C" int lh_comm_destroy(void* comm) {
    if (!comm) return LH_ERR_ARG;
    Rccl& r = rccl();
    if (!r.ok) return LH_ERR_UNSUPPORTED;
    return r.CommDestroy(comm) == 0 ? LH_OK : LH_ERR_LAUNCH;
}
