// az_rccl.hip -- the proposal exchange of an image-sharded run as ONE ncclAllGather on the context's stream
// (every rank contributes the fixed-size result records of its images; rank order = image order).
// RCCL is bound at run time (dlopen / dlsym): the process that already holds an RCCL (PyTorch-ROCm ships its own
// librccl.so) must not get a second one linked in, and a process that never exchanges anything needs none.
#include "az_dev.h"

#include <dlfcn.h>
#include <cstring>
#include <string>

namespace {

struct NcclUid { char internal[128]; };                      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_uid)(NcclUid *);
typedef int (*fn_init_rank)(void **, int, NcclUid, int);     // ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int)
typedef int (*fn_all_gather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);

struct Rccl {
    void *lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    // the RCCL this process already has (by soname or by the name PyTorch loads it under), else ROCm's
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) { r.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return r; }
    r.get_uid = (fn_get_uid)dlsym(r.lib, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(r.lib, "ncclCommInitRank");
    r.all_gather = (fn_all_gather)dlsym(r.lib, "ncclAllGather");
    r.destroy = (fn_destroy)dlsym(r.lib, "ncclCommDestroy");
    r.errstr = (fn_errstr)dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_uid || !r.init_rank || !r.all_gather || !r.destroy) { r.why = "librccl.so lacks an expected symbol"; r.lib = nullptr; }
    return r;
}

std::string nccl_err(int rc)
{
    Rccl &r = rccl();
    return std::string("RCCL error ") + std::to_string(rc) + (r.errstr ? std::string(": ") + r.errstr(rc) : std::string());
}

}  // namespace

// 0 ok; otherwise *why says what failed
int azk_rccl_unique_id(void *id128, std::string *why)
{
    Rccl &r = rccl();
    if (!r.lib) { *why = r.why; return -1; }
    NcclUid u;
    const int rc = r.get_uid(&u);
    if (rc) { *why = nccl_err(rc); return -1; }
    std::memcpy(id128, u.internal, 128);
    return 0;
}

int azk_rccl_init(const void *id128, int nranks, int rank, void **comm_out, std::string *why)
{
    Rccl &r = rccl();
    if (!r.lib) { *why = r.why; return -1; }
    NcclUid u;
    std::memcpy(u.internal, id128, 128);
    void *comm = nullptr;
    const int rc = r.init_rank(&comm, nranks, u, rank);
    if (rc) { *why = nccl_err(rc); return -1; }
    *comm_out = comm;
    return 0;
}

int azk_rccl_all_gather(void *comm, hipStream_t s, const void *send, void *recv, size_t bytes, std::string *why)
{
    Rccl &r = rccl();
    if (!r.lib) { *why = r.why; return -1; }
    const int rc = r.all_gather(send, recv, bytes, /* ncclUint8 */ 1, comm, s);
    if (rc) { *why = nccl_err(rc); return -1; }
    return 0;
}

void azk_rccl_destroy(void *comm)
{
    Rccl &r = rccl();
    if (r.lib && comm) r.destroy(comm);
}
