// RCCL communicator of the multi-GPU solve: one process per GPU, the collectives
// of the sharded dog-leg step (solver.cpp) issued on the problem's own HIP
// stream from C++. Python only hands over the 128-byte unique id (rank 0 makes
// it; any side channel - torch.distributed over gloo in mrcal_amd/parallel.py -
// carries it to the other ranks).
//
// RCCL is not linked: it is opened at run time. A process that has PyTorch
// loaded already has a copy (torch/lib/librccl.so); that copy is used if it
// is there, so that ONE RCCL lives in the process. Otherwise ROCm's
// (librccl.so.1). MRCAL_AMD_RCCL=<path> overrides.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "host_state.hpp"
#include "../../include/mrcal_amd.h"

using namespace mrcal_amd;

namespace {
struct UniqueId { char internal[128]; };      // ncclUniqueId (rccl.h:40-43)
typedef void* Comm;
enum { kSum = 0, kDouble = 8 };               // ncclSum, ncclFloat64 (rccl.h:448,467)
struct Rccl
{
    void* handle = NULL;
    int         (*GetUniqueId)(UniqueId*) = NULL;
    int         (*CommInitRank)(Comm*, int, UniqueId, int) = NULL;
    int         (*CommDestroy)(Comm) = NULL;
    int         (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = NULL;
    const char* (*GetErrorString)(int) = NULL;
};
Rccl* rccl()
{
    static Rccl R;
    static bool tried = false;
    if(tried) return R.handle ? &R : NULL;
    tried = true;
    const char* env = getenv("MRCAL_AMD_RCCL");
    void* h = NULL;
    if(env && *env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    static const char* names[] = { "librccl.so", "librccl.so.1" };
    for(int i = 0; i < 2 && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);      // a copy this process already has
    for(int i = 1; i >= 0 && !h; i--) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if(!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if(!h) { set_error("could not open RCCL (librccl.so): %s", dlerror()); return NULL; }
    R.GetUniqueId    = (int (*)(UniqueId*))                  dlsym(h, "ncclGetUniqueId");
    R.CommInitRank   = (int (*)(Comm*, int, UniqueId, int))  dlsym(h, "ncclCommInitRank");
    R.CommDestroy    = (int (*)(Comm))                       dlsym(h, "ncclCommDestroy");
    R.AllReduce      = (int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t)) dlsym(h, "ncclAllReduce");
    R.GetErrorString = (const char* (*)(int))                dlsym(h, "ncclGetErrorString");
    if(!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllReduce)
    {
        set_error("the RCCL library lacks ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllReduce");
        return NULL;
    }
    R.handle = h;
    return &R;
}
const char* errstr(Rccl* R, int e) { return (R && R->GetErrorString) ? R->GetErrorString(e) : "?"; }
} // namespace

struct mrcal_amd_comm
{
    Comm comm  = NULL;
    int  rank  = 0, world = 1;
    long Ncollectives = 0;
};

extern "C" {

bool mrcal_amd_comm_unique_id(void* id128)
{
    last_error_string().clear();
    Rccl* R = rccl();
    if(!R) return false;
    UniqueId id;
    const int e = R->GetUniqueId(&id);
    if(e != 0) { set_error("ncclGetUniqueId: %s", errstr(R, e)); return false; }
    memcpy(id128, &id, sizeof(id));
    return true;
}

mrcal_amd_comm_t* mrcal_amd_comm_create(const void* id128, int rank, int world)
{
    last_error_string().clear();
    Rccl* R = rccl();
    if(!R) return NULL;
    if(world < 1 || rank < 0 || rank >= world) { set_error("mrcal_amd_comm_create(): rank %d of %d", rank, world); return NULL; }
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    mrcal_amd_comm* c = new mrcal_amd_comm();
    c->rank = rank; c->world = world;
    const int e = R->CommInitRank(&c->comm, world, id, rank);
    if(e != 0) { set_error("ncclCommInitRank: %s", errstr(R, e)); delete c; return NULL; }
    return c;
}

void mrcal_amd_comm_destroy(mrcal_amd_comm_t* c)
{
    if(c == NULL) return;
    Rccl* R = rccl();
    if(R && c->comm) R->CommDestroy(c->comm);
    delete c;
}

int  mrcal_amd_comm_rank (const mrcal_amd_comm_t* c) { return c ? c->rank  : 0; }
int  mrcal_amd_comm_world(const mrcal_amd_comm_t* c) { return c ? c->world : 1; }
long mrcal_amd_comm_Ncollectives(const mrcal_amd_comm_t* c) { return c ? c->Ncollectives : 0; }

// in-place sum over the ranks of n doubles in device memory, queued on `stream`
bool mrcal_amd_comm_allreduce_sum(mrcal_amd_comm_t* c, double* buf, int64_t n, void* stream)
{
    if(c == NULL || n <= 0) return true;
    Rccl* R = rccl();
    if(!R) return false;
    const int e = R->AllReduce(buf, buf, (size_t)n, kDouble, kSum, c->comm, (hipStream_t)stream);
    if(e != 0) { set_error("ncclAllReduce: %s", errstr(R, e)); return false; }
    c->Ncollectives++;
    return true;
}

} // extern "C"
