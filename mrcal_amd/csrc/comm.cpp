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
//
// A second transport, for the ranks of ONE host: the sum staged through a POSIX
// shared-memory segment (mrcal_amd_comm_create_host). It is synchronous and
// slow (two copies across PCIe and a process barrier per collective) and it is
// not what a multi-GPU solve should run on; it exists because RCCL refuses two
// ranks on one device, and with it the C++ sharded solve - its collectives, the
// outlier pass, the gather - runs at world > 1 on a one-GPU box, in the order
// and with the arithmetic (sum in rank order, the same on every rank) of the
// real thing. A rank that does not show up within the timeout is an error, not
// a hang.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <string>
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
    int         (*CommCount)(Comm, int*) = NULL;
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
    R.CommCount      = (int (*)(Comm, int*))                 dlsym(h, "ncclCommCount");        // (optional)
    if(!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllReduce)
    {
        set_error("the RCCL library lacks ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllReduce");
        return NULL;
    }
    R.handle = h;
    return &R;
}
const char* errstr(Rccl* R, int e) { return (R && R->GetErrorString) ? R->GetErrorString(e) : "?"; }

// ---- the host transport ------------------------------------------------------
// The segment: a header, then two banks of `world` slots of HOST_SLOT doubles. A collective writes the
// rank's slot of the current bank, meets the others at the barrier, sums the bank's slots in rank order and
// flips the bank: the next collective writes the other bank while a slow rank may still be reading this one,
// and the one after that is behind the next barrier
enum { HOST_SLOT = 1 << 15 };
struct HostHeader
{
    std::atomic<int> ready, arrived, generation;
    int world;
};
static_assert(sizeof(std::atomic<int>) == sizeof(int), "lock-free atomics in shared memory");
struct HostTransport
{
    std::string name;
    HostHeader* hdr    = NULL;
    double*     slots  = NULL;      // [2][world][HOST_SLOT]
    size_t      bytes  = 0;
    double*     staging = NULL;     // pinned [HOST_SLOT]
    int         bank   = 0;
    double      timeout_s = 120.0;
};
double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9*(double)ts.tv_nsec;
}
bool host_barrier(HostTransport* T, int world)
{
    HostHeader* H = T->hdr;
    const int gen = H->generation.load(std::memory_order_acquire);
    if(H->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == world)
    {
        H->arrived.store(0, std::memory_order_relaxed);
        H->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    const double t0 = now_s();
    for(long spin = 0; H->generation.load(std::memory_order_acquire) == gen; spin++)
    {
        if((spin & 63) == 63) sched_yield();
        if((spin & 4095) == 4095 && now_s() - t0 > T->timeout_s)
        {
            set_error("host communicator '%s': a rank did not reach the collective within %.0f s", T->name.c_str(), T->timeout_s);
            return false;
        }
    }
    return true;
}
void host_close(HostTransport* T)
{
    if(T == NULL) return;
    if(T->hdr)     munmap((void*)T->hdr, T->bytes);
    if(T->staging) (void)hipHostFree(T->staging);
    delete T;
}
} // namespace

struct mrcal_amd_comm
{
    Comm comm  = NULL;
    HostTransport* host = NULL;
    int  rank  = 0, world = 1;
    long Ncollectives = 0;
    long long Ndoubles = 0;      // summed over the collectives
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

// name: "/something", the same on every rank and not in use (rank 0 creates the segment, the others wait for it;
// it is unlinked as soon as everybody has it mapped)
mrcal_amd_comm_t* mrcal_amd_comm_create_host(const char* name, int rank, int world)
{
    last_error_string().clear();
    if(world < 1 || rank < 0 || rank >= world || name == NULL || name[0] != '/')
    { set_error("mrcal_amd_comm_create_host(): rank %d of %d, name '%s'", rank, world, name ? name : "(null)"); return NULL; }
    HostTransport* T = new HostTransport();
    T->name  = name;
    T->bytes = 4096 + (size_t)2*world*HOST_SLOT*sizeof(double);
    if(const char* env = getenv("MRCAL_AMD_HOST_COMM_TIMEOUT")) { const double v = atof(env); if(v > 0.) T->timeout_s = v; }
    int fd = -1;
    const double t0 = now_s();
    if(rank == 0)
    {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if(fd < 0 || ftruncate(fd, (off_t)T->bytes) != 0)
        { set_error("mrcal_amd_comm_create_host(): cannot create '%s'", name); if(fd >= 0) { close(fd); shm_unlink(name); } host_close(T); return NULL; }
    }
    else
        for(;;)
        {
            struct stat st;
            fd = shm_open(name, O_RDWR, 0600);
            if(fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == T->bytes) break;
            if(fd >= 0) close(fd);
            fd = -1;
            if(now_s() - t0 > T->timeout_s) { set_error("mrcal_amd_comm_create_host(): '%s' did not appear", name); host_close(T); return NULL; }
            usleep(1000);
        }
    void* m = mmap(NULL, T->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if(m == MAP_FAILED) { set_error("mrcal_amd_comm_create_host(): mmap('%s') failed", name); if(rank == 0) shm_unlink(name); host_close(T); return NULL; }
    T->hdr   = (HostHeader*)m;
    T->slots = (double*)((char*)m + 4096);
    if(rank == 0)
    {
        // (a fresh segment is zero: arrived = generation = 0)
        T->hdr->world = world;
        T->hdr->ready.store(1, std::memory_order_release);
    }
    else
        while(T->hdr->ready.load(std::memory_order_acquire) != 1)
        {
            if(now_s() - t0 > T->timeout_s) { set_error("mrcal_amd_comm_create_host(): '%s' was never initialized", name); host_close(T); return NULL; }
            usleep(1000);
        }
    if(T->hdr->world != world) { set_error("mrcal_amd_comm_create_host(): '%s' belongs to a world of %d, not %d", name, T->hdr->world, world); host_close(T); return NULL; }
    if(hipHostMalloc((void**)&T->staging, (size_t)HOST_SLOT*sizeof(double), hipHostMallocDefault) != hipSuccess)
    { set_error("mrcal_amd_comm_create_host(): hipHostMalloc failed"); if(rank == 0) shm_unlink(name); host_close(T); return NULL; }
    const bool together = host_barrier(T, world);
    if(rank == 0) shm_unlink(name);
    if(!together) { host_close(T); return NULL; }
    mrcal_amd_comm* c = new mrcal_amd_comm();
    c->rank = rank; c->world = world; c->host = T;
    return c;
}

void mrcal_amd_comm_destroy(mrcal_amd_comm_t* c)
{
    if(c == NULL) return;
    if(c->host) host_close(c->host);
    else
    {
        Rccl* R = rccl();
        if(R && c->comm) R->CommDestroy(c->comm);
    }
    delete c;
}

int  mrcal_amd_comm_rank (const mrcal_amd_comm_t* c) { return c ? c->rank  : 0; }
int  mrcal_amd_comm_world(const mrcal_amd_comm_t* c) { return c ? c->world : 1; }
long mrcal_amd_comm_Ncollectives(const mrcal_amd_comm_t* c) { return c ? c->Ncollectives : 0; }
long long mrcal_amd_comm_Ndoubles(const mrcal_amd_comm_t* c) { return c ? c->Ndoubles : 0; }
// how many ranks the TRANSPORT says the communicator has (ncclCommCount(); the host transport's segment): what a
// scaling run reports beside the world size it was told, -1 if the library cannot say
int mrcal_amd_comm_world_observed(const mrcal_amd_comm_t* c)
{
    if(c == NULL) return 1;
    if(c->host) return c->world;
    Rccl* R = rccl();
    int n = -1;
    if(R && R->CommCount && c->comm && R->CommCount(c->comm, &n) == 0) return n;
    return -1;
}

// in-place sum over the ranks of n doubles in device memory, queued on `stream`
bool mrcal_amd_comm_allreduce_sum(mrcal_amd_comm_t* c, double* buf, int64_t n, void* stream)
{
    if(c == NULL || n <= 0) return true;
    c->Ndoubles += n;
    if(c->host)
    {
        HostTransport* T = c->host;
        hipStream_t st = (hipStream_t)stream;
        for(int64_t i0 = 0; i0 < n; i0 += HOST_SLOT)
        {
            const int64_t m = (n - i0 < HOST_SLOT) ? n - i0 : (int64_t)HOST_SLOT;
            double* bank = T->slots + (size_t)T->bank*c->world*HOST_SLOT;
            if(hipMemcpyAsync(T->staging, buf + i0, (size_t)m*sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
               hipStreamSynchronize(st) != hipSuccess)
            { set_error("host all-reduce: the copy from the device failed"); return false; }
            memcpy(bank + (size_t)c->rank*HOST_SLOT, T->staging, (size_t)m*sizeof(double));
            if(!host_barrier(T, c->world)) return false;
            for(int64_t i = 0; i < m; i++)
            {
                double sum = bank[i];
                for(int r = 1; r < c->world; r++) sum += bank[(size_t)r*HOST_SLOT + i];
                T->staging[i] = sum;
            }
            T->bank ^= 1;
            if(hipMemcpyAsync(buf + i0, T->staging, (size_t)m*sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
               hipStreamSynchronize(st) != hipSuccess)
            { set_error("host all-reduce: the copy to the device failed"); return false; }
        }
        c->Ncollectives++;
        return true;
    }
    Rccl* R = rccl();
    if(!R) return false;
    const int e = R->AllReduce(buf, buf, (size_t)n, kDouble, kSum, c->comm, (hipStream_t)stream);
    if(e != 0) { set_error("ncclAllReduce: %s", errstr(R, e)); return false; }
    c->Ncollectives++;
    return true;
}

} // extern "C"
