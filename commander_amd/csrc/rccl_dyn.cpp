#include "rccl_dyn.hpp"

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

#include "common_host.hpp"

#if !defined(CMDR_EMUL)
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == cmdr::kRcclIdBytes, "ncclUniqueId size");
static_assert((int)ncclFloat64 == 8 && (int)ncclSum == 0, "RCCL enum values");
#endif

namespace cmdr {
namespace {

struct Id128 { char b[kRcclIdBytes]; };   // passed by value like ncclUniqueId (a struct of 128 chars)

struct Api {
    void* h = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(Id128*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommSplit)(void*, int, int, void**, void*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    std::string err;
};

Api& api() {
    static Api A;
    static std::once_flag once;
    std::call_once(once, [] {
        // a process that already holds an RCCL (torch's bundled copy) gets that one: same SONAME
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // CMDR_RCCL_LIB names THE library to bind (no fall-back to the default names: a wrong path is an error)
        const char* forced = std::getenv("CMDR_RCCL_LIB");
        if (forced) A.h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        else
            for (const char* n : names) {
                if (A.h) break;
                A.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            }
        if (!A.h) {
            const char* e = dlerror();           // ONE call: dlerror() clears the message it returns
            A.err = std::string("cannot load librccl: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(A.h, n);
            if (!p && A.err.empty()) A.err = std::string("librccl lacks ") + n;
            return p;
        };
        A.GetVersion = (decltype(A.GetVersion))sym("ncclGetVersion");
        A.GetUniqueId = (decltype(A.GetUniqueId))sym("ncclGetUniqueId");
        A.CommInitRank = (decltype(A.CommInitRank))sym("ncclCommInitRank");
        A.CommSplit = (decltype(A.CommSplit))sym("ncclCommSplit");
        A.CommDestroy = (decltype(A.CommDestroy))sym("ncclCommDestroy");
        A.CommCount = (decltype(A.CommCount))sym("ncclCommCount");
        A.CommUserRank = (decltype(A.CommUserRank))sym("ncclCommUserRank");
        A.AllReduce = (decltype(A.AllReduce))sym("ncclAllReduce");
        A.GetErrorString = (decltype(A.GetErrorString))sym("ncclGetErrorString");
        A.ReduceScatter = (decltype(A.ReduceScatter))sym("ncclReduceScatter");
        A.AllGather = (decltype(A.AllGather))sym("ncclAllGather");
        A.GroupStart = (decltype(A.GroupStart))sym("ncclGroupStart");
        A.GroupEnd = (decltype(A.GroupEnd))sym("ncclGroupEnd");
    });
    if (!A.err.empty()) throw Error("RCCL unavailable: " + A.err);
    return A;
}

void check(int rc, const char* what) {
    if (rc == 0) return;
    Api& A = api();
    throw Error(std::string(what) + " failed: " + (A.GetErrorString ? A.GetErrorString(rc) : "?") + " (" +
                std::to_string(rc) + ")");
}

}  // namespace

void RcclComm::group_start() { check(api().GroupStart(), "ncclGroupStart"); }
void RcclComm::group_end() { check(api().GroupEnd(), "ncclGroupEnd"); }

int RcclComm::version() {
    int v = 0;
    check(api().GetVersion(&v), "ncclGetVersion");
    return v;
}

void RcclComm::unique_id(char out[kRcclIdBytes]) {
    Id128 id;
    std::memset(&id, 0, sizeof(id));
    check(api().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, id.b, kRcclIdBytes);
}

void RcclComm::init(const char idb[kRcclIdBytes], int rank, int nranks) {
    CMDR_REQUIRE(!comm_, "RCCL communicator already initialised");
    CMDR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    Id128 id;
    std::memcpy(id.b, idb, kRcclIdBytes);
    check(api().CommInitRank(&comm_, nranks, id, rank), "ncclCommInitRank");
}

void RcclComm::split_from(const RcclComm& parent, int color, int key) {
    CMDR_REQUIRE(!comm_ && parent.comm_, "bad communicator state for a split");
    check(api().CommSplit(parent.comm_, color, key, &comm_, nullptr), "ncclCommSplit");
}

void RcclComm::destroy() {
    if (comm_) (void)api().CommDestroy(comm_);
    comm_ = nullptr;
}

RcclComm::~RcclComm() { destroy(); }

int RcclComm::size() const {
    int n = 0;
    CMDR_REQUIRE(comm_, "no communicator");
    check(api().CommCount(comm_, &n), "ncclCommCount");
    return n;
}

int RcclComm::rank() const {
    int r = 0;
    CMDR_REQUIRE(comm_, "no communicator");
    check(api().CommUserRank(comm_, &r), "ncclCommUserRank");
    return r;
}

void RcclComm::allreduce_sum(double* dev, int64_t n, void* hip_stream) const {
    CMDR_REQUIRE(comm_, "no communicator");
    check(api().AllReduce(dev, dev, (size_t)n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm_, hip_stream), "ncclAllReduce");
}

void RcclComm::reduce_scatter_sum(double* dev, int64_t count, void* hip_stream) const {
    CMDR_REQUIRE(comm_, "no communicator");
    const int r = rank();     // in-place form: recvbuff = sendbuff + rank * recvcount
    check(api().ReduceScatter(dev, dev + (int64_t)r * count, (size_t)count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm_, hip_stream),
          "ncclReduceScatter");
}

void RcclComm::all_gather(double* dev, int64_t count, void* hip_stream) const {
    CMDR_REQUIRE(comm_, "no communicator");
    const int r = rank();     // in-place form: sendbuff = recvbuff + rank * sendcount
    check(api().AllGather(dev + (int64_t)r * count, dev, (size_t)count, /*ncclFloat64*/ 8, comm_, hip_stream), "ncclAllGather");
}

}  // namespace cmdr
