// RCCL (the ROCm collective library; backend "nccl" of torch.distributed IS this library) bound at run time with
// dlopen, so libcmdr_hip.so has no link-time dependency on it and single-GPU users never load it.  Only what the CR
// path needs: communicator bootstrap from a 128-byte unique id that the host language broadcasts (MPI_Bcast in the
// Fortran driver, torch.distributed in bench.py), a sub-communicator split, and the stream-ordered all-reduce of fp64
// vectors that replaces libsharp2's MPI exchange + mpi_dot_product (commander3/src/comm_utils.f90:599-614).
#pragma once
#include <cstdint>

namespace cmdr {

constexpr int kRcclIdBytes = 128;   // sizeof(ncclUniqueId)

class RcclComm {
  public:
    RcclComm() = default;
    ~RcclComm();
    RcclComm(const RcclComm&) = delete;
    RcclComm& operator=(const RcclComm&) = delete;
    static void unique_id(char out[kRcclIdBytes]);                       // ncclGetUniqueId
    void init(const char id[kRcclIdBytes], int rank, int nranks);        // ncclCommInitRank on the current device
    void split_from(const RcclComm& parent, int color, int key);         // ncclCommSplit (collective over parent)
    void destroy();                                                      // ncclCommDestroy; ready() is false afterwards
    bool ready() const { return comm_ != nullptr; }
    int size() const;                                                    // ncclCommCount, read back from the library
    int rank() const;                                                    // ncclCommUserRank
    void allreduce_sum(double* dev, int64_t n, void* hip_stream) const;  // in place, enqueued on hip_stream
    // in place on a buffer of size() * count doubles: rank r ends up with the sum of everybody's chunk r at offset r * count
    void reduce_scatter_sum(double* dev, int64_t count, void* hip_stream) const;
    // in place: everybody's chunk (at offset rank * count) is distributed to all
    void all_gather(double* dev, int64_t count, void* hip_stream) const;
    static int version();                                                // ncclGetVersion
    static void group_start();                                           // ncclGroupStart / ncclGroupEnd: several
    static void group_end();                                             // all-reduces of one stream as one operation
  private:
    void* comm_ = nullptr;
};

}  // namespace cmdr
