// Rccl.h -- the halo wire of a multi-GPU run: RCCL point-to-point over xGMI, issued from C++ inside the library.
//
// Reference analogue: the MPI calls inside Halo (components/omega/src/base/Halo.h:851-897 startReceives /
// startSends, Halo.cpp:607-700): one MPI_Irecv + MPI_Isend per neighbour, then host polling with MPI_Test
// behind a device-wide fence.  Here one ncclGroupStart / ncclRecv + ncclSend per neighbour / ncclGroupEnd on
// the HIP stream the pack and unpack kernels run on: the exchange is ordered by the stream, the host never
// waits, and a stepper can put it on its own communication stream next to interior compute.
// One process per GPU; the communicator is created from a unique id the launcher distributes (any side
// channel: torch.distributed/gloo in bench.py, MPI_Bcast in Omega's own driver).
#ifndef OMEGA_AMD_RCCL_H
#define OMEGA_AMD_RCCL_H

#include "Base.h"

namespace OMEGA {

class RcclComm {
 public:
   static constexpr int UniqueIdBytes = 128; ///< sizeof(ncclUniqueId)
   /// rank 0 of the job calls this and distributes the bytes (ncclGetUniqueId)
   static void getUniqueId(char Id[UniqueIdBytes]);
   /// collective over all NRanks processes; the calling process must already have selected its GPU
   /// (deviceInit).  ncclCommInitRank.
   RcclComm(const char Id[UniqueIdBytes], int NRanks, int Rank);
   ~RcclComm();
   RcclComm(const RcclComm &)            = delete;
   RcclComm &operator=(const RcclComm &) = delete;

   int NRanks = 0, Rank = -1; ///< as reported back by RCCL (ncclCommCount / ncclCommUserRank)
   int Version = 0;           ///< ncclGetVersion
   I8 NExchanges = 0;         ///< grouped exchanges issued so far

   /// One grouped exchange: for every i < N receive RecvBytes[i] from Peers[i] into RecvPtrs[i] and send
   /// SendBytes[i] from SendPtrs[i] to it, all on stream S (messages travel as bytes: R8, I4, ... payloads alike).
   /// Returns 0, or a non-zero code with the RCCL error string in lastError().
   int exchange(int N, const int *Peers, void *const *SendPtrs, const size_t *SendBytes, void *const *RecvPtrs,
                const size_t *RecvBytes, hipStream_t S);
   /// ncclAllGather of BytesPerRank bytes from every rank on stream S: Recv[NRanks][BytesPerRank] (device memory)
   int allGather(const void *Send, void *Recv, size_t BytesPerRank, hipStream_t S);
   const std::string &lastError() const { return LastError; }
   /// ncclCommAbort: after any error, or when the job decides not to use this communicator (e.g. another rank could
   /// not create its own).  Every later exchange returns an error.
   void abort();
   bool failed() const { return Failed; }

   /// HaloTransportFn-compatible thunk (Halo.h): Ctx is the RcclComm
   static int transport(void *Ctx, int NNghbr, const int *Tasks, void *const *SendPtrs, const size_t *SendBytes,
                        void *const *RecvPtrs, const size_t *RecvBytes, void *Stream);

 private:
   void *Comm  = nullptr; ///< ncclComm_t
   bool Failed = false;
   std::string LastError;
};

} // namespace OMEGA
#endif
