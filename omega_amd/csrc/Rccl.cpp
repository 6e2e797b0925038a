// Rccl.cpp -- see Rccl.h.
#include "Rccl.h"

#include <rccl/rccl.h>

#include <cstring>

namespace OMEGA {

static_assert(sizeof(ncclUniqueId) == RcclComm::UniqueIdBytes, "ncclUniqueId size changed");

#define RCCL_CHECK(call)                                                                                           \
   do {                                                                                                            \
      ncclResult_t r_ = (call);                                                                                    \
      if (r_ != ncclSuccess)                                                                                       \
         ::OMEGA::abortError(__FILE__, __LINE__, std::string(#call) + ": " + ncclGetErrorString(r_));              \
   } while (0)

void RcclComm::getUniqueId(char Id[UniqueIdBytes]) {
   ncclUniqueId U;
   RCCL_CHECK(ncclGetUniqueId(&U));
   std::memcpy(Id, &U, sizeof(U));
}

RcclComm::RcclComm(const char Id[UniqueIdBytes], int NRanks_, int Rank_) {
   OMEGA_REQUIRE(NRanks_ >= 1 && Rank_ >= 0 && Rank_ < NRanks_, "RcclComm: bad rank / size");
   ncclUniqueId U;
   std::memcpy(&U, Id, sizeof(U));
   ncclComm_t C = nullptr;
   RCCL_CHECK(ncclCommInitRank(&C, NRanks_, U, Rank_));
   Comm = C;
   try { // the destructor does not run when the constructor throws: give the communicator back here
      RCCL_CHECK(ncclCommCount(C, &NRanks));
      RCCL_CHECK(ncclCommUserRank(C, &Rank));
      RCCL_CHECK(ncclGetVersion(&Version));
      OMEGA_REQUIRE(NRanks == NRanks_ && Rank == Rank_, "RcclComm: communicator reports a different rank / size");
   } catch (...) {
      abort();
      throw;
   }
}

RcclComm::~RcclComm() {
   if (Comm)
      (void)ncclCommDestroy(static_cast<ncclComm_t>(Comm));
}

// An error on a communicator is fatal for the job (no retry anywhere in this library or in bench.py): the
// communicator is aborted -- queued operations are dropped, peers see the failure instead of waiting for messages
// that never come -- and every later exchange is refused.
void RcclComm::abort() {
   if (Comm)
      (void)ncclCommAbort(static_cast<ncclComm_t>(Comm));
   Comm   = nullptr;
   Failed = true;
}

int RcclComm::exchange(int N, const int *Peers, void *const *SendPtrs, const size_t *SendBytes, void *const *RecvPtrs,
                       const size_t *RecvBytes, hipStream_t S) {
   if (Failed || !Comm) {
      LastError = "RcclComm::exchange: the communicator was aborted after an earlier error (" + LastError + ")";
      return 1;
   }
   ncclComm_t C = static_cast<ncclComm_t>(Comm);
   auto Fail    = [&](const char *What, ncclResult_t R) {
      LastError = std::string(What) + ": " + ncclGetErrorString(R);
      abort(); // a partly enqueued group must not leave peers waiting for the rest
      return 1;
   };
   for (int I = 0; I < N; ++I)
      if (Peers[I] < 0 || Peers[I] >= NRanks) {
         LastError = "RcclComm::exchange: bad peer or message size";
         return 1;
      }
   ncclResult_t R = ncclGroupStart();
   if (R != ncclSuccess)
      return Fail("ncclGroupStart", R);
   // receives first, then sends (the reference posts its MPI_Irecv's before the MPI_Isend's too, Halo.h:851-897);
   // inside a group the order only matters for matching several messages between the same pair of ranks
   for (int I = 0; I < N && R == ncclSuccess; ++I)
      if (RecvBytes[I])
         R = ncclRecv(RecvPtrs[I], RecvBytes[I], ncclInt8, Peers[I], C, S);
   for (int I = 0; I < N && R == ncclSuccess; ++I)
      if (SendBytes[I])
         R = ncclSend(SendPtrs[I], SendBytes[I], ncclInt8, Peers[I], C, S);
   const ncclResult_t RE = ncclGroupEnd(); // always close the group
   if (R != ncclSuccess)
      return Fail("ncclSend/ncclRecv", R);
   if (RE != ncclSuccess)
      return Fail("ncclGroupEnd", RE);
   ++NExchanges;
   return 0;
}

int RcclComm::allGather(const void *Send, void *Recv, size_t BytesPerRank, hipStream_t S) {
   if (Failed || !Comm) {
      LastError = "RcclComm::allGather: the communicator was aborted after an earlier error (" + LastError + ")";
      return 1;
   }
   const ncclResult_t R = ncclAllGather(Send, Recv, BytesPerRank, ncclInt8, static_cast<ncclComm_t>(Comm), S);
   if (R != ncclSuccess) {
      LastError = std::string("ncclAllGather: ") + ncclGetErrorString(R);
      abort();
      return 1;
   }
   return 0;
}

int RcclComm::transport(void *Ctx, int NNghbr, const int *Tasks, void *const *SendPtrs, const size_t *SendBytes,
                        void *const *RecvPtrs, const size_t *RecvBytes, void *Stream) {
   return static_cast<RcclComm *>(Ctx)->exchange(NNghbr, Tasks, SendPtrs, SendBytes, RecvPtrs, RecvBytes,
                                                 static_cast<hipStream_t>(Stream));
}

} // namespace OMEGA
