// Halo.h -- halo exchange between the ranks of one node (one process per GPU).
//
// Interface after the reference's Halo (components/omega/src/base/Halo.h:258-279,
// 767-915): exchangeFullArrayHalo(Array, MeshElement).  Exchange lists follow
// components/omega/src/base/Halo.cpp:455-600: for each neighbour, the receive list holds
// my halo elements owned by it (by halo layer, then local index order) and the send list
// holds my owned elements in the order of the neighbour's receive list; messages are
// packed as Buf[(T*NList + I)*K + k] (Halo.h:344-351, 390-397).
//
// MI355X-first design: lists are derived locally (every rank can derive any rank's
// numbering, Decomp.h), packing/unpacking are HIP kernels on the caller's stream, and the
// wire is a pluggable transport invoked in stream order -- in production RCCL send/recv
// over xGMI issued through torch.distributed (backend "nccl" = RCCL) on the same HIP
// stream; no host polling, no device-wide fences (the reference fences the whole device
// and polls MPI_Test, Halo.cpp:703-757, Halo.h:851-907).  exchangeState() ships h, u and
// tracers of one exchange point as ONE message per neighbour (the reference makes three
// rounds).
#ifndef OMEGA_AMD_HALO_H
#define OMEGA_AMD_HALO_H

#include "Base.h"
#include "Decomp.h"

namespace OMEGA {

/// Ships one message to / from each neighbour.  Called in stream order on `Stream`:
/// send buffers are complete when work already queued on Stream has run; the transport
/// must make later work on Stream wait for the receives.  Returns 0 on success.
typedef int (*HaloTransportFn)(void *Ctx, int NNghbr, const int *Tasks, void *const *SendPtrs,
                               const size_t *SendBytes, void *const *RecvPtrs, const size_t *RecvBytes,
                               void *Stream);

class Halo {
 public:
   Halo(const std::string &Name, const Decomp *InDecomp);
   ~Halo();

   I4 MyTask, NNghbr = 0, HaloWidth;
   std::vector<I4> NeighborList; ///< sorted task ids

   /// Exchange lists [kind][neighbour], concatenated over halo layers (local indices).
   std::vector<std::vector<I4>> SendLists[3], RecvLists[3];

   void setTransport(HaloTransportFn Fn, void *Ctx) {
      Transport    = Fn;
      TransportCtx = Ctx;
   }
   /// Use caller-owned device buffers (e.g. torch tensors registered with RCCL).
   void setBuffers(int INghbr, void *SendPtr, void *RecvPtr, size_t Bytes);
   /// Bytes needed per neighbour for exchanging arrays of `TotSizeCell`, `TotSizeEdge`,
   /// `TotSizeVertex` values per element in one message.
   size_t requiredBytes(int INghbr, size_t TotSizeCell, size_t TotSizeEdge, size_t TotSizeVertex) const;

   I4 exchangeFullArrayHalo(const Array2DReal &A, MeshElement E, hipStream_t S);
   I4 exchangeFullArrayHalo(const Array3DReal &A, MeshElement E, hipStream_t S);
   /// One aggregated message per neighbour: [h on cells][u on edges][tracers on cells].
   I4 exchangeState(const Array2DReal &H, const Array2DReal &U, const Array3DReal *Tr, int NT, hipStream_t S);

 private:
   struct Piece {
      Real *Ptr;
      MeshElement Elem;
      int NT, RowsSize, K;
   };
   I4 exchangePieces(const std::vector<Piece> &Pieces, hipStream_t S);
   void ensureDevice();
   void ensureBuffers(const std::vector<size_t> &Need);

   HaloTransportFn Transport = nullptr;
   void *TransportCtx        = nullptr;
   bool DeviceReady          = false;
   std::vector<Array1DI4> SendListsD[3], RecvListsD[3];
   std::vector<void *> SendBuf, RecvBuf;
   std::vector<size_t> BufBytes;
   std::vector<std::shared_ptr<DeviceBuffer>> OwnedSend, OwnedRecv;
   std::vector<char> External;
};

} // namespace OMEGA
#endif
