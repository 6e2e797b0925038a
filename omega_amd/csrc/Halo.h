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
// numbering, Decomp.h); ONE pack kernel gathers every row that travels -- all neighbours, all
// arrays of the exchange point -- into one contiguous device buffer through a device-resident
// job table (per exchange signature: which array, which row), the wire is a pluggable transport
// invoked in stream order -- in production RcclComm (Rccl.h): grouped ncclSend / ncclRecv over
// xGMI issued from C++ on the same HIP stream -- and ONE unpack kernel scatters the received
// rows; no host polling, no device-wide fences (the reference fences the whole device and polls
// MPI_Test, Halo.cpp:703-757, Halo.h:851-907).  exchangeState() ships h, u and tracers of one
// exchange point as ONE message per neighbour (the reference makes three rounds).
#ifndef OMEGA_AMD_HALO_H
#define OMEGA_AMD_HALO_H

#include "Base.h"
#include "Decomp.h"

#include <array>
#include <map>

namespace OMEGA {

/// Ships one message to / from each neighbour.  Called in stream order on `Stream`:
/// send buffers are complete when work already queued on Stream has run; the transport
/// must make later work on Stream wait for the receives.  Returns 0 on success.
typedef int (*HaloTransportFn)(void *Ctx, int NNghbr, const int *Tasks, void *const *SendPtrs,
                               const size_t *SendBytes, void *const *RecvPtrs, const size_t *RecvBytes,
                               void *Stream);

class Halo : public Registry<Halo> {
 public:
   Halo(const std::string &Name, const Decomp *InDecomp);
   ~Halo();

   I4 MyTask, NumTasks = 1, NNghbr = 0, HaloWidth;
   std::vector<I4> NeighborList; ///< sorted task ids

   /// Exchange lists [kind][neighbour], concatenated over halo layers (local indices).
   std::vector<std::vector<I4>> SendLists[3], RecvLists[3];

   /// (choosing a wire replaces the one chosen before: a peer wire bound to this Halo is unbound)
   void setTransport(HaloTransportFn Fn, void *Ctx);
   /// production wire: grouped RCCL send / recv on the exchange's stream (the communicator outlives the Halo)
   void useRccl(class RcclComm *Comm);
   /// the other stream-ordered wire: direct peer copies into the neighbours' mailboxes (PeerWire.h); the mailbox of
   /// the wire replaces this Halo's receive buffer.  The wire must be connected and outlive the Halo.
   void usePeerWire(class PeerWire *Wire);
   /// rows of K values this rank receives in one exchange of NTCell / NTEdge / NTVertex arrays-per-element
   /// (to size a PeerWire mailbox: bytes = recvRows(...) * K * 8)
   size_t recvRows(size_t NTCell, size_t NTEdge, size_t NTVertex) const;
   /// ": <message of the wire>" after an exchange returned -1 through the peer wire, else ""
   std::string wireError() const;
   /// Bytes needed per neighbour for exchanging arrays of `TotSizeCell`, `TotSizeEdge`,
   /// `TotSizeVertex` values per element in one message.
   size_t requiredBytes(int INghbr, size_t TotSizeCell, size_t TotSizeEdge, size_t TotSizeVertex) const;

   I4 exchangeFullArrayHalo(const Array2DReal &A, MeshElement E, hipStream_t S);
   I4 exchangeFullArrayHalo(const Array3DReal &A, MeshElement E, hipStream_t S);
   /// the reference's signature (Halo.h:767: exchangeFullArrayHalo(Array, MeshElement)) for every array type above and
   /// below: queued on this Halo's `Stream` (default: the null stream, where the reference's Kokkos kernels run)
   hipStream_t Stream = nullptr;
   template <class ArrayT> I4 exchangeFullArrayHalo(const ArrayT &A, MeshElement E) { return exchangeFullArrayHalo(A, E, Stream); }
   /// caller-owned raw device array [NT][RowsSize][Pitch] of which K values per row are levels
   I4 exchangeRaw(Real *Ptr, int NT, int RowsSize, int K, int Pitch, MeshElement E, hipStream_t S);
   /// the same for the reference's other element types (Halo.h:304-760 packs I4 / I8 / R4 / R8 arrays of rank 1-5): an
   /// array of ElemBytes-byte values (4 or 8), [NT][RowsSize][Pitch]; rank 1 is NT = 1, K = Pitch = 1; ranks 4 and 5
   /// fold their leading extents into NT (the element index is always the second-to-last, Halo.h:418-470)
   I4 exchangeRawBytes(void *Ptr, int ElemBytes, int NT, int RowsSize, int K, int Pitch, MeshElement E, hipStream_t S);
   I4 exchangeFullArrayHalo(const Array1DI4 &A, MeshElement E, hipStream_t S);
   I4 exchangeFullArrayHalo(const Array2DI4 &A, MeshElement E, hipStream_t S);
   I4 exchangeFullArrayHalo(const Array1DReal &A, MeshElement E, hipStream_t S);
   /// One aggregated message per neighbour: [h on cells][u on edges][tracers on cells].
   I4 exchangeState(const Array2DReal &H, const Array2DReal &U, const Array3DReal *Tr, int NT, hipStream_t S);
   /// globalSum of the reference (base/Reductions.h:71-88, 150-190: MPI_Allreduce with the double-double operator) for
   /// NPairs (<= 64) local double-double partial sums at once: the ranks' (hi, lo) pairs are all-gathered over this
   /// Halo's wire -- ncclAllGather on the RCCL communicator, PeerWire::allGather on the peer wire -- and combined in rank
   /// order with the ddSum operator (combineDD), so every rank gets the same bits whatever the partition.  LocalPairs
   /// [NPairs][2] and HiLo [NPairs][2] are host arrays; synchronises S.  Collective.  One rank: the combination alone.
   /// Returns 0, or -1 (wireError() / the message of the OmegaError says why; a host-staged test transport has no
   /// all-gather).
   I4 globalSumDD(const double *LocalPairs, int NPairs, double *HiLo, hipStream_t S);
   static constexpr int MaxSumPairs = 64;
   /// Initialisation-time half of exchangeState for arrays of these shapes: builds the job tables and allocates the
   /// message buffers, so that the exchanges inside a time step allocate nothing (the reference allocates its buffers
   /// in the Halo constructor, Halo.cpp:92-134; here their size depends on what travels together).
   void reserveState(const Array2DReal &H, const Array2DReal &U, const Array3DReal *Tr, int NT);
   /// The wire's verdict after the host has synchronised with an exchange's stream: 0, or -1 when a peer-wire wait gave up
   /// (wireError() then names it).  exchange*() can only report what is known when the work is QUEUED.
   I4 checkWire() const;

 private:
   struct Piece {
      void *Ptr;
      MeshElement Elem;
      int NT, RowsSize, K, Pitch; ///< Pitch: row pitch of the array in values (>= K)
      int ElemBytes = 8;          ///< bytes per value (8: R8 / I8, 4: I4 / R4)
   };
   /// Everything about an exchange that does not depend on the array pointers: the job tables of the pack and
   /// the unpack kernel (one job = one row of K values: which piece, which row of its [NT*RowsSize][K] plane
   /// stack; job j owns buffer row j) and the per-neighbour message extents inside the contiguous buffers.
   /// Message layout per neighbour = the reference's, piece after piece: Buf[(T*NList + I)*K + k]
   /// (Halo.h:344-351, 390-397).
   struct Plan {
      int K = 0, Pitch = 0, ElemBytes = 8;
      size_t NSendRows = 0, NRecvRows = 0;
      Array1DI4 SendJobs, RecvJobs; ///< [NRows][2] = (piece, row)
      std::vector<size_t> SendOff, RecvOff, SendBytes, RecvBytes; ///< per neighbour, bytes
      std::vector<size_t> RemoteOff; ///< per neighbour: where my message starts in ITS receive buffer, bytes
   };
   const Plan &planFor(const std::vector<Piece> &Pieces);
   I4 exchangePieces(const std::vector<Piece> &Pieces, hipStream_t S);
   void unbindPeer();
 public:
   /// called by ~PeerWire of the wire bound to this Halo (the binding is two-way: PeerWire.h)
   void peerWireGone(const class PeerWire *Wire);
 private:
   void ensureWireResources();
   void ensureBuffers(size_t SendBytes, size_t RecvBytes);

   /// [neighbour][kind]: number of the NEIGHBOUR's halo elements of that kind owned by tasks below mine = rows (per
   /// array-per-element) that precede my message in its receive buffer.  Derived locally, like the lists.
   std::vector<std::array<size_t, 3>> PeerRecvPrefix;
   class PeerWire *Peer      = nullptr;
   class RcclComm *Rccl      = nullptr;
   Array1DReal GatherIn, GatherOut; ///< device scratch of globalSumDD: [2 * MaxSumPairs], [NumTasks][2 * MaxSumPairs]
   std::string SumError;
   HaloTransportFn Transport = nullptr;
   void *TransportCtx        = nullptr;
   std::map<std::vector<int>, Plan> Plans; ///< keyed by (Elem, NT, RowsSize, K) of every piece
   std::shared_ptr<DeviceBuffer> SendBuf, RecvBuf;
   /// end of the previous exchange: the message buffers are shared by all exchanges of this Halo, so an exchange
   /// issued on another stream first waits for the previous one (a no-op on the same stream)
   hipEvent_t EvLast = nullptr;
   bool HaveLast     = false;
   std::vector<void *> SendPtrs, RecvPtrs;
};

} // namespace OMEGA
#endif
