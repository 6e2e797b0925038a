// Halo.cpp -- see Halo.h.
#include "Halo.h"
#include "PeerWire.h"
#include "Rccl.h"
#include "kernels/Kernels.h"

#include <algorithm>
#include <set>

namespace OMEGA {

Halo::Halo(const std::string &, const Decomp *D) {
   MyTask    = D->MyTask;
   NumTasks  = D->NumTasks;
   HaloWidth = D->HaloWidth;

   // my halo elements, by kind: (NOwned, NAll, Loc)
   struct Kind {
      I4 NOwned, NAll;
      const HostArrayI4 *Loc;
      const std::vector<I4> *TaskOf, *LocOf;
   };
   const Kind Kinds[3] = {{D->NCellsOwned, D->NCellsAll, &D->CellLocH, &D->CellTask, &D->CellLocAll},
                          {D->NEdgesOwned, D->NEdgesAll, &D->EdgeLocH, &D->EdgeTask, &D->EdgeLocAll},
                          {D->NVerticesOwned, D->NVerticesAll, &D->VertexLocH, &D->VertexTask, &D->VertexLocAll}};

   // Other ranks' ordered element lists (derived locally; the reference exchanges them
   // over MPI, Halo.cpp:331-442, 566-600).
   std::vector<LocalSets> Sets(NumTasks);
   std::set<I4> Nbrs;
   for (int Kd = 0; Kd < 3; ++Kd)
      for (I4 I = Kinds[Kd].NOwned; I < Kinds[Kd].NAll; ++I)
         Nbrs.insert((*Kinds[Kd].Loc)(I, 0));
   std::vector<char> HaveSet(NumTasks, 0);
   for (int T = 0; T < NumTasks; ++T) {
      if (T == MyTask)
         continue;
      Sets[T]    = D->computeLocalSets(T);
      HaveSet[T] = 1;
      // does T's halo hold anything I own?
      const std::vector<I4> *IDs[3]  = {&Sets[T].CellID, &Sets[T].EdgeID, &Sets[T].VertexID};
      const I4 Owned[3]              = {Sets[T].NCellsOwned, Sets[T].NEdgesOwned, Sets[T].NVerticesOwned};
      for (int Kd = 0; Kd < 3; ++Kd)
         for (size_t I = Owned[Kd]; I < IDs[Kd]->size(); ++I)
            if ((*Kinds[Kd].TaskOf)[(*IDs[Kd])[I]] == MyTask) {
               Nbrs.insert(T);
               break;
            }
   }
   Nbrs.erase(MyTask);
   NeighborList.assign(Nbrs.begin(), Nbrs.end());
   NNghbr = (I4)NeighborList.size();

   for (int Kd = 0; Kd < 3; ++Kd) {
      SendLists[Kd].assign(NNghbr, {});
      RecvLists[Kd].assign(NNghbr, {});
      // receive lists: my halo elements owned by each neighbour, in local index order
      // (= by halo layer, then order within the layer)
      for (I4 I = Kinds[Kd].NOwned; I < Kinds[Kd].NAll; ++I) {
         const I4 T  = (*Kinds[Kd].Loc)(I, 0);
         const int N = (int)(std::lower_bound(NeighborList.begin(), NeighborList.end(), T) - NeighborList.begin());
         RecvLists[Kd][N].push_back(I);
      }
      // send lists: the neighbour's receive list, translated to my local addresses
      for (int N = 0; N < NNghbr; ++N) {
         const LocalSets &S            = Sets[NeighborList[N]];
         const std::vector<I4> &IDs    = Kd == 0 ? S.CellID : (Kd == 1 ? S.EdgeID : S.VertexID);
         const I4 Owned                = Kd == 0 ? S.NCellsOwned : (Kd == 1 ? S.NEdgesOwned : S.NVerticesOwned);
         for (size_t I = Owned; I < IDs.size(); ++I)
            if ((*Kinds[Kd].TaskOf)[IDs[I]] == MyTask)
               SendLists[Kd][N].push_back((*Kinds[Kd].LocOf)[IDs[I]]);
      }
   }
   // where my message lands in each neighbour's receive buffer: its layout is neighbour-major in ascending task
   // order (planFor), and every owner of one of its halo elements is one of its neighbours
   PeerRecvPrefix.assign(NNghbr, {0, 0, 0});
   for (int N = 0; N < NNghbr; ++N) {
      const LocalSets &S             = Sets[NeighborList[N]];
      const std::vector<I4> *IDs[3]  = {&S.CellID, &S.EdgeID, &S.VertexID};
      const I4 Owned[3]              = {S.NCellsOwned, S.NEdgesOwned, S.NVerticesOwned};
      for (int Kd = 0; Kd < 3; ++Kd)
         for (size_t I = Owned[Kd]; I < IDs[Kd]->size(); ++I)
            if ((*Kinds[Kd].TaskOf)[(*IDs[Kd])[I]] < MyTask)
               ++PeerRecvPrefix[N][Kd];
   }
   SendPtrs.assign(NNghbr, nullptr);
   RecvPtrs.assign(NNghbr, nullptr);
}

// Device-side resources exist from the moment a wire is chosen (a Halo without a wire is a host object: its lists are
// used by CPU-side tools and tests): not at the first exchange -- no resource is created inside a time step.
void Halo::ensureWireResources() {
   if (NumTasks > 1 && !GatherIn.Ptr) {
      GatherIn  = Array1DReal("HaloGatherIn", 2 * MaxSumPairs);
      GatherOut = Array1DReal("HaloGatherOut", NumTasks * 2 * MaxSumPairs);
   }
   if (NNghbr > 0 && !EvLast) {
      HIP_CHECK(hipEventCreateWithFlags(&EvLast, hipEventDisableTiming));
      noteDeviceResource();
   }
}

Halo::~Halo() {
   unbindPeer();
   if (EvLast)
      (void)hipEventDestroy(EvLast);
}

void Halo::unbindPeer() {
   if (Peer && Peer->BoundTo == this)
      Peer->BoundTo = nullptr;
   Peer = nullptr;
}

// ~PeerWire of the wire bound to this Halo: from here on the Halo has no wire
void Halo::peerWireGone(const PeerWire *Wire) {
   if (Peer == Wire)
      Peer = nullptr;
}

void Halo::setTransport(HaloTransportFn Fn, void *Ctx) {
   unbindPeer();
   if (Fn)
      ensureWireResources();
   Rccl         = nullptr;
   Transport    = Fn;
   TransportCtx = Ctx;
}

void Halo::useRccl(RcclComm *Comm) {
   OMEGA_REQUIRE(Comm != nullptr, "Halo::useRccl: no communicator");
   OMEGA_REQUIRE(Comm->Rank == MyTask, "Halo::useRccl: the communicator's rank is not this Halo's task");
   for (I4 T : NeighborList)
      OMEGA_REQUIRE(T < Comm->NRanks, "Halo::useRccl: a neighbour task is outside the communicator");
   setTransport(&RcclComm::transport, Comm);
   Rccl = Comm;
}

void Halo::usePeerWire(PeerWire *Wire) {
   OMEGA_REQUIRE(Wire != nullptr && Wire->connected(), "Halo::usePeerWire: the wire is not connected");
   OMEGA_REQUIRE(Wire->Rank == MyTask, "Halo::usePeerWire: the wire's rank is not this Halo's task");
   OMEGA_REQUIRE(NNghbr <= PeerWire::MaxPeers, "Halo::usePeerWire: too many neighbours");
   OMEGA_REQUIRE(!Wire->bound() || Wire->BoundTo == this, "Halo::usePeerWire: this wire already serves another Halo");
   for (I4 T : NeighborList)
      OMEGA_REQUIRE(T < Wire->NRanks, "Halo::usePeerWire: a neighbour task is outside the wire");
   unbindPeer();
   ensureWireResources();
   Wire->BoundTo = this;
   Peer         = Wire;
   Rccl         = nullptr;
   Transport    = nullptr;
   TransportCtx = nullptr;
}

I4 Halo::checkWire() const { return (Peer && Peer->status() != 0) ? -1 : 0; }

I4 Halo::globalSumDD(const double *LocalPairs, int NPairs, double *HiLo, hipStream_t S) {
   OMEGA_REQUIRE(LocalPairs && HiLo && NPairs >= 0 && NPairs <= MaxSumPairs, "Halo::globalSumDD: 0..64 pairs per call");
   SumError.clear();
   if (NPairs == 0)
      return 0;
   if (NumTasks == 1) {
      for (int I = 0; I < NPairs; ++I)
         combineDD(LocalPairs + 2 * I, 1, HiLo + 2 * I);
      return 0;
   }
   if (!Peer && !Rccl) {
      SumError = ": Halo::globalSumDD needs the RCCL or the peer wire (a caller-supplied transport has no all-gather)";
      return -1;
   }
   const int NVals = 2 * NPairs;
   HIP_CHECK(hipMemcpyAsync(GatherIn.Ptr, LocalPairs, NVals * sizeof(double), hipMemcpyHostToDevice, S));
   const int Err = Peer ? Peer->allGather(GatherIn.Ptr, NVals, GatherOut.Ptr, S)
                        : Rccl->allGather(GatherIn.Ptr, GatherOut.Ptr, NVals * sizeof(double), S);
   if (Err != 0) {
      if (Rccl)
         SumError = ": " + Rccl->lastError();
      return -1;
   }
   std::vector<double> All((size_t)NumTasks * NVals), Pairs((size_t)NumTasks * 2);
   HIP_CHECK(hipMemcpyAsync(All.data(), GatherOut.Ptr, All.size() * sizeof(double), hipMemcpyDeviceToHost, S));
   HIP_CHECK(hipStreamSynchronize(S));
   if (checkWire() != 0)
      return -1;
   for (int I = 0; I < NPairs; ++I) { // rank order: the same bits on every rank
      for (int R = 0; R < NumTasks; ++R)
         Pairs[2 * R] = All[(size_t)R * NVals + 2 * I], Pairs[2 * R + 1] = All[(size_t)R * NVals + 2 * I + 1];
      combineDD(Pairs.data(), NumTasks, HiLo + 2 * I);
   }
   return 0;
}

std::string Halo::wireError() const {
   if (!SumError.empty())
      return SumError;
   if (!Peer)
      return "";
   if (!Peer->lastError().empty())
      return ": " + Peer->lastError();
   if (int St = Peer->status())
      return ": PeerWire: a wait for a peer gave up (status " + std::to_string(St) +
             ": bit 0 (1) = my previous message was never consumed, bit 1 (2) = a neighbour's message never arrived, bit 2 (4) "
             "= an all-gather of a global sum never got a rank's values); after bit 0 / 1 the halo was left as it was";
   return "";
}

size_t Halo::recvRows(size_t NTC, size_t NTE, size_t NTV) const {
   size_t R = 0;
   for (int N = 0; N < NNghbr; ++N)
      R += RecvLists[0][N].size() * NTC + RecvLists[1][N].size() * NTE + RecvLists[2][N].size() * NTV;
   return R;
}

size_t Halo::requiredBytes(int N, size_t TC, size_t TE, size_t TV) const {
   const size_t S = SendLists[0][N].size() * TC + SendLists[1][N].size() * TE + SendLists[2][N].size() * TV;
   const size_t R = RecvLists[0][N].size() * TC + RecvLists[1][N].size() * TE + RecvLists[2][N].size() * TV;
   return std::max(S, R) * sizeof(Real);
}

void Halo::ensureBuffers(size_t SendBytes, size_t RecvBytes) {
   const bool GrowS = !SendBuf || SendBuf->Bytes < SendBytes, GrowR = !RecvBuf || RecvBuf->Bytes < RecvBytes;
   if (!GrowS && !GrowR)
      return;
   HIP_CHECK(hipDeviceSynchronize()); // growing: nothing may still use the old buffers
   if (GrowS)
      SendBuf = std::make_shared<DeviceBuffer>(SendBytes);
   if (GrowR)
      RecvBuf = std::make_shared<DeviceBuffer>(RecvBytes);
}

const Halo::Plan &Halo::planFor(const std::vector<Piece> &Pieces) {
   std::vector<int> Key;
   for (const Piece &P : Pieces) {
      Key.push_back((int)P.Elem), Key.push_back(P.NT), Key.push_back(P.RowsSize), Key.push_back(P.K);
      Key.push_back(P.Pitch), Key.push_back(P.ElemBytes);
      OMEGA_REQUIRE(P.K == Pieces[0].K && P.Pitch == Pieces[0].Pitch && P.Pitch >= P.K && P.ElemBytes == Pieces[0].ElemBytes,
                    "Halo: arrays exchanged together must have the same number of levels, row pitch and element size");
      OMEGA_REQUIRE(P.ElemBytes == 4 || P.ElemBytes == 8, "Halo: element size must be 4 or 8 bytes");
   }
   auto It = Plans.find(Key);
   if (It != Plans.end())
      return It->second;
   Plan Pl;
   Pl.K = Pieces[0].K, Pl.Pitch = Pieces[0].Pitch, Pl.ElemBytes = Pieces[0].ElemBytes;
   Pl.SendOff.assign(NNghbr, 0), Pl.RecvOff.assign(NNghbr, 0);
   Pl.SendBytes.assign(NNghbr, 0), Pl.RecvBytes.assign(NNghbr, 0);
   const size_t RowBytes = (size_t)Pl.K * Pl.ElemBytes;
   auto Build            = [&](const std::vector<std::vector<I4>> *Lists, std::vector<size_t> &Off,
                    std::vector<size_t> &Bytes, const char *Nm, size_t &NRows) {
      std::vector<I4> Jobs;
      for (int N = 0; N < NNghbr; ++N) {
         Off[N] = Jobs.size() / 2 * RowBytes;
         for (size_t Ip = 0; Ip < Pieces.size(); ++Ip) {
            const Piece &P            = Pieces[Ip];
            const std::vector<I4> &L = Lists[P.Elem][N];
            for (int T = 0; T < P.NT; ++T)
               for (I4 Row : L) {
                  const size_t Plane = (size_t)T * P.RowsSize + (size_t)Row;
                  OMEGA_REQUIRE(Plane < ((size_t)1 << 31), "Halo: array too large for the 32-bit job table");
                  Jobs.push_back((I4)Ip), Jobs.push_back((I4)Plane);
               }
         }
         Bytes[N] = Jobs.size() / 2 * RowBytes - Off[N];
      }
      NRows = Jobs.size() / 2;
      Array1DI4 D(Nm, (int)std::max<size_t>(Jobs.size(), 2));
      if (!Jobs.empty())
         copyToDevice(D.Ptr, Jobs.data(), Jobs.size() * sizeof(I4));
      return D;
   };
   Pl.RemoteOff.assign(NNghbr, 0);
   for (int N = 0; N < NNghbr; ++N)
      for (const Piece &P : Pieces)
         Pl.RemoteOff[N] += (size_t)P.NT * PeerRecvPrefix[N][P.Elem] * RowBytes;
   Pl.SendJobs = Build(SendLists, Pl.SendOff, Pl.SendBytes, "HaloSendJobs", Pl.NSendRows);
   Pl.RecvJobs = Build(RecvLists, Pl.RecvOff, Pl.RecvBytes, "HaloRecvJobs", Pl.NRecvRows);
   return Plans.emplace(Key, std::move(Pl)).first->second;
}

I4 Halo::exchangePieces(const std::vector<Piece> &Pieces, hipStream_t S) {
   if (NNghbr == 0)
      return 0;
   OMEGA_REQUIRE(Transport != nullptr || Peer != nullptr, "Halo: no transport set for a multi-rank exchange");
   OMEGA_REQUIRE(!Pieces.empty() && Pieces.size() <= (size_t)HaloMaxPieces, "Halo: too many arrays in one exchange");
   const Plan &Pl        = planFor(Pieces);
   const size_t RowBytes = (size_t)Pl.K * Pl.ElemBytes;
   ensureBuffers(Pl.NSendRows * RowBytes, Peer ? 0 : Pl.NRecvRows * RowBytes);
   if (Peer)
      OMEGA_REQUIRE(Pl.NRecvRows * RowBytes <= Peer->mailboxBytes(),
                    "Halo: this exchange receives " + std::to_string(Pl.NRecvRows * RowBytes) +
                        " bytes, more than the peer wire's mailbox holds: create the wire with Halo::recvRows() * K * 8");
   char *const RecvBase = static_cast<char *>(Peer ? Peer->mailbox() : RecvBuf->Ptr);
   HaloBases B{};
   for (size_t I = 0; I < Pieces.size(); ++I)
      B.P[I] = Pieces[I].Ptr;
   for (int N = 0; N < NNghbr; ++N) {
      SendPtrs[N] = static_cast<char *>(SendBuf->Ptr) + Pl.SendOff[N];
      RecvPtrs[N] = RecvBase + Pl.RecvOff[N];
   }
   if (HaveLast)
      HIP_CHECK(hipStreamWaitEvent(S, EvLast, 0)); // the shared buffers are free once the previous exchange is done
   // pack: one launch for every neighbour and array (Halo.h:324-414)
   launchHaloPackAll(SendBuf->Ptr, B, Pl.SendJobs.Ptr, Pl.NSendRows, Pl.K, Pl.Pitch, Pl.ElemBytes, S);
   const int Err = Peer ? Peer->put(NNghbr, NeighborList.data(), SendPtrs.data(), Pl.SendBytes.data(),
                                    Pl.RemoteOff.data(), S)
                        : Transport(TransportCtx, NNghbr, NeighborList.data(), SendPtrs.data(), Pl.SendBytes.data(),
                                    RecvPtrs.data(), Pl.RecvBytes.data(), (void *)S);
   if (Err != 0)
      return -1;
   // unpack: one launch (Halo.h:566-653)
   launchHaloUnpackAll(B, RecvBase, Pl.RecvJobs.Ptr, Pl.NRecvRows, Pl.K, Pl.Pitch, Pl.ElemBytes, S,
                       Peer ? Peer->statusWord() : nullptr);
   if (Peer && Peer->release(NNghbr, NeighborList.data(), S) != 0)
      return -1;
   HIP_CHECK(hipEventRecord(EvLast, S));
   HaveLast = true;
   return 0;
}
I4 Halo::exchangeFullArrayHalo(const Array2DReal &A, MeshElement E, hipStream_t S) {
   return exchangePieces({Piece{A.Ptr, E, 1, A.Ext[0], A.Ext[1], A.Pitch}}, S);
}
I4 Halo::exchangeFullArrayHalo(const Array3DReal &A, MeshElement E, hipStream_t S) {
   return exchangePieces({Piece{A.Ptr, E, A.Ext[0], A.Ext[1], A.Ext[2], A.Pitch}}, S);
}
I4 Halo::exchangeRaw(Real *Ptr, int NT, int RowsSize, int K, int Pitch, MeshElement E, hipStream_t S) {
   return exchangePieces({Piece{Ptr, E, NT, RowsSize, K, Pitch > 0 ? Pitch : K}}, S);
}
I4 Halo::exchangeRawBytes(void *Ptr, int ElemBytes, int NT, int RowsSize, int K, int Pitch, MeshElement E, hipStream_t S) {
   Piece P{Ptr, E, NT, RowsSize, K, Pitch > 0 ? Pitch : K};
   P.ElemBytes = ElemBytes;
   return exchangePieces({P}, S);
}
I4 Halo::exchangeFullArrayHalo(const Array1DI4 &A, MeshElement E, hipStream_t S) {
   return exchangeRawBytes(A.Ptr, sizeof(I4), 1, A.Ext[0], 1, 1, E, S);
}
I4 Halo::exchangeFullArrayHalo(const Array2DI4 &A, MeshElement E, hipStream_t S) {
   return exchangeRawBytes(A.Ptr, sizeof(I4), 1, A.Ext[0], A.Ext[1], A.Pitch, E, S);
}
I4 Halo::exchangeFullArrayHalo(const Array1DReal &A, MeshElement E, hipStream_t S) {
   return exchangeRawBytes(A.Ptr, sizeof(Real), 1, A.Ext[0], 1, 1, E, S);
}
I4 Halo::exchangeState(const Array2DReal &H, const Array2DReal &U, const Array3DReal *Tr, int NT, hipStream_t S) {
   std::vector<Piece> P{Piece{H.Ptr, OnCell, 1, H.Ext[0], H.Ext[1], H.Pitch},
                        Piece{U.Ptr, OnEdge, 1, U.Ext[0], U.Ext[1], U.Pitch}};
   if (Tr && NT > 0)
      P.push_back(Piece{Tr->Ptr, OnCell, NT, Tr->Ext[1], Tr->Ext[2], Tr->Pitch});
   return exchangePieces(P, S);
}
void Halo::reserveState(const Array2DReal &H, const Array2DReal &U, const Array3DReal *Tr, int NT) {
   if (NNghbr == 0)
      return;
   std::vector<Piece> P{Piece{H.Ptr, OnCell, 1, H.Ext[0], H.Ext[1], H.Pitch},
                        Piece{U.Ptr, OnEdge, 1, U.Ext[0], U.Ext[1], U.Pitch}};
   if (Tr && NT > 0)
      P.push_back(Piece{Tr->Ptr, OnCell, NT, Tr->Ext[1], Tr->Ext[2], Tr->Pitch});
   const Plan &Pl        = planFor(P);
   const size_t RowBytes = (size_t)Pl.K * Pl.ElemBytes;
   ensureBuffers(Pl.NSendRows * RowBytes, Peer ? 0 : Pl.NRecvRows * RowBytes);
}

} // namespace OMEGA
