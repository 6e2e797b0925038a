// Halo.cpp -- see Halo.h.
#include "Halo.h"
#include "kernels/Kernels.h"

#include <algorithm>
#include <set>

namespace OMEGA {

Halo::Halo(const std::string &, const Decomp *D) {
   MyTask    = D->MyTask;
   HaloWidth = D->HaloWidth;
   const int NumTasks = D->NumTasks;

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
   SendBuf.assign(NNghbr, nullptr);
   RecvBuf.assign(NNghbr, nullptr);
   BufBytes.assign(NNghbr, 0);
   OwnedSend.resize(NNghbr);
   OwnedRecv.resize(NNghbr);
   External.assign(NNghbr, 0);
}

Halo::~Halo() {}

void Halo::ensureDevice() {
   if (DeviceReady)
      return;
   for (int Kd = 0; Kd < 3; ++Kd) {
      SendListsD[Kd].resize(NNghbr);
      RecvListsD[Kd].resize(NNghbr);
      for (int N = 0; N < NNghbr; ++N) {
         auto Up = [](const std::vector<I4> &V, const char *Nm) {
            Array1DI4 A(Nm, (int)std::max<size_t>(V.size(), 1));
            if (!V.empty())
               copyToDevice(A.Ptr, V.data(), V.size() * sizeof(I4));
            return A;
         };
         SendListsD[Kd][N] = Up(SendLists[Kd][N], "HaloSendList");
         RecvListsD[Kd][N] = Up(RecvLists[Kd][N], "HaloRecvList");
      }
   }
   DeviceReady = true;
}

void Halo::setBuffers(int N, void *SendPtr, void *RecvPtr, size_t Bytes) {
   OMEGA_REQUIRE(N >= 0 && N < NNghbr, "Halo::setBuffers: neighbour index out of range");
   SendBuf[N]  = SendPtr;
   RecvBuf[N]  = RecvPtr;
   BufBytes[N] = Bytes;
   External[N] = 1;
}

size_t Halo::requiredBytes(int N, size_t TC, size_t TE, size_t TV) const {
   const size_t S = SendLists[0][N].size() * TC + SendLists[1][N].size() * TE + SendLists[2][N].size() * TV;
   const size_t R = RecvLists[0][N].size() * TC + RecvLists[1][N].size() * TE + RecvLists[2][N].size() * TV;
   return std::max(S, R) * sizeof(Real);
}

void Halo::ensureBuffers(const std::vector<size_t> &Need) {
   for (int N = 0; N < NNghbr; ++N) {
      if (Need[N] <= BufBytes[N])
         continue;
      OMEGA_REQUIRE(!External[N], "Halo: caller-owned exchange buffer too small for this exchange");
      HIP_CHECK(hipDeviceSynchronize()); // growing: nothing may still use the old buffers
      OwnedSend[N] = std::make_shared<DeviceBuffer>(Need[N]);
      OwnedRecv[N] = std::make_shared<DeviceBuffer>(Need[N]);
      SendBuf[N]   = OwnedSend[N]->Ptr;
      RecvBuf[N]   = OwnedRecv[N]->Ptr;
      BufBytes[N]  = Need[N];
   }
}

I4 Halo::exchangePieces(const std::vector<Piece> &Pieces, hipStream_t S) {
   if (NNghbr == 0)
      return 0;
   OMEGA_REQUIRE(Transport != nullptr, "Halo: no transport set for a multi-rank exchange");
   ensureDevice();
   std::vector<size_t> SendBytes(NNghbr, 0), RecvBytes(NNghbr, 0), Need(NNghbr, 0);
   for (int N = 0; N < NNghbr; ++N) {
      for (const Piece &P : Pieces) {
         SendBytes[N] += SendLists[P.Elem][N].size() * (size_t)P.NT * P.K * sizeof(Real);
         RecvBytes[N] += RecvLists[P.Elem][N].size() * (size_t)P.NT * P.K * sizeof(Real);
      }
      Need[N] = std::max(SendBytes[N], RecvBytes[N]);
   }
   ensureBuffers(Need);
   // pack (Halo.h:324-414)
   for (int N = 0; N < NNghbr; ++N) {
      size_t Off = 0;
      for (const Piece &P : Pieces) {
         const int NList = (int)SendLists[P.Elem][N].size();
         launchHaloPack(reinterpret_cast<Real *>(static_cast<char *>(SendBuf[N]) + Off), P.Ptr,
                        SendListsD[P.Elem][N].Ptr, NList, P.NT, P.RowsSize, P.K, S);
         Off += (size_t)NList * P.NT * P.K * sizeof(Real);
      }
   }
   const int Err = Transport(TransportCtx, NNghbr, NeighborList.data(), SendBuf.data(), SendBytes.data(),
                             RecvBuf.data(), RecvBytes.data(), (void *)S);
   if (Err != 0)
      return -1;
   // unpack (Halo.h:566-653)
   for (int N = 0; N < NNghbr; ++N) {
      size_t Off = 0;
      for (const Piece &P : Pieces) {
         const int NList = (int)RecvLists[P.Elem][N].size();
         launchHaloUnpack(P.Ptr, reinterpret_cast<const Real *>(static_cast<char *>(RecvBuf[N]) + Off),
                          RecvListsD[P.Elem][N].Ptr, NList, P.NT, P.RowsSize, P.K, S);
         Off += (size_t)NList * P.NT * P.K * sizeof(Real);
      }
   }
   return 0;
}

I4 Halo::exchangeFullArrayHalo(const Array2DReal &A, MeshElement E, hipStream_t S) {
   return exchangePieces({Piece{A.Ptr, E, 1, A.Ext[0], A.Ext[1]}}, S);
}
I4 Halo::exchangeFullArrayHalo(const Array3DReal &A, MeshElement E, hipStream_t S) {
   return exchangePieces({Piece{A.Ptr, E, A.Ext[0], A.Ext[1], A.Ext[2]}}, S);
}
I4 Halo::exchangeState(const Array2DReal &H, const Array2DReal &U, const Array3DReal *Tr, int NT, hipStream_t S) {
   std::vector<Piece> P{Piece{H.Ptr, OnCell, 1, H.Ext[0], H.Ext[1]}, Piece{U.Ptr, OnEdge, 1, U.Ext[0], U.Ext[1]}};
   if (Tr && NT > 0)
      P.push_back(Piece{Tr->Ptr, OnCell, NT, Tr->Ext[1], Tr->Ext[2]});
   return exchangePieces(P, S);
}

} // namespace OMEGA
