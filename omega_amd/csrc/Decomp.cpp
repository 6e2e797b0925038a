// Decomp.cpp -- see Decomp.h.  Rules restated from the reference
// (components/omega/src/base/Decomp.cpp): cell halo layers :1017-1080, edge
// ownership + ordering :1476-1610, vertex ditto :1740-1880, XxOnCell compaction
// :2030-2064, EdgesOnEdge :2187-2199, global->local translation :553-712.
#include "Decomp.h"
#include "Partition.h"


#include <algorithm>
#include <array>
#include <numeric>

namespace OMEGA {

void Decomp::partitionRCB() { OMEGA::partitionRCB(G, NumTasks, CellTask); }

// The sequence that numbers cells inside every group: global id, or a Morton curve through the cell centres
// (21 bits per coordinate, interleaved; planar meshes have a constant z and reduce to the 2-D curve).
void Decomp::buildCellOrder() {
   CellSeq.resize(NCellsGlobal);
   std::iota(CellSeq.begin(), CellSeq.end(), 0);
   if (Order == LocalOrder::Curve || Order == LocalOrder::Hilbert) {
      OMEGA_REQUIRE(G.XCell && G.YCell, "Decomp: curve ordering needs cell coordinates");
      const R8 *C[3] = {G.XCell, G.YCell, G.ZCell};
      R8 Mn[3] = {0, 0, 0}, Sc[3] = {0, 0, 0};
      for (int A = 0; A < 3; ++A) {
         if (!C[A])
            continue;
         R8 Lo = 1e300, Hi = -1e300;
         for (I4 I = 0; I < NCellsGlobal; ++I)
            Lo = std::min(Lo, C[A][I]), Hi = std::max(Hi, C[A][I]);
         Mn[A] = Lo;
         Sc[A] = Hi > Lo ? 2097151.0 / (Hi - Lo) : 0.0;
      }
      // isotropic quantisation (the longest axis uses the 21 bits): keeps the curve's cells square
      const R8 S = std::min({Sc[0] > 0 ? Sc[0] : 1e300, Sc[1] > 0 ? Sc[1] : 1e300, Sc[2] > 0 ? Sc[2] : 1e300});
      auto Spread = [](unsigned long long V) { // 21 bits -> every third bit
         V &= 0x1fffffULL;
         V = (V | V << 32) & 0x1f00000000ffffULL;
         V = (V | V << 16) & 0x1f0000ff0000ffULL;
         V = (V | V << 8) & 0x100f00f00f00f00fULL;
         V = (V | V << 4) & 0x10c30c30c30c30c3ULL;
         V = (V | V << 2) & 0x1249249249249249ULL;
         return V;
      };
      // Hilbert index of a point of the 2^21-per-axis grid: the axes are brought into "transposed" form by the
      // usual sequence of conditional reflections / exchanges from the top bit down and a Gray decode, then
      // interleaved like the Morton key.  Axes with no extent do not take part.
      int Ax[3], ND = 0;
      for (int A = 0; A < 3; ++A)
         if (C[A] && Sc[A] > 0)
            Ax[ND++] = A;
      auto HilbertKey = [&](const unsigned long long *Q) {
         unsigned long long X[3] = {Q[0], Q[1], Q[2]};
         const unsigned long long Top = 1ULL << 20;
         for (unsigned long long B = Top; B > 1; B >>= 1) { // undo the excess work of the Gray code, top down
            const unsigned long long P = B - 1;
            for (int I = 0; I < ND; ++I) {
               if (X[I] & B) {
                  X[0] ^= P; // reflect the low bits of axis 0
               } else {      // exchange the low bits of axis 0 and axis I
                  const unsigned long long T = (X[0] ^ X[I]) & P;
                  X[0] ^= T;
                  X[I] ^= T;
               }
            }
         }
         for (int I = 1; I < ND; ++I) // Gray encode
            X[I] ^= X[I - 1];
         unsigned long long T = 0;
         for (unsigned long long B = Top; B > 1; B >>= 1)
            if (X[ND - 1] & B)
               T ^= B - 1;
         for (int I = 0; I < ND; ++I)
            X[I] ^= T;
         unsigned long long K = 0; // axis 0 carries the most significant bit of every digit
         for (int I = 0; I < ND; ++I)
            K |= Spread(X[I]) << (ND - 1 - I);
         return K;
      };
      std::vector<unsigned long long> Key(NCellsGlobal);
      for (I4 I = 0; I < NCellsGlobal; ++I) {
         unsigned long long K = 0;
         if (Order == LocalOrder::Hilbert && ND > 0) {
            unsigned long long Q[3] = {0, 0, 0};
            for (int D = 0; D < ND; ++D)
               Q[D] = (unsigned long long)((C[Ax[D]][I] - Mn[Ax[D]]) * S) & 0x1fffffULL;
            K = HilbertKey(Q);
         } else {
            for (int A = 0; A < 3; ++A)
               if (C[A] && Sc[A] > 0)
                  K |= Spread((unsigned long long)((C[A][I] - Mn[A]) * S)) << A;
         }
         Key[I] = K;
      }
      std::sort(CellSeq.begin(), CellSeq.end(), [&](I4 A, I4 B) { return Key[A] < Key[B] || (Key[A] == Key[B] && A < B); });
   }
   CellRank.resize(NCellsGlobal);
   for (I4 I = 0; I < NCellsGlobal; ++I)
      CellRank[CellSeq[I]] = I;
}

// Recursive median bisection along the widest axis (see LocalOrder::KdTree).  Splits of more than 32 cells fall on a
// multiple of 32 nearest to the middle, 32 -> 16 + 16, 16 -> 8 + 8: aligned runs of 8, 16 and 32 cells are subtrees.
// Deterministic (ties by global id): every rank derives every rank's numbering.
void Decomp::kdOrder(std::vector<I4> &List, size_t Begin, size_t End) const {
   OMEGA_REQUIRE(G.XCell && G.YCell, "Decomp: k-d ordering needs cell coordinates");
   kdOrderRange(List, Begin, End);
}

void Decomp::kdOrderRange(std::vector<I4> &List, size_t Begin, size_t End) const {
   const R8 *Cd[3] = {G.XCell, G.YCell, G.ZCell};
   std::vector<std::pair<size_t, size_t>> Stack;
   Stack.emplace_back(Begin, End);
   while (!Stack.empty()) {
      const auto [Lo, Hi] = Stack.back();
      Stack.pop_back();
      const size_t M = Hi - Lo;
      if (M <= 1)
         continue;
      int Ax    = 0;
      R8 BestEx = -1;
      for (int A = 0; A < 3; ++A) {
         if (!Cd[A])
            continue;
         R8 Mn = 1e300, Mx = -1e300;
         for (size_t I = Lo; I < Hi; ++I)
            Mn = std::min(Mn, Cd[A][List[I]]), Mx = std::max(Mx, Cd[A][List[I]]);
         if (Mx - Mn > BestEx)
            BestEx = Mx - Mn, Ax = A;
      }
      if (M <= 8) {
         // A leaf.  std::nth_element fixes which cells a side holds, not how it arranges them: the leaf is SORTED with the
         // same (coordinate, global id) order, so the numbering is a function of the coordinates alone -- every rank
         // derives every rank's numbering (CellLocAll, the halo lists) whatever its libstdc++ does inside nth_element.
         const R8 *X = Cd[Ax];
         std::sort(List.begin() + Lo, List.begin() + Hi, [X](I4 A, I4 B) { return X[A] < X[B] || (X[A] == X[B] && A < B); });
         continue;
      }
      size_t NL;
      if (M > 32) {
         NL = (M / 2 + 16) / 32 * 32;
         if (NL == 0)
            NL = 32;
         if (NL >= M)
            NL = (M - 1) / 32 * 32;
      } else {
         NL = M > 16 ? 16 : 8;
      }
      const R8 *X = Cd[Ax];
      std::nth_element(List.begin() + Lo, List.begin() + Lo + NL, List.begin() + Hi,
                       [X](I4 A, I4 B) { return X[A] < X[B] || (X[A] == X[B] && A < B); });
      Stack.emplace_back(Lo + NL, Hi);
      Stack.emplace_back(Lo, Lo + NL);
   }
}

// Owner task and local address on the owner for every global cell/edge/vertex.
void Decomp::computeOwnership() {
   // every task's owned cells in numbering order: the sequence (global-id order in the reference, :1000-1015), regrouped
   // wave by wave where asked for
   OwnedSeq.assign(NumTasks, {});
   for (I4 C : CellSeq)
      OwnedSeq[CellTask[C]].push_back(C);
   CellLocAll.assign(NCellsGlobal, 0);
   for (int T = 0; T < NumTasks; ++T) {
      orderGroup(OwnedSeq[T], 0, OwnedSeq[T].size());
      for (size_t L = 0; L < OwnedSeq[T].size(); ++L)
         CellLocAll[OwnedSeq[T][L]] = (I4)L;
   }

   // edge owner = task of the first valid cell in CellsOnEdge (:1476-1486)
   EdgeTask.assign(NEdgesGlobal, -1);
   for (I4 E = 0; E < NEdgesGlobal; ++E)
      for (int J = 0; J < 2; ++J) {
         I4 C = G.CellsOnEdge[2 * (size_t)E + J];
         if (C >= 0 && C < NCellsGlobal) {
            EdgeTask[E] = CellTask[C];
            break;
         }
      }
   VertexTask.assign(NVerticesGlobal, -1);
   for (I4 V = 0; V < NVerticesGlobal; ++V)
      for (int J = 0; J < VertexDegree; ++J) {
         I4 C = G.CellsOnVertex[(size_t)V * VertexDegree + J];
         if (C >= 0 && C < NCellsGlobal) {
            VertexTask[V] = CellTask[C];
            break;
         }
      }
   // owned edges / vertices are numbered in order of encounter around the owner's
   // owned cells (:1559-1583, :1824-1850)
   EdgeLocAll.assign(NEdgesGlobal, -1);
   VertexLocAll.assign(NVerticesGlobal, -1);
   std::vector<I4> ECount(NumTasks, 0), VCount(NumTasks, 0);
   for (int T = 0; T < NumTasks; ++T)
   for (I4 C : OwnedSeq[T]) {
      for (int J = 0; J < MaxEdges; ++J) {
         I4 E = G.EdgesOnCell[(size_t)C * MaxEdges + J];
         if (E >= 0 && E < NEdgesGlobal && EdgeTask[E] == T && EdgeLocAll[E] < 0)
            EdgeLocAll[E] = ECount[T]++;
         I4 V = G.VerticesOnCell[(size_t)C * MaxEdges + J];
         if (V >= 0 && V < NVerticesGlobal && VertexTask[V] == T && VertexLocAll[V] < 0)
            VertexLocAll[V] = VCount[T]++;
      }
   }
}

LocalSets Decomp::computeLocalSets(I4 Task) const {
   LocalSets S;
   // ---- cells: owned in global order, then HaloWidth BFS layers, each sorted ----
   std::vector<char> InList(NCellsGlobal, 0);
   S.CellID = OwnedSeq[Task];
   for (I4 C : S.CellID)
      InList[C] = 1;
   S.NCellsOwned = (I4)S.CellID.size();
   S.NCellsHalo.assign(HaloWidth, 0);
   size_t Start = 0, End = S.CellID.size();
   for (int Halo = 0; Halo < HaloWidth; ++Halo) {
      std::vector<I4> Layer;
      for (size_t L = Start; L < End; ++L) {
         const I4 C = S.CellID[L];
         for (int J = 0; J < MaxEdges; ++J) {
            I4 Nbr = G.CellsOnCell[(size_t)C * MaxEdges + J];
            if (Nbr < 0 || Nbr >= NCellsGlobal)
               continue;
            if (CellTask[Nbr] != Task && !InList[Nbr]) {
               InList[Nbr] = 1;
               Layer.push_back(Nbr);
            }
         }
      }
      std::sort(Layer.begin(), Layer.end(), [&](I4 A, I4 B) { return CellRank[A] < CellRank[B]; });
      S.CellID.insert(S.CellID.end(), Layer.begin(), Layer.end());
      orderGroup(S.CellID, S.CellID.size() - Layer.size(), S.CellID.size());
      S.NCellsHalo[Halo] = (I4)S.CellID.size();
      Start              = End;
      End                = S.CellID.size();
   }
   const I4 NCAll = (I4)S.CellID.size();

   // ---- edges and vertices: same scheme, through EdgesOnCell / VerticesOnCell ----
   auto Build = [&](const I4 *XOnCell, I4 NGlobal, const std::vector<I4> &XTask, std::vector<I4> &ID,
                    I4 &NOwned, std::vector<I4> &NHalo) {
      std::vector<char> State(NGlobal, 0); // 1 = needed & unprocessed, 2 = processed
      I4 NAll = 0, NOwnedHalo1 = 0;
      for (I4 L = 0; L < NCAll; ++L) {
         const I4 C = S.CellID[L];
         for (int J = 0; J < MaxEdges; ++J) {
            I4 X = XOnCell[(size_t)C * MaxEdges + J];
            if (X < 0 || X >= NGlobal)
               continue;
            if (!State[X]) {
               State[X] = 1;
               ++NAll;
               if (L < S.NCellsOwned)
                  ++NOwnedHalo1;
            }
         }
      }
      // NOwnedHalo1 counted elements first seen on an owned cell; since owned cells come
      // first in CellID this equals |elements around owned cells|.
      ID.assign(NAll, -1);
      NHalo.assign(HaloWidth, 0);
      NHalo[0]      = NOwnedHalo1;
      I4 Count      = 0;
      I4 HaloCount  = NOwnedHalo1 - 1; // first halo level is stored in reverse, from the end
      for (I4 L = 0; L < S.NCellsOwned; ++L) {
         const I4 C = S.CellID[L];
         for (int J = 0; J < MaxEdges; ++J) {
            I4 X = XOnCell[(size_t)C * MaxEdges + J];
            if (X < 0 || X >= NGlobal || State[X] != 1)
               continue;
            State[X] = 2;
            if (XTask[X] == Task)
               ID[Count++] = X;
            else
               ID[HaloCount--] = X;
         }
      }
      NOwned       = Count;
      I4 CellStart = S.NCellsOwned;
      HaloCount    = NHalo[0];
      for (int Halo = 0; Halo < HaloWidth; ++Halo) {
         const I4 CellEnd = S.NCellsHalo[Halo];
         for (I4 L = CellStart; L < CellEnd; ++L) {
            const I4 C = S.CellID[L];
            for (int J = 0; J < MaxEdges; ++J) {
               I4 X = XOnCell[(size_t)C * MaxEdges + J];
               if (X < 0 || X >= NGlobal || State[X] != 1)
                  continue;
               State[X]        = 2;
               ID[HaloCount++] = X;
            }
         }
         CellStart = CellEnd;
         if (Halo + 1 < HaloWidth)
            NHalo[Halo + 1] = HaloCount;
      }
      OMEGA_REQUIRE(HaloCount == NAll, "Decomp: element list construction lost elements");
   };
   Build(G.EdgesOnCell, NEdgesGlobal, EdgeTask, S.EdgeID, S.NEdgesOwned, S.NEdgesHalo);
   Build(G.VerticesOnCell, NVerticesGlobal, VertexTask, S.VertexID, S.NVerticesOwned, S.NVerticesHalo);
   return S;
}

void Decomp::buildLocalConnectivity(const LocalSets &S) {
   // global -> local maps; anything not local (or missing) maps to the sentinel NXxAll
   std::vector<I4> G2LC(NCellsGlobal + 1, NCellsAll), G2LE(NEdgesGlobal + 1, NEdgesAll),
       G2LV(NVerticesGlobal + 1, NVerticesAll);
   for (I4 L = 0; L < NCellsAll; ++L)
      G2LC[S.CellID[L]] = L;
   for (I4 L = 0; L < NEdgesAll; ++L)
      G2LE[S.EdgeID[L]] = L;
   for (I4 L = 0; L < NVerticesAll; ++L)
      G2LV[S.VertexID[L]] = L;
   auto MapC = [&](I4 X) { return (X >= 0 && X < NCellsGlobal) ? G2LC[X] : NCellsAll; };
   auto MapE = [&](I4 X) { return (X >= 0 && X < NEdgesGlobal) ? G2LE[X] : NEdgesAll; };
   auto MapV = [&](I4 X) { return (X >= 0 && X < NVerticesGlobal) ? G2LV[X] : NVerticesAll; };

   const int ME = MaxEdges, ME2 = 2 * MaxEdges, VD = VertexDegree;
   CellsOnCellH    = HostArrayI4(NCellsSize, ME, 1, NCellsAll);
   EdgesOnCellH    = HostArrayI4(NCellsSize, ME, 1, NEdgesAll);
   VerticesOnCellH = HostArrayI4(NCellsSize, ME, 1, NVerticesAll);
   NEdgesOnCellH   = HostArrayI4(NCellsSize, 1, 1, 0);
   for (I4 L = 0; L < NCellsAll; ++L) {
      const size_t C = S.CellID[L];
      int EdgeCount  = 0;
      for (int J = 0; J < ME; ++J) {
         CellsOnCellH(L, J)    = MapC(G.CellsOnCell[C * ME + J]);
         VerticesOnCellH(L, J) = MapV(G.VerticesOnCell[C * ME + J]);
         const I4 E            = G.EdgesOnCell[C * ME + J];
         if (E >= 0 && E < NEdgesGlobal) // only active edges are stored, and counted (:2043-2064)
            EdgesOnCellH(L, EdgeCount++) = MapE(E);
      }
      NEdgesOnCellH(L) = EdgeCount;
   }

   CellsOnEdgeH    = HostArrayI4(NEdgesSize, 2, 1, NCellsAll);
   VerticesOnEdgeH = HostArrayI4(NEdgesSize, 2, 1, NVerticesAll);
   EdgesOnEdgeH    = HostArrayI4(NEdgesSize, ME2, 1, NEdgesAll);
   NEdgesOnEdgeH   = HostArrayI4(NEdgesSize, 1, 1, 0);
   for (I4 L = 0; L < NEdgesAll; ++L) {
      const size_t E = S.EdgeID[L];
      for (int J = 0; J < 2; ++J) {
         CellsOnEdgeH(L, J)    = MapC(G.CellsOnEdge[E * 2 + J]);
         VerticesOnEdgeH(L, J) = MapV(G.VerticesOnEdge[E * 2 + J]);
      }
      // Missing entries stay in place as the sentinel (:2187-2199).  The reference counts
      // every slot (zero padding included) in NEdgesOnEdge; trailing sentinel slots only
      // ever add exact zeros (weight 0 x sentinel row 0), so they are trimmed here.
      int Last = 0;
      for (int J = 0; J < ME2; ++J) {
         const I4 X         = G.EdgesOnEdge[E * ME2 + J];
         EdgesOnEdgeH(L, J) = MapE(X);
         if (X >= 0 && X < NEdgesGlobal)
            Last = J + 1;
      }
      NEdgesOnEdgeH(L) = Last;
   }

   CellsOnVertexH = HostArrayI4(NVerticesSize, VD, 1, NCellsAll);
   EdgesOnVertexH = HostArrayI4(NVerticesSize, VD, 1, NEdgesAll);
   for (I4 L = 0; L < NVerticesAll; ++L) {
      const size_t V = S.VertexID[L];
      for (int J = 0; J < VD; ++J) {
         CellsOnVertexH(L, J) = MapC(G.CellsOnVertex[V * VD + J]);
         EdgesOnVertexH(L, J) = MapE(G.EdgesOnVertex[V * VD + J]);
      }
   }
}

Decomp::Decomp(const GlobalMeshDesc &Mesh, I4 NParts, I4 MyTask_, I4 HaloWidth_, const I4 *UserCellTask,
               LocalOrder Order_)
    : HaloWidth(HaloWidth_), NumTasks(NParts), MyTask(MyTask_), Order(Order_), G(Mesh) {
   OMEGA_REQUIRE(NParts >= 1 && MyTask_ >= 0 && MyTask_ < NParts, "Decomp: bad task / part count");
   OMEGA_REQUIRE(HaloWidth_ >= 1, "Decomp: HaloWidth must be >= 1");
   OMEGA_REQUIRE(G.NCells > 0 && G.CellsOnCell && G.EdgesOnCell && G.VerticesOnCell && G.CellsOnEdge &&
                     G.VerticesOnEdge && G.EdgesOnEdge && G.CellsOnVertex && G.EdgesOnVertex,
                 "Decomp: incomplete global mesh connectivity");
   NCellsGlobal    = G.NCells;
   NEdgesGlobal    = G.NEdges;
   NVerticesGlobal = G.NVertices;
   MaxEdges        = G.MaxEdges;
   VertexDegree    = G.VertexDegree;

   CellTask.assign(NCellsGlobal, 0);
   if (UserCellTask) {
      for (I4 C = 0; C < NCellsGlobal; ++C) {
         OMEGA_REQUIRE(UserCellTask[C] >= 0 && UserCellTask[C] < NParts, "Decomp: cell task out of range");
         CellTask[C] = UserCellTask[C];
      }
   } else if (NParts > 1) {
      partitionRCB();
   }
   buildCellOrder();
   computeOwnership();

   LocalSets S    = computeLocalSets(MyTask);
   NCellsOwned    = S.NCellsOwned;
   NCellsAll      = (I4)S.CellID.size();
   NCellsSize     = NCellsAll + 1;
   NEdgesOwned    = S.NEdgesOwned;
   NEdgesAll      = (I4)S.EdgeID.size();
   NEdgesSize     = NEdgesAll + 1;
   NVerticesOwned = S.NVerticesOwned;
   NVerticesAll   = (I4)S.VertexID.size();
   NVerticesSize  = NVerticesAll + 1;

   auto Fill = [&](HostArrayI4 &NHaloH, const std::vector<I4> &NHalo, HostArrayI4 &IDH, HostArrayI4 &LocH,
                   const std::vector<I4> &ID, I4 NAll, I4 NGlobal, const std::vector<I4> &TaskOf,
                   const std::vector<I4> &LocOf) {
      NHaloH = HostArrayI4(HaloWidth);
      for (int I = 0; I < HaloWidth; ++I)
         NHaloH(I) = NHalo[I];
      IDH  = HostArrayI4(NAll + 1, 1, 1, NGlobal + 1);
      LocH = HostArrayI4(NAll + 1, 2, 1, 0);
      for (I4 L = 0; L < NAll; ++L) {
         IDH(L)     = ID[L] + 1; // 1-based global ids, as in the reference
         LocH(L, 0) = TaskOf[ID[L]];
         LocH(L, 1) = LocOf[ID[L]];
      }
      LocH(NAll, 0) = MyTask;
      LocH(NAll, 1) = NAll;
   };
   Fill(NCellsHaloH, S.NCellsHalo, CellIDH, CellLocH, S.CellID, NCellsAll, NCellsGlobal, CellTask, CellLocAll);
   Fill(NEdgesHaloH, S.NEdgesHalo, EdgeIDH, EdgeLocH, S.EdgeID, NEdgesAll, NEdgesGlobal, EdgeTask, EdgeLocAll);
   Fill(NVerticesHaloH, S.NVerticesHalo, VertexIDH, VertexLocH, S.VertexID, NVerticesAll, NVerticesGlobal,
        VertexTask, VertexLocAll);

   buildLocalConnectivity(S);
}

} // namespace OMEGA
