// Partition.cpp -- see Partition.h.
#include "Partition.h"

#include <algorithm>
#include <numeric>
#include <queue>

namespace OMEGA {

namespace {

struct Graph {
   const GlobalMeshDesc &G;
   int ME;
   I4 N;
   /// neighbour J of cell C, or -1
   I4 nbr(I4 C, int J) const {
      const I4 X = G.CellsOnCell[(size_t)C * ME + J];
      return (X >= 0 && X < N) ? X : -1;
   }
};

// breadth-first search inside the cells marked Part[c] == Tag; returns the visit order (one component, from Root)
void bfs(const Graph &Gr, const std::vector<I4> &Part, I4 Tag, I4 Root, std::vector<I4> &Order, std::vector<I4> &Mark,
         I4 Stamp) {
   Order.clear();
   Order.push_back(Root);
   Mark[Root] = Stamp;
   for (size_t Head = 0; Head < Order.size(); ++Head) {
      const I4 C = Order[Head];
      for (int J = 0; J < Gr.ME; ++J) {
         const I4 X = Gr.nbr(C, J);
         if (X >= 0 && Part[X] == Tag && Mark[X] != Stamp) {
            Mark[X] = Stamp;
            Order.push_back(X);
         }
      }
   }
}

// Fiduccia-Mattheyses style refinement of a bisection of `Cells` into Tag / NewTag
void refineBisection(const Graph &Gr, std::vector<I4> &Part, const std::vector<I4> &Cells, I4 Tag, I4 NewTag,
                     size_t NLeft, std::vector<I4> &Mark, I4 &Stamp) {

   // Fiduccia-Mattheyses style refinement of the bisection: passes of single-cell moves with the best gain first,
   // sizes kept within +-Tol of the target, each pass rolled back to its best prefix
   const size_t Target = NLeft, Total = Cells.size();
   const size_t Tol    = std::max<size_t>(1, Total / 1000); // a tenth of a percent per bisection level
   auto Gain           = [&](I4 C) {
      int Same = 0, Other = 0;
      for (int J = 0; J < Gr.ME; ++J) {
         const I4 X = Gr.nbr(C, J);
         if (X < 0 || (Part[X] != Tag && Part[X] != NewTag))
            continue;
         (Part[X] == Part[C] ? Same : Other)++;
      }
      return Other - Same;
   };
   size_t Left = NLeft;
   for (int Pass = 0; Pass < 8; ++Pass) {
      const I4 LockStamp = ++Stamp;
      std::priority_queue<std::pair<int, I4>> Q[2]; // [0]: cells of Tag, [1]: cells of NewTag
      for (I4 C : Cells) {
         const int Gn = Gain(C);
         if (Gn > -Gr.ME) // boundary cells only (interior cells have gain -degree)
            Q[Part[C] == Tag ? 0 : 1].push({Gn, -C});
      }
      std::vector<I4> Moves;
      long Cum = 0, Best = 0;
      size_t BestLen = 0, BestLeft = Left;
      for (size_t It = 0; It < Total; ++It) {
         // side to move from: the one that is too big, else the better top gain
         int Side = -1;
         for (int S = 0; S < 2; ++S) // drop stale / locked entries
            while (!Q[S].empty()) {
               const I4 C = -Q[S].top().second;
               if (Mark[C] == LockStamp || (Part[C] == Tag ? 0 : 1) != S || Gain(C) != Q[S].top().first)
                  Q[S].pop();
               else
                  break;
            }
         const bool CanFrom0 = !Q[0].empty() && Left > Target - std::min(Tol, Target);
         const bool CanFrom1 = !Q[1].empty() && Left < Target + Tol;
         if (CanFrom0 && CanFrom1)
            Side = (Q[0].top().first >= Q[1].top().first) ? 0 : 1;
         else if (CanFrom0)
            Side = 0;
         else if (CanFrom1)
            Side = 1;
         if (Side < 0)
            break;
         const int Gn = Q[Side].top().first;
         const I4 C   = -Q[Side].top().second;
         Q[Side].pop();
         if (Gn < -1 && Cum + Gn < Best - 8) // hill climbing budget exhausted
            break;
         Part[C] = Side == 0 ? NewTag : Tag;
         Left += Side == 0 ? -1 : 1;
         Mark[C] = LockStamp;
         Moves.push_back(C);
         Cum += Gn;
         if (Cum > Best) {
            Best = Cum, BestLen = Moves.size(), BestLeft = Left;
         }
         for (int J = 0; J < Gr.ME; ++J) { // neighbours' gains changed
            const I4 X = Gr.nbr(C, J);
            if (X >= 0 && (Part[X] == Tag || Part[X] == NewTag) && Mark[X] != LockStamp)
               Q[Part[X] == Tag ? 0 : 1].push({Gain(X), -X});
         }
      }
      for (size_t I = Moves.size(); I > BestLen; --I) { // roll back past the best prefix
         const I4 C = Moves[I - 1];
         Part[C]    = Part[C] == Tag ? NewTag : Tag;
      }
      Left = BestLeft;
      if (Best <= 0)
         break;
   }
}

// Split the cells of `Cells` (all with Part == Tag) into Tag (NLeft cells) and NewTag, minimising the cut.
void bisect(const Graph &Gr, std::vector<I4> &Part, const std::vector<I4> &Cells, I4 Tag, I4 NewTag, size_t NLeft,
            std::vector<I4> &Mark, I4 &Stamp) {
   if (NLeft == 0) {
      for (I4 C : Cells)
         Part[C] = NewTag;
      return;
   }
   if (NLeft >= Cells.size())
      return;
   // level structure from a pseudo-peripheral cell (two sweeps), components appended one after the other
   std::vector<I4> Order, Seq;
   Seq.reserve(Cells.size());
   std::vector<char> Done; // via Mark stamps: a cell is "sequenced" when Mark == SeqStamp
   const I4 SeqStamp = ++Stamp;
   std::vector<I4> InSeq(0);
   for (I4 Start : Cells) {
      if (Mark[Start] == SeqStamp)
         continue;
      bfs(Gr, Part, Tag, Start, Order, Mark, ++Stamp);
      const I4 Far = Order.back();
      bfs(Gr, Part, Tag, Far, Order, Mark, ++Stamp);
      const I4 Far2 = Order.back();
      bfs(Gr, Part, Tag, Far2, Order, Mark, ++Stamp);
      for (I4 C : Order) {
         Seq.push_back(C);
      }
      // mark the component as sequenced (the stamps above differ from SeqStamp, so re-mark)
      for (I4 C : Order)
         Mark[C] = SeqStamp;
      // later components must not be confused by stale stamps: SeqStamp is the only "done" stamp we test
   }
   // candidate seeds: the level structure, and -- when the mesh carries cell centres -- the coordinate order along
   // the longest axis (on periodic / torus-like graphs a breadth-first front wraps around and splits into a diamond;
   // the refinement cannot undo that globally).  Each seed is refined, the lower cut wins.
   std::vector<std::vector<I4>> Seeds;
   Seeds.push_back(Seq);
   if (Gr.G.XCell && Gr.G.YCell) {
      const R8 *Cd[3] = {Gr.G.XCell, Gr.G.YCell, Gr.G.ZCell};
      int Axis = 0;
      R8 BestExt = -1;
      for (int A = 0; A < 3; ++A) {
         if (!Cd[A])
            continue;
         R8 Lo = 1e300, Hi = -1e300;
         for (I4 C : Cells)
            Lo = std::min(Lo, Cd[A][C]), Hi = std::max(Hi, Cd[A][C]);
         if (Hi - Lo > BestExt)
            BestExt = Hi - Lo, Axis = A;
      }
      std::vector<I4> ByCoord(Cells);
      const R8 *X = Cd[Axis];
      std::sort(ByCoord.begin(), ByCoord.end(), [X](I4 A, I4 B) { return X[A] < X[B] || (X[A] == X[B] && A < B); });
      Seeds.push_back(std::move(ByCoord));
   }
   auto CutOf = [&]() {
      I8 Cut = 0;
      for (I4 C : Cells)
         if (Part[C] == Tag)
            for (int J = 0; J < Gr.ME; ++J) {
               const I4 X = Gr.nbr(C, J);
               if (X >= 0 && Part[X] == NewTag)
                  ++Cut;
            }
      return Cut;
   };
   std::vector<I4> BestAssign;
   I8 BestCut = -1;
   for (const std::vector<I4> &Seed : Seeds) {
      for (size_t I = 0; I < Seed.size(); ++I)
         Part[Seed[I]] = I < NLeft ? Tag : NewTag;
      refineBisection(Gr, Part, Cells, Tag, NewTag, NLeft, Mark, Stamp);
      const I8 Cut = CutOf();
      if (BestCut < 0 || Cut < BestCut) {
         BestCut = Cut;
         BestAssign.resize(Cells.size());
         for (size_t I = 0; I < Cells.size(); ++I)
            BestAssign[I] = Part[Cells[I]];
      }
   }
   for (size_t I = 0; I < Cells.size(); ++I)
      Part[Cells[I]] = BestAssign[I];
}

void recurse(const Graph &Gr, std::vector<I4> &Part, I4 Tag, I4 NP, I4 &NextTag, std::vector<I4> &Mark, I4 &Stamp,
             std::vector<std::pair<I4, I4>> &Final, I4 Part0) {
   if (NP == 1) {
      Final.push_back({Tag, Part0});
      return;
   }
   std::vector<I4> Cells;
   for (I4 C = 0; C < Gr.N; ++C)
      if (Part[C] == Tag)
         Cells.push_back(C);
   const I4 NPLeft    = NP / 2;
   const size_t NLeft = (size_t)((double)Cells.size() * NPLeft / NP + 0.5);
   const I4 NewTag    = NextTag++;
   bisect(Gr, Part, Cells, Tag, NewTag, NLeft, Mark, Stamp);
   recurse(Gr, Part, Tag, NPLeft, NextTag, Mark, Stamp, Final, Part0);
   recurse(Gr, Part, NewTag, NP - NPLeft, NextTag, Mark, Stamp, Final, Part0 + NPLeft);
}

} // namespace

namespace {
// Recursive coordinate bisection: split `Idx[Lo,Hi)` into NP parts, part ids
// starting at Part0.  Deterministic (ties broken by global id).
void rcbSplit(const GlobalMeshDesc &G, std::vector<I4> &Idx, size_t Lo, size_t Hi, I4 Part0,
              I4 NP, std::vector<I4> &Task) {
   if (NP == 1) {
      for (size_t I = Lo; I < Hi; ++I)
         Task[Idx[I]] = Part0;
      return;
   }
   const R8 *C[3] = {G.XCell, G.YCell, G.ZCell};
   int Axis = 0;
   R8 Best  = -1;
   for (int A = 0; A < 3; ++A) {
      if (!C[A])
         continue;
      R8 Mn = 1e300, Mx = -1e300;
      for (size_t I = Lo; I < Hi; ++I) {
         R8 V = C[A][Idx[I]];
         Mn   = std::min(Mn, V);
         Mx   = std::max(Mx, V);
      }
      if (Mx - Mn > Best) {
         Best = Mx - Mn;
         Axis = A;
      }
   }
   const I4 NPLeft = NP / 2;
   const size_t N  = Hi - Lo;
   const size_t NLeft = (size_t)((double)N * NPLeft / NP + 0.5);
   const R8 *X = C[Axis];
   auto Cmp    = [X](I4 A, I4 B) { return X[A] < X[B] || (X[A] == X[B] && A < B); };
   std::nth_element(Idx.begin() + Lo, Idx.begin() + Lo + NLeft, Idx.begin() + Hi, Cmp);
   rcbSplit(G, Idx, Lo, Lo + NLeft, Part0, NPLeft, Task);
   rcbSplit(G, Idx, Lo + NLeft, Hi, Part0 + NPLeft, NP - NPLeft, Task);
}

} // namespace


void partitionRCB(const GlobalMeshDesc &G, I4 NParts, std::vector<I4> &CellTask) {
   OMEGA_REQUIRE(G.XCell && G.YCell, "Decomp: RCB partitioner needs cell coordinates");
   CellTask.assign(G.NCells, 0);
   if (NParts <= 1)
      return;
   std::vector<I4> Idx(G.NCells);
   std::iota(Idx.begin(), Idx.end(), 0);
   rcbSplit(G, Idx, 0, Idx.size(), 0, NParts, CellTask);
}

I8 edgeCut(const GlobalMeshDesc &G, const std::vector<I4> &T) {
   I8 Cut = 0;
   for (I4 C = 0; C < G.NCells; ++C)
      for (int J = 0; J < G.MaxEdges; ++J) {
         const I4 X = G.CellsOnCell[(size_t)C * G.MaxEdges + J];
         if (X > C && X < G.NCells && T[X] != T[C])
            ++Cut;
      }
   return Cut;
}

void partitionGraph(const GlobalMeshDesc &G, I4 NParts, std::vector<I4> &CellTask) {
   OMEGA_REQUIRE(NParts >= 1 && G.CellsOnCell && G.NCells > 0, "partitionGraph: bad arguments");
   const Graph Gr{G, G.MaxEdges, G.NCells};
   std::vector<I4> Part(G.NCells, 0), Mark(G.NCells, 0);
   I4 NextTag = 1, Stamp = 0;
   std::vector<std::pair<I4, I4>> Final;
   recurse(Gr, Part, 0, NParts, NextTag, Mark, Stamp, Final, 0);
   std::vector<I4> TagToPart(NextTag, 0);
   for (auto &P : Final)
      TagToPart[P.first] = P.second;
   CellTask.resize(G.NCells);
   for (I4 C = 0; C < G.NCells; ++C)
      CellTask[C] = TagToPart[Part[C]];

   // greedy k-way boundary refinement: move a boundary cell to the neighbouring part it shares most edges with when
   // that lowers the cut and keeps every part within 3 % of the mean size
   std::vector<I8> Size(NParts, 0);
   for (I4 C = 0; C < G.NCells; ++C)
      ++Size[CellTask[C]];
   const I8 MaxSize = (I8)((double)G.NCells / NParts * 1.01) + 1, MinSize = (I8)((double)G.NCells / NParts * 0.99);
   for (int Pass = 0; Pass < 4; ++Pass) {
      I8 Moved = 0;
      for (I4 C = 0; C < G.NCells; ++C) {
         const I4 Me = CellTask[C];
         int Cnt[16], NCand = 0, Own = 0;
         I4 Cand[16];
         for (int J = 0; J < Gr.ME; ++J) {
            const I4 X = Gr.nbr(C, J);
            if (X < 0)
               continue;
            const I4 T = CellTask[X];
            if (T == Me) {
               ++Own;
               continue;
            }
            int K = 0;
            while (K < NCand && Cand[K] != T)
               ++K;
            if (K == NCand && NCand < 16)
               Cand[NCand] = T, Cnt[NCand++] = 0;
            if (K < 16)
               ++Cnt[K];
         }
         int Best = -1;
         for (int K = 0; K < NCand; ++K)
            if (Cnt[K] > Own && (Best < 0 || Cnt[K] > Cnt[Best]) && Size[Cand[K]] < MaxSize && Size[Me] > MinSize)
               Best = K;
         if (Best >= 0) {
            CellTask[C] = Cand[Best];
            --Size[Me], ++Size[Cand[Best]];
            ++Moved;
         }
      }
      if (!Moved)
         break;
   }
}

} // namespace OMEGA
