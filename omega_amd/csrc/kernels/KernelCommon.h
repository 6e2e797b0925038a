// KernelCommon.h -- launch geometry and device helpers shared by the HIP kernels.
//
// Thread mapping (all element kernels): blockDim = (TX, TY).  threadIdx.x walks the
// vertical index (innermost / contiguous in memory, so a wavefront reads 64 consecutive
// level-chunks = coalesced 512 B..1 KiB per gathered row), threadIdx.y walks the
// elements of the workgroup's tile.  With K even each thread owns 2 adjacent levels and
// moves them as one 16-byte load/store (global_load_dwordx4).  The tile's connectivity
// and level-independent coefficients are staged once per workgroup into LDS; inside the
// sweep every lane of a column reads the same LDS word (broadcast, conflict-free).
//
// blockIdx -> tile is remapped so that each XCD (blocks b, b+8, b+16, ... share one
// XCD and its 4 MiB L2) walks ONE contiguous eighth of the element range: neighbouring
// elements' rows are then re-read from that XCD's L2 instead of being fetched by all 8.
#ifndef OMEGA_AMD_KERNELCOMMON_H
#define OMEGA_AMD_KERNELCOMMON_H

#include <cstdlib>
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>

#include "../Base.h"
#include "../Tuning.h"

namespace OMEGA {

typedef double dv2 __attribute__((ext_vector_type(2)));

template <class T> struct VecW;
template <> struct VecW<double> {
   static constexpr int W = 1;
};
template <> struct VecW<dv2> {
   static constexpr int W = 2;
};

/// Load / store the level-chunk Kv of row `Row` of a [rows][K] array.
template <class T> __device__ __forceinline__ T ldk(const double *P, int Row, int K, int Kv) {
   return *reinterpret_cast<const T *>(P + (size_t)Row * K + (size_t)Kv * VecW<T>::W);
}
template <class T> __device__ __forceinline__ void stk(double *P, int Row, int K, int Kv, T V) {
   *reinterpret_cast<T *>(P + (size_t)Row * K + (size_t)Kv * VecW<T>::W) = V;
}

__device__ __forceinline__ double kmax(double A, double B) { return A < B ? B : A; }
__device__ __forceinline__ dv2 kmax(dv2 A, dv2 B) {
   dv2 R;
   R.x = kmax(A.x, B.x);
   R.y = kmax(A.y, B.y);
   return R;
}
/// upwind select: U > 0 -> A, U < 0 -> B, U == 0 -> max(A, B)
__device__ __forceinline__ double upwind(double U, double A, double B) {
   return U > 0 ? A : (U < 0 ? B : kmax(A, B));
}
__device__ __forceinline__ dv2 upwind(dv2 U, dv2 A, dv2 B) {
   dv2 R;
   R.x = upwind(U.x, A.x, B.x);
   R.y = upwind(U.y, A.y, B.y);
   return R;
}
template <class T> __device__ __forceinline__ T splat(double V);
template <> __device__ __forceinline__ double splat<double>(double V) { return V; }
template <> __device__ __forceinline__ dv2 splat<dv2>(double V) {
   dv2 R;
   R.x = V;
   R.y = V;
   return R;
}

__device__ __forceinline__ double getc(double V, int) { return V; }
__device__ __forceinline__ double getc(dv2 V, int I) { return I == 0 ? V.x : V.y; }
__device__ __forceinline__ void setc(double &V, int, double X) { V = X; }
__device__ __forceinline__ void setc(dv2 &V, int I, double X) {
   if (I == 0)
      V.x = X;
   else
      V.y = X;
}

/// Bijective XCD-aware remap of a block id onto [0, N): blocks that share an XCD
/// (equal b % 8) get consecutive tiles.
__device__ __forceinline__ int xcdRemap(int B, int N) {
   const int Q = N >> 3, R = N & 7;
   const int Xcd = B & 7, Idx = B >> 3;
   return (Xcd < R ? Xcd * (Q + 1) : R * (Q + 1) + (Xcd - R) * Q) + Idx;
}

/// Bump allocator over the dynamic LDS block (8-byte granules).
struct LdsCarver {
   unsigned char *P;
   template <class U> __device__ __forceinline__ U *take(int N) {
      U *R = reinterpret_cast<U *>(P);
      P += (((size_t)N * sizeof(U) + 7) >> 3) << 3;
      return R;
   }
};
__host__ __device__ inline size_t ldsRound8(size_t B) { return ((B + 7) >> 3) << 3; }

/// Launch geometry for an N-element, K-level sweep.
struct Geom {
   dim3 Grid, Block;
   int KV;   ///< level-chunks per column (K / W)
   int Tile; ///< elements per workgroup
   int W;    ///< levels per thread (1 or 2)
   /// Tail split: the first NFull tiles are one workgroup each (all level chunks); each of the remaining tiles -- the
   /// sweep's last, partial round of workgroups -- is spread over TailSplit workgroups (one level chunk each), so
   /// that the launch drains with short workgroups instead of waiting for a few long ones.  TailSplit 1: off.
   int NFull = 0, TailSplit = 1;
};
inline Geom makeGeom(int N, int K, int MaxW = 2, int Pitch = 0, int MaxTY = 0) {
   if (Pitch <= 0)
      Pitch = K;
   Geom G;
   G.W  = (K % 2 == 0 && MaxW >= 2) ? 2 : 1;
   G.KV = K / G.W;
   // threadIdx.x spans ONE 128-byte line of a column (8 level-pairs, or 16 single levels) and
   // threadIdx.y the elements of the tile, so a workgroup issues every gather of its tile for one
   // line-deep level chunk at the same instant: rows shared between neighbouring elements of
   // the tile are then served by L1/L2 while still resident (at ~6 TB/s an XCD's 4 MiB L2
   // turns over in a few microseconds, so reuse separated by a whole sweep is lost).  The
   // workgroup walks the remaining level chunks with the x-stride loop of tileKernel.
   // Columns whose byte length is not a multiple of the line would make every 128-byte chunk straddle two
   // lines: the library's own arrays pad such rows to whole lines (Base.h: levelPitch, K = 60 -> pitch 64);
   // for short columns (K < 16) and caller-owned compact arrays threadIdx.x spans the whole column instead.
   const int LineTX   = 128 / (8 * G.W);
   const bool Aligned = (Pitch * 8) % 128 == 0; // rows start on line boundaries (levelPitch pads K >= 16 to lines)
   int TX             = Aligned ? (G.KV < LineTX ? G.KV : LineTX) : (G.KV < 64 ? G.KV : 64);
   int TY = 256 / TX;
   // bodies with a short tracer loop run a little better on half-size tiles (QU30-sized, 6 tracers: -0.6..-1 %,
   // EC30to60-sized, 2 tracers: -1.3 %); with 37 tracers the full tile is better (+3.5 % for the half tile): the
   // launchers pass MaxTY = 16 for NT <= 8
   while (MaxTY > 0 && TY > MaxTY && TY > 8)
      TY /= 2;
   // small sweeps (an eighth of the QU30-sized mesh per GPU is ~1900 tiles of 32 elements: one round of
   // workgroups on 256 CUs): smaller tiles even out the tail (measured 0.884 -> 0.860 ms at that size)
   while (TY > 8 && (N + TY - 1) / TY < 4096)
      TY /= 2;
   if (TY < 1)
      TY = 1;
   G.Block = dim3(TX, TY, 1);
   G.Tile  = TY;
   int NTiles = (N + G.Tile - 1) / G.Tile;
   // Small sweeps (QU240-sized meshes, the per-GPU share of a partitioned mesh): a workgroup walking its level
   // chunks one after the other is a chain of dependent memory round trips with too few workgroups in flight to
   // hide it, so the chunks go to separate workgroups (gridDim.y) -- more, shorter workgroups; each stages its own
   // copy of the tile's tables.
   int NChunks = (G.KV + TX - 1) / TX, Split = 1;
   while (Split < NChunks && (long)NTiles * Split < 1200) // measured: QU240-sized (882 tiles) 77 -> 64 us at 2,
      Split *= 2;                                         // 74 us at 4; an eighth of QU30 (7225 tiles) loses at any
   if (Split > NChunks)
      Split = NChunks;
   G.Grid  = dim3(NTiles > 0 ? NTiles : 1, Split > 0 ? Split : 1, 1);
   G.NFull = NTiles;
   // (wave slots: 8 waves per CU for the kernels that matter -- two per SIMD; the tail is what the last round leaves)
   const int WavesPerWG     = (TX * TY + 63) / 64;
   const int Cap            = 256 * 8 / (WavesPerWG > 0 ? WavesPerWG : 1);
   if (Split == 1 && NChunks > 1 && NTiles >= Cap) {
      const int R = NTiles % Cap;
      if (R > 0) {
         G.NFull     = NTiles - R;
         G.TailSplit = NChunks;
         G.Grid      = dim3(G.NFull + R * NChunks, 1, 1);
      }
   }
   return G;
}

/// The generic tile kernel: stage -> barrier -> column sweeps.
/// Bodies may define `static constexpr int MinWaves` (2nd __launch_bounds__ argument: minimum
/// waves per SIMD, i.e. the VGPR budget) and `static constexpr int MaxW` (levels per thread).
/// Bodies with small tables may declare `static constexpr bool HoistTables = true`: the compiler may then keep the
/// tile's LDS tables in registers across the level chunks.  For every other body a compiler-only fence at the top of
/// each chunk keeps the (loop-invariant) LDS reads inside the loop -- hoisted, the tables of the big kernels would
/// occupy more registers than the kernel has (the accessors of FusedKernels.hip use buffer instructions, whose
/// stores provably do not alias LDS, so nothing else stops the hoisting).
template <class B, class = void> struct BodyHoistTables {
   static constexpr bool V = false;
};
template <class B> struct BodyHoistTables<B, std::enable_if_t<B::HoistTables>> {
   static constexpr bool V = true;
};
template <class B> __device__ __forceinline__ void chunkFence() {
#ifndef OMEGA_NO_CHUNK_FENCE
   if constexpr (!BodyHoistTables<B>::V)
      __asm__ volatile("" ::: "memory");
#endif
}
template <class B, class = void> struct BodyOpaqueLe {
   static constexpr bool V = false;
};
template <class B> struct BodyOpaqueLe<B, std::enable_if_t<B::OpaqueLe>> {
   static constexpr bool V = true;
};
template <class B, class = void> struct BodyCooperative {
   static constexpr bool V = false;
};
template <class B> struct BodyCooperative<B, std::enable_if_t<B::Cooperative>> {
   static constexpr bool V = true;
};
template <class B, class = void> struct BodyMinWaves {
   static constexpr int V = 1;
};
template <class B> struct BodyMinWaves<B, decltype((void)B::MinWaves)> {
   static constexpr int V = B::MinWaves;
};
template <class B, class = void> struct BodyMaxW {
   static constexpr int V = 2;
};
template <class B> struct BodyMaxW<B, decltype((void)B::MaxW)> {
   static constexpr int V = B::MaxW;
};

#ifndef OMEGA_LB
#define OMEGA_LB 256
#endif
template <class Body, class T>
__global__ void __launch_bounds__(OMEGA_LB, BodyMinWaves<Body>::V)
    tileKernel(Body B, int N, int KV, int Tile, int NFull, int TailSplit) {
   extern __shared__ __align__(16) unsigned char Lds[];
   // level chunks of this workgroup: C0, C0 + CS, ...  (whole tile: gridDim.y-way split; tail tile: one chunk each)
   int TileId, C0 = blockIdx.y, CS = gridDim.y;
   if ((int)blockIdx.x < NFull) {
      TileId = xcdRemap(blockIdx.x, NFull);
   } else {
      const int Bt = blockIdx.x - NFull;
      TileId       = NFull + Bt / TailSplit;
      C0           = Bt % TailSplit;
      CS           = TailSplit;
   }
   const int First = TileId * Tile;
   int Cnt          = N - First;
   if (Cnt > Tile)
      Cnt = Tile;
   typename Body::Lds L = B.carve(Lds, Tile);
   const int Tid        = threadIdx.y * blockDim.x + threadIdx.x;
   const int NThr       = blockDim.x * blockDim.y;
   if (Cnt > 0)
      B.stage(L, First, Cnt, Tid, NThr);
   __syncthreads();
#ifdef OMEGA_STAGE_ONLY // measurement build: what the staging of the tables costs (time, HBM bytes)
   return;
#endif
   if constexpr (BodyCooperative<Body>::V) { // bodies with workgroup barriers walk the tile themselves, all threads
      B.template computeTile<T>(L, First, Cnt, C0, CS, KV);
   } else {
      for (int Le = threadIdx.y; Le < Cnt; Le += blockDim.y)
         for (int Kv = C0 * blockDim.x + threadIdx.x; Kv < KV; Kv += blockDim.x * CS) {
            chunkFence<Body>();
            if constexpr (BodyOpaqueLe<Body>::V) {
               // the tile-local element index made opaque per chunk: the LDS table addresses derived from it are
               // recomputed (a few VALU adds) instead of living in VGPRs across the chunks -- for a body at the
               // register limit (profiles/r04_opaque_le.json)
               int LeO = Le;
               __asm__ volatile("" : "+v"(LeO));
               B.template compute<T>(L, LeO, First + Le, Kv);
            } else
               B.template compute<T>(L, Le, First + Le, Kv);
         }
   }
}

/// Two INDEPENDENT sweeps in one launch: the first NTilesA workgroups run body A over its tiles, the others body B.
/// Saves a launch (what small sweeps are made of) and lets the tail of one sweep overlap the ramp of the other.
/// Each half keeps its own XCD mapping.  Register budget and LDS are the larger of the two bodies'.
template <class BA, class BB, class T>
__global__ void __launch_bounds__(OMEGA_LB, (BodyMinWaves<BA>::V < BodyMinWaves<BB>::V ? BodyMinWaves<BA>::V
                                                                                       : BodyMinWaves<BB>::V))
    tileKernel2(BA A, BB Bb, int NA, int NB, int KV, int Tile, int NTilesA, int NFullB, int TailSplit) {
   extern __shared__ __align__(16) unsigned char Lds[];
   const int Tid  = threadIdx.y * blockDim.x + threadIdx.x;
   const int NThr = blockDim.x * blockDim.y;
   if ((int)blockIdx.x < NTilesA) {
      const int Ta    = xcdRemap(blockIdx.x, NTilesA);
      const int First = Ta * Tile;
      const int Cnt   = NA - First < Tile ? NA - First : Tile;
      typename BA::Lds L = A.carve(Lds, Tile);
      if (Cnt > 0)
         A.stage(L, First, Cnt, Tid, NThr);
      __syncthreads();
#ifdef OMEGA_STAGE_ONLY
      return;
#endif
      for (int Le = threadIdx.y; Le < Cnt; Le += blockDim.y)
         for (int Kv = blockIdx.y * blockDim.x + threadIdx.x; Kv < KV; Kv += blockDim.x * gridDim.y)
         {
            chunkFence<BA>();
            A.template compute<T>(L, Le, First + Le, Kv);
         }
   } else {
      // (the second body's last tiles are the launch's tail: one level chunk per workgroup, see Geom::TailSplit)
      const int Bb_ = blockIdx.x - NTilesA;
      int TileId, C0 = blockIdx.y, CS = gridDim.y;
      if (Bb_ < NFullB) {
         TileId = xcdRemap(Bb_, NFullB);
      } else {
         TileId = NFullB + (Bb_ - NFullB) / TailSplit;
         C0     = (Bb_ - NFullB) % TailSplit;
         CS     = TailSplit;
      }
      const int First = TileId * Tile;
      const int Cnt   = NB - First < Tile ? NB - First : Tile;
      typename BB::Lds L = Bb.carve(Lds, Tile);
      if (Cnt > 0)
         Bb.stage(L, First, Cnt, Tid, NThr);
      __syncthreads();
#ifdef OMEGA_STAGE_ONLY
      return;
#endif
      for (int Le = threadIdx.y; Le < Cnt; Le += blockDim.y)
         for (int Kv = C0 * blockDim.x + threadIdx.x; Kv < KV; Kv += blockDim.x * CS)
         {
            chunkFence<BB>();
            Bb.template compute<T>(L, Le, First + Le, Kv);
         }
   }
}

template <class B, class = void> struct BodyHasNT {
   static constexpr bool V = false;
};
template <class B> struct BodyHasNT<B, decltype((void)std::declval<B &>().NT)> {
   static constexpr bool V = true;
};
template <class B> inline int bodyMaxTY(const B &Body) {
   if constexpr (BodyHasNT<B>::V)
      return Body.NT <= 8 ? 16 : 0;
   else
      return 0;
}
template <class B, class = void> struct BodyHasKLog {
   static constexpr bool V = false;
};
template <class B> struct BodyHasKLog<B, decltype((void)std::declval<B &>().KLog)> {
   static constexpr bool V = true;
};
/// Bodies that keep per-wavefront areas in LDS (the prefetch slots of CellPVFinalTracerBody) declare `int NWv`: the
/// launchers set it to the wavefronts per workgroup before asking for ldsBytes().
template <class B, class = void> struct BodyHasNWv {
   static constexpr bool V = false;
};
template <class B> struct BodyHasNWv<B, decltype((void)std::declval<B &>().NWv)> {
   static constexpr bool V = true;
};
template <class B> inline void setWaves(B &Body, const Geom &G) {
   if constexpr (BodyHasNWv<B>::V)
      Body.NWv = (int)(G.Block.x * G.Block.y + 63) / 64;
}
/// Sweep of elements [0, N) x K levels.  Every body addresses rows through its member `K`, which this launcher
/// sets to the row pitch of the arrays (levelPitch(K) for the library's own arrays; Pitch >= K for caller-owned
/// ones, e.g. compact raw arrays of the C ABI); bodies that also need the level COUNT (bottom level of the drag
/// term) declare `int KLog`.
template <class Body> void launchTile(const Body &B0, int N, int K, hipStream_t S, int Pitch = -1) {
   if (N <= 0)
      return;
   Body B = B0;
   B.K    = Pitch > 0 ? Pitch : levelPitch(K);
   if constexpr (BodyHasKLog<Body>::V)
      B.KLog = K;
   Geom G           = makeGeom(N, K, BodyMaxW<Body>::V, B.K, bodyMaxTY(B));
   setWaves(B, G);
   const size_t Lds = B.ldsBytes(G.Tile);
   if constexpr (BodyMaxW<Body>::V >= 2) {
      if (G.W == 2) {
         hipLaunchKernelGGL((tileKernel<Body, dv2>), G.Grid, G.Block, Lds, S, B, N, G.KV, G.Tile, G.NFull, G.TailSplit);
         HIP_CHECK(hipGetLastError());
         return;
      }
   }
   hipLaunchKernelGGL((tileKernel<Body, double>), G.Grid, G.Block, Lds, S, B, N, G.KV, G.Tile, G.NFull, G.TailSplit);
   HIP_CHECK(hipGetLastError());
}

/// launchTile for two independent sweeps of the same level count (see tileKernel2); either may be empty
template <class BA, class BB> void launchTile2(const BA &A0, int NA, const BB &B0, int NB, int K, hipStream_t S) {
   static_assert(BodyMaxW<BA>::V == BodyMaxW<BB>::V, "launchTile2: bodies must agree on the levels per thread");
   if (NA <= 0) {
      launchTile(B0, NB, K, S);
      return;
   }
   if (NB <= 0) {
      launchTile(A0, NA, K, S);
      return;
   }
   BA A  = A0;
   BB Bb = B0;
   A.K = Bb.K = levelPitch(K);
   if constexpr (BodyHasKLog<BA>::V)
      A.KLog = K;
   if constexpr (BodyHasKLog<BB>::V)
      Bb.KLog = K;
   const int TyA = bodyMaxTY(A), TyB = bodyMaxTY(Bb);
   Geom G        = makeGeom(NA + NB, K, BodyMaxW<BA>::V, A.K, TyA > TyB ? TyA : TyB);
   const int NTA = (NA + G.Tile - 1) / G.Tile, NTB = (NB + G.Tile - 1) / G.Tile;
   setWaves(A, G), setWaves(Bb, G);
   const size_t LA = A.ldsBytes(G.Tile), LB = Bb.ldsBytes(G.Tile), Lds = LA > LB ? LA : LB;
   // tail split on the second body's last tiles (the combined sweep's last, partial round of workgroups)
   int NFullB = NTB, TailSplit = 1;
   dim3 Grid(NTA + NTB, G.TailSplit > 1 ? 1 : G.Grid.y, 1);
   if (G.TailSplit > 1) {
      const int R = (NTA + NTB) - G.NFull; // makeGeom(NA + NB): tiles of the partial round (tile counts differ by <= 1)
      if (R > 0 && R <= NTB) {
         NFullB    = NTB - R;
         TailSplit = G.TailSplit;
         Grid      = dim3(NTA + NFullB + R * TailSplit, 1, 1);
      }
   }
   if constexpr (BodyMaxW<BA>::V >= 2) {
      if (G.W == 2) {
         hipLaunchKernelGGL((tileKernel2<BA, BB, dv2>), Grid, G.Block, Lds, S, A, Bb, NA, NB, G.KV, G.Tile, NTA, NFullB,
                            TailSplit);
         HIP_CHECK(hipGetLastError());
         return;
      }
   }
   hipLaunchKernelGGL((tileKernel2<BA, BB, double>), Grid, G.Block, Lds, S, A, Bb, NA, NB, G.KV, G.Tile, NTA, NFullB,
                      TailSplit);
   HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// Any number of INDEPENDENT sweeps in one launch (the generalisation of tileKernel2): body I owns the workgroups
// [TileStart[I], TileStart[I+1]).  Used where a mesh needs list launches next to a full sweep -- the heptagons of a
// hexagon mesh on the wide tables, the pentagons' ring instantiations -- so that a few hundred scattered cells ride
// along with the big sweep instead of costing a launch (and an idle chip) each.  No tail split: the short list
// workgroups at the end of the launch ARE the tail.  The bodies travel as kernel arguments (4 KiB in total).
constexpr int MaxSweeps = 4;
struct SweepPlan {
   int N[MaxSweeps];             ///< elements of each sweep
   int TileStart[MaxSweeps + 1]; ///< first workgroup of each sweep; [NBodies] = grid size
};
template <int I, class T, class B, class... Rest>
__device__ __forceinline__ void runSweep(const SweepPlan &Pl, int KV, int Tile, unsigned char *Lds, const B &Body,
                                         const Rest &...More) {
   if ((int)blockIdx.x < Pl.TileStart[I + 1]) {
      const int NTiles = Pl.TileStart[I + 1] - Pl.TileStart[I];
      const int Tl     = xcdRemap(blockIdx.x - Pl.TileStart[I], NTiles);
      const int First  = Tl * Tile;
      const int Cnt    = Pl.N[I] - First < Tile ? Pl.N[I] - First : Tile;
      typename B::Lds L = Body.carve(Lds, Tile);
      const int Tid     = threadIdx.y * blockDim.x + threadIdx.x;
      if (Cnt > 0)
         Body.stage(L, First, Cnt, Tid, blockDim.x * blockDim.y);
      __syncthreads();
      if constexpr (BodyCooperative<B>::V) {
         Body.template computeTile<T>(L, First, Cnt, blockIdx.y, gridDim.y, KV);
      } else {
         for (int Le = threadIdx.y; Le < Cnt; Le += blockDim.y)
            for (int Kv = blockIdx.y * blockDim.x + threadIdx.x; Kv < KV; Kv += blockDim.x * gridDim.y) {
               chunkFence<B>();
               Body.template compute<T>(L, Le, First + Le, Kv);
            }
      }
      return;
   }
   if constexpr (sizeof...(Rest) > 0)
      runSweep<I + 1, T>(Pl, KV, Tile, Lds, More...);
}
template <class... Bs> struct MinWavesOf;
template <class B> struct MinWavesOf<B> {
   static constexpr int V = BodyMinWaves<B>::V;
};
template <class B, class... Bs> struct MinWavesOf<B, Bs...> {
   static constexpr int V = BodyMinWaves<B>::V < MinWavesOf<Bs...>::V ? BodyMinWaves<B>::V : MinWavesOf<Bs...>::V;
};
template <class T, class... Bs>
__global__ void __launch_bounds__(OMEGA_LB, MinWavesOf<Bs...>::V) tileKernelV(SweepPlan Pl, int KV, int Tile, Bs... Bodies) {
   extern __shared__ __align__(16) unsigned char Lds[];
   runSweep<0, T>(Pl, KV, Tile, Lds, Bodies...);
}

/// launchTileV(K, S, BodyA, NA, BodyB, NB, ...): the sweeps BodyX over [0, NX) in one launch (sweeps with N <= 0 get no
/// workgroups).  All bodies must agree on the levels per thread.
namespace detail {
template <class B> inline void prepBody(B &Body, int K) {
   Body.K = levelPitch(K);
   if constexpr (BodyHasKLog<B>::V)
      Body.KLog = K;
}
} // namespace detail
template <class B0, class B1> void launchTileV(int K, hipStream_t S, const B0 &A0, int N0, const B1 &A1, int N1) {
   static_assert(BodyMaxW<B0>::V == BodyMaxW<B1>::V, "launchTileV: bodies must agree on the levels per thread");
   static_assert(sizeof(B0) + sizeof(B1) + sizeof(SweepPlan) <= 3900, "launchTileV: kernel arguments exceed 4 KiB");
   B0 X0 = A0;
   B1 X1 = A1;
   detail::prepBody(X0, K), detail::prepBody(X1, K);
   const int Ns[2] = {N0 > 0 ? N0 : 0, N1 > 0 ? N1 : 0};
   const int Ty0 = bodyMaxTY(X0), Ty1 = bodyMaxTY(X1);
   Geom G = makeGeom(Ns[0] + Ns[1], K, BodyMaxW<B0>::V, X0.K, Ty0 > Ty1 ? Ty0 : Ty1);
   SweepPlan Pl{};
   Pl.TileStart[0] = 0;
   for (int I = 0; I < 2; ++I) {
      Pl.N[I]             = Ns[I];
      Pl.TileStart[I + 1] = Pl.TileStart[I] + (Ns[I] + G.Tile - 1) / G.Tile;
   }
   if (Pl.TileStart[2] == 0)
      return;
   setWaves(X0, G), setWaves(X1, G);
   const size_t L0 = X0.ldsBytes(G.Tile), L1 = X1.ldsBytes(G.Tile), Lds = L0 > L1 ? L0 : L1;
   const dim3 Grid(Pl.TileStart[2], G.TailSplit > 1 ? 1 : G.Grid.y, 1);
   if constexpr (BodyMaxW<B0>::V >= 2) {
      if (G.W == 2) {
         hipLaunchKernelGGL((tileKernelV<dv2, B0, B1>), Grid, G.Block, Lds, S, Pl, G.KV, G.Tile, X0, X1);
         HIP_CHECK(hipGetLastError());
         return;
      }
   }
   hipLaunchKernelGGL((tileKernelV<double, B0, B1>), Grid, G.Block, Lds, S, Pl, G.KV, G.Tile, X0, X1);
   HIP_CHECK(hipGetLastError());
}
template <class B0, class B1, class B2>
void launchTileV(int K, hipStream_t S, const B0 &A0, int N0, const B1 &A1, int N1, const B2 &A2, int N2) {
   static_assert(BodyMaxW<B0>::V == BodyMaxW<B1>::V && BodyMaxW<B0>::V == BodyMaxW<B2>::V,
                 "launchTileV: bodies must agree on the levels per thread");
   static_assert(sizeof(B0) + sizeof(B1) + sizeof(B2) + sizeof(SweepPlan) <= 3900, "launchTileV: kernel arguments exceed 4 KiB");
   B0 X0 = A0;
   B1 X1 = A1;
   B2 X2 = A2;
   detail::prepBody(X0, K), detail::prepBody(X1, K), detail::prepBody(X2, K);
   const int Ns[3] = {N0 > 0 ? N0 : 0, N1 > 0 ? N1 : 0, N2 > 0 ? N2 : 0};
   int Ty = bodyMaxTY(X0);
   Ty     = bodyMaxTY(X1) > Ty ? bodyMaxTY(X1) : Ty;
   Ty     = bodyMaxTY(X2) > Ty ? bodyMaxTY(X2) : Ty;
   Geom G = makeGeom(Ns[0] + Ns[1] + Ns[2], K, BodyMaxW<B0>::V, X0.K, Ty);
   SweepPlan Pl{};
   Pl.TileStart[0] = 0;
   for (int I = 0; I < 3; ++I) {
      Pl.N[I]             = Ns[I];
      Pl.TileStart[I + 1] = Pl.TileStart[I] + (Ns[I] + G.Tile - 1) / G.Tile;
   }
   if (Pl.TileStart[3] == 0)
      return;
   setWaves(X0, G), setWaves(X1, G), setWaves(X2, G);
   size_t Lds = X0.ldsBytes(G.Tile);
   Lds        = X1.ldsBytes(G.Tile) > Lds ? X1.ldsBytes(G.Tile) : Lds;
   Lds        = X2.ldsBytes(G.Tile) > Lds ? X2.ldsBytes(G.Tile) : Lds;
   const dim3 Grid(Pl.TileStart[3], G.TailSplit > 1 ? 1 : G.Grid.y, 1);
   if constexpr (BodyMaxW<B0>::V >= 2) {
      if (G.W == 2) {
         hipLaunchKernelGGL((tileKernelV<dv2, B0, B1, B2>), Grid, G.Block, Lds, S, Pl, G.KV, G.Tile, X0, X1, X2);
         HIP_CHECK(hipGetLastError());
         return;
      }
   }
   hipLaunchKernelGGL((tileKernelV<double, B0, B1, B2>), Grid, G.Block, Lds, S, Pl, G.KV, G.Tile, X0, X1, X2);
   HIP_CHECK(hipGetLastError());
}

} // namespace OMEGA
#endif
