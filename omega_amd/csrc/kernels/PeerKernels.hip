// PeerKernels.hip -- the one-wavefront signal / wait kernels of the peer-copy halo wire (PeerWire.h).
// Lane i serves neighbour i.  Stores into a peer's flag block are system-scope release atomics (everything this
// stream did before -- the copies into that peer's mailbox -- is visible before the flag); loads of the local flags are
// system-scope acquire atomics on uncached memory.  A wait sleeps between polls and gives up after TimeoutTicks of
// the constant 100 MHz counter, so every wave reaches its exit whatever the peers do.
#include "../PeerWire.h"

namespace OMEGA {

__device__ static inline void peerWaitLane(const unsigned long long *Flag, unsigned long long Seq, int *Status, int Bit,
                                           long long TimeoutTicks) {
   const long long T0 = wall_clock64();
   while (__hip_atomic_load(Flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < Seq) {
      if (wall_clock64() - T0 > TimeoutTicks) {
         atomicOr_system(Status, Bit);
         break;
      }
      __builtin_amdgcn_s_sleep(32);
   }
}

__global__ void __launch_bounds__(64)
peerWaitKernel(const unsigned long long *Local, PeerFlagIdx Idx, unsigned long long Seq, int *Status, int Bit,
               long long TimeoutTicks) {
   const int I = threadIdx.x;
   if (I < Idx.N)
      peerWaitLane(Local + Idx.I[I], Seq, Status, Bit, TimeoutTicks);
}

__global__ void __launch_bounds__(64)
peerSignalWaitKernel(PeerFlagPtrs Remote, unsigned long long Seq, const unsigned long long *Local, PeerFlagIdx Idx,
                     int Wait, int *Status, int Bit, long long TimeoutTicks) {
   const int I = threadIdx.x;
   // the "consumed" signal (Wait == 0) of an exchange whose message never arrived is not sent: nothing was consumed
   if (!Wait && __hip_atomic_load(Status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)
      return;
   if (I < Remote.N)
      __hip_atomic_store(Remote.P[I], Seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   if (Wait && I < Idx.N)
      peerWaitLane(Local + Idx.I[I], Seq, Status, Bit, TimeoutTicks);
}

// All-gather (PeerWire::allGather).  Flag block layout: [arrived NRanks][consumed NRanks][gatherSeq NRanks]
// [parity 2][from NRanks][MaxGatherVals].  Lane r: my values into rank r's slots [Seq & 1][me], release-store Seq into
// its gatherSeq[me]; then wait for gatherSeq[r] >= Seq locally and copy rank r's values out.
__global__ void __launch_bounds__(64)
peerAllGatherKernel(PeerBlockPtrs Blocks, int NRanks, int Rank, const unsigned long long *In, int NVals,
                    unsigned long long *Out, unsigned long long Seq, int *Status, long long TimeoutTicks) {
   constexpr int MV      = PeerWire::MaxGatherVals;
   const int R           = threadIdx.x;
   const size_t SeqOff   = 2 * (size_t)NRanks;
   const size_t SlotBase = 3 * (size_t)NRanks + (size_t)(Seq & 1) * NRanks * MV;
   if (R >= NRanks)
      return;
   if (R == Rank) {
      for (int V = 0; V < NVals; ++V)
         Out[(size_t)Rank * NVals + V] = In[V];
      return;
   }
   unsigned long long *Dst = Blocks.P[R] + SlotBase + (size_t)Rank * MV;
   for (int V = 0; V < NVals; ++V)
      __hip_atomic_store(Dst + V, In[V], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
   __hip_atomic_store(Blocks.P[R] + SeqOff + Rank, Seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   const unsigned long long *Local = Blocks.P[Rank];
   peerWaitLane(Local + SeqOff + R, Seq, Status, 4, TimeoutTicks);
   const unsigned long long *Src = Local + SlotBase + (size_t)R * MV;
   for (int V = 0; V < NVals; ++V)
      Out[(size_t)R * NVals + V] = __hip_atomic_load(Src + V, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launchPeerAllGather(const PeerBlockPtrs &Blocks, int NRanks, int Rank, const unsigned long long *In, int NVals,
                         unsigned long long *Out, unsigned long long Seq, int *Status, long long TimeoutTicks,
                         hipStream_t S) {
   hipLaunchKernelGGL(peerAllGatherKernel, dim3(1), dim3(64), 0, S, Blocks, NRanks, Rank, In, NVals, Out, Seq, Status,
                      TimeoutTicks);
   HIP_CHECK(hipGetLastError());
}

void launchPeerWait(const unsigned long long *Local, const PeerFlagIdx &Idx, unsigned long long Seq, int *Status, int Bit,
                    long long TimeoutTicks, hipStream_t S) {
   if (Idx.N == 0)
      return;
   hipLaunchKernelGGL(peerWaitKernel, dim3(1), dim3(64), 0, S, Local, Idx, Seq, Status, Bit, TimeoutTicks);
   HIP_CHECK(hipGetLastError());
}

void launchPeerSignalWait(const PeerFlagPtrs &Remote, unsigned long long Seq, const unsigned long long *Local,
                          const PeerFlagIdx &Idx, bool Wait, int *Status, int Bit, long long TimeoutTicks, hipStream_t S) {
   if (Remote.N == 0 && (!Wait || Idx.N == 0))
      return;
   hipLaunchKernelGGL(peerSignalWaitKernel, dim3(1), dim3(64), 0, S, Remote, Seq, Local, Idx, Wait ? 1 : 0, Status, Bit,
                      TimeoutTicks);
   HIP_CHECK(hipGetLastError());
}

} // namespace OMEGA
