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
   if (I < Remote.N)
      __hip_atomic_store(Remote.P[I], Seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   if (Wait && I < Idx.N)
      peerWaitLane(Local + Idx.I[I], Seq, Status, Bit, TimeoutTicks);
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
