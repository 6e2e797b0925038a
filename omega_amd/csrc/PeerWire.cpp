// PeerWire.cpp -- see PeerWire.h.
#include "PeerWire.h"
#include "Halo.h"

#include <cstring>

namespace OMEGA {

namespace {
struct HandleBlock {
   hipIpcMemHandle_t Mailbox, Flags;
   unsigned long long MailboxBytes;
   int Rank, NRanks;
};
static_assert(sizeof(HandleBlock) <= (size_t)PeerWire::HandleBytes, "PeerWire::HandleBytes too small");
constexpr double TicksPerSecond = 1.0e8; // wall_clock64(): constant 100 MHz counter on gfx9
} // namespace

PeerWire::PeerWire(int NRanks_, int Rank_, size_t MailboxBytes_)
    : NRanks(NRanks_), Rank(Rank_), MailboxBytes((MailboxBytes_ + 255) / 256 * 256) {
   OMEGA_REQUIRE(NRanks >= 1 && Rank >= 0 && Rank < NRanks, "PeerWire: bad rank / size");
   OMEGA_REQUIRE(MailboxBytes > 0, "PeerWire: empty mailbox");
   // uncached (fine-grained) device memory: written by peers through the fabric, read by local kernels
   HIP_CHECK(hipExtMallocWithFlags(&Mailbox, MailboxBytes, hipDeviceMallocUncached));
   void *F = nullptr;
   HIP_CHECK(hipExtMallocWithFlags(&F, flagWords() * sizeof(unsigned long long), hipDeviceMallocUncached));
   noteDeviceResource(2);
   Flags = static_cast<unsigned long long *>(F);
   HIP_CHECK(hipMemset(Mailbox, 0, MailboxBytes));
   HIP_CHECK(hipMemset(Flags, 0, flagWords() * sizeof(unsigned long long)));
   HIP_CHECK(hipStreamSynchronize(nullptr)); // the fills have run before any peer can see these buffers (Device.cpp)
   void *St = nullptr;
   HIP_CHECK(hipHostMalloc(&St, sizeof(int), hipHostMallocMapped));
   Status  = static_cast<int *>(St);
   *Status = 0;
   HIP_CHECK(hipDeviceSynchronize());
   setTimeout(60.0);
   PeerMailbox.assign(NRanks, nullptr);
   PeerFlags.assign(NRanks, nullptr);
   PeerMailboxBytes.assign(NRanks, 0);
}

PeerWire::~PeerWire() {
   if (BoundTo)
      BoundTo->peerWireGone(this);
   BoundTo = nullptr;
   (void)hipDeviceSynchronize();
   for (int R = 0; R < NRanks; ++R) {
      if (PeerMailbox[R])
         (void)hipIpcCloseMemHandle(PeerMailbox[R]);
      if (PeerFlags[R])
         (void)hipIpcCloseMemHandle(PeerFlags[R]);
   }
   if (Mailbox)
      (void)hipFree(Mailbox);
   if (Flags)
      (void)hipFree(Flags);
   if (Status)
      (void)hipHostFree(Status);
}

void PeerWire::setTimeout(double Seconds) {
   OMEGA_REQUIRE(Seconds > 0 && Seconds <= 600, "PeerWire: timeout out of range");
   TimeoutTicks = (long long)(Seconds * TicksPerSecond);
}

void PeerWire::localHandle(char Out[HandleBytes]) const {
   HandleBlock B;
   std::memset(&B, 0, sizeof(B));
   HIP_CHECK(hipIpcGetMemHandle(&B.Mailbox, Mailbox));
   HIP_CHECK(hipIpcGetMemHandle(&B.Flags, Flags));
   B.MailboxBytes = MailboxBytes;
   B.Rank = Rank, B.NRanks = NRanks;
   std::memset(Out, 0, HandleBytes);
   std::memcpy(Out, &B, sizeof(B));
}

void PeerWire::connect(const char *All) {
   OMEGA_REQUIRE(!Connected, "PeerWire: already connected");
   for (int R = 0; R < NRanks; ++R) {
      HandleBlock B;
      std::memcpy(&B, All + (size_t)R * HandleBytes, sizeof(B));
      OMEGA_REQUIRE(B.Rank == R && B.NRanks == NRanks, "PeerWire: handle blocks are not in rank order");
      PeerMailboxBytes[R] = B.MailboxBytes;
      if (R == Rank)
         continue;
      HIP_CHECK(hipIpcOpenMemHandle(&PeerMailbox[R], B.Mailbox, hipIpcMemLazyEnablePeerAccess));
      void *F = nullptr;
      HIP_CHECK(hipIpcOpenMemHandle(&F, B.Flags, hipIpcMemLazyEnablePeerAccess));
      PeerFlags[R] = static_cast<unsigned long long *>(F);
   }
   Connected = true;
}

int PeerWire::status() const { return *static_cast<volatile int *>(Status); }

int PeerWire::put(int N, const int *Peers, void *const *SendPtrs, const size_t *SendBytes, const size_t *RemoteOff,
                  hipStream_t S) {
   auto Fail = [&](const std::string &Msg) {
      LastError = Msg;
      return 1;
   };
   if (!Connected)
      return Fail("PeerWire: not connected");
   if (N < 0 || N > MaxPeers)
      return Fail("PeerWire: too many neighbours in one exchange");
   if (int St = status())
      return Fail("PeerWire: an earlier exchange timed out waiting for a peer (status " + std::to_string(St) + ")");
   PeerFlagIdx ConsumedIdx{}, ArrivedIdx{};
   PeerFlagPtrs ArrivedAt{};
   for (int I = 0; I < N; ++I) {
      const int P = Peers[I];
      if (P < 0 || P >= NRanks || P == Rank || !PeerMailbox[P])
         return Fail("PeerWire: bad peer rank");
      if (RemoteOff[I] + SendBytes[I] > PeerMailboxBytes[P])
         return Fail("PeerWire: message does not fit the peer's mailbox (" + std::to_string(RemoteOff[I] + SendBytes[I]) +
                     " > " + std::to_string(PeerMailboxBytes[P]) + " bytes): create the wire with a larger mailbox");
      ConsumedIdx.I[I] = NRanks + P;
      ArrivedIdx.I[I]  = P;
      ArrivedAt.P[I]   = PeerFlags[P] + Rank; // arrived[me] in the peer's block
   }
   ConsumedIdx.N = ArrivedIdx.N = ArrivedAt.N = N;
   const unsigned long long Seq = (unsigned long long)(NExchanges + 1);
   // my previous message must have been read before I overwrite the peer's mailbox
   if (Seq > 1)
      launchPeerWait(Flags, ConsumedIdx, Seq - 1, Status, 1, TimeoutTicks, S);
   for (int I = 0; I < N; ++I)
      if (SendBytes[I])
         HIP_CHECK(hipMemcpyAsync(static_cast<char *>(PeerMailbox[Peers[I]]) + RemoteOff[I], SendPtrs[I], SendBytes[I],
                                  hipMemcpyDeviceToDevice, S));
   launchPeerSignalWait(ArrivedAt, Seq, Flags, ArrivedIdx, true, Status, 2, TimeoutTicks, S);
   ++NExchanges;
   return 0;
}

int PeerWire::allGather(const void *In, int NVals, void *Out, hipStream_t S) {
   auto Fail = [&](const std::string &Msg) {
      LastError = Msg;
      return 1;
   };
   if (!Connected)
      return Fail("PeerWire: not connected");
   if (NRanks > MaxRanksGather || NVals < 1 || NVals > MaxGatherVals)
      return Fail("PeerWire::allGather: at most " + std::to_string(MaxRanksGather) + " ranks and " +
                  std::to_string(MaxGatherVals) + " values per rank");
   if (int St = status())
      return Fail("PeerWire: an earlier exchange timed out waiting for a peer (status " + std::to_string(St) + ")");
   PeerBlockPtrs B{};
   for (int R = 0; R < NRanks; ++R)
      B.P[R] = R == Rank ? Flags : PeerFlags[R];
   launchPeerAllGather(B, NRanks, Rank, static_cast<const unsigned long long *>(In), NVals,
                       static_cast<unsigned long long *>(Out), (unsigned long long)(NGathers + 1), Status, TimeoutTicks, S);
   ++NGathers;
   return 0;
}

int PeerWire::release(int N, const int *Peers, hipStream_t S) {
   if (N < 0 || N > MaxPeers) {
      LastError = "PeerWire: too many neighbours in one exchange";
      return 1;
   }
   PeerFlagPtrs ConsumedAt{};
   for (int I = 0; I < N; ++I) {
      const int P = Peers[I];
      if (!Connected || P < 0 || P >= NRanks || P == Rank || !PeerFlags[P]) {
         LastError = "PeerWire: bad peer rank";
         return 1;
      }
      ConsumedAt.P[I] = PeerFlags[P] + NRanks + Rank; // consumed[me] in the peer's block
   }
   ConsumedAt.N = N;
   PeerFlagIdx None{};
   launchPeerSignalWait(ConsumedAt, (unsigned long long)NExchanges, Flags, None, false, Status, 0, TimeoutTicks, S);
   return 0;
}

} // namespace OMEGA
