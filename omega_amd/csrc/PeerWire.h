// PeerWire.h -- the second halo wire: direct peer copies between the GPUs of one node (xGMI), stream-ordered end to
// end, no host synchronisation and no library between the ranks.
//
// Reference analogue: the MPI calls inside Halo (components/omega/src/base/Halo.h:851-907 startReceives /
// startSends + the MPI_Test polling loop).  There every exchange fences the device and the HOST polls for
// completion.  Here nothing ever waits on the host:
//
//   * every rank owns a MAILBOX in device memory (its receive buffer) and a small block of FLAGS, both exported once
//     with hipIpcGetMemHandle and opened by the peers (one process per GPU; on a one-GPU test box all ranks' memory
//     is on the same device, which is what lets the multi-rank tests exercise this wire there);
//   * exchange number s on stream S:  pack kernel -> [wait until every neighbour has consumed my message s-1]
//     -> one hipMemcpyAsync(device to device) per neighbour straight into ITS mailbox at the offset its unpack
//     kernel expects (both sides derive the layout from the mesh, Halo.cpp) -> [set "arrived = s" in every
//     neighbour's flags, wait until all my neighbours' messages s have arrived] -> unpack kernel ->
//     [set "consumed = s" in every neighbour's flags];
//   * the bracketed steps are one-wavefront kernels on S: system-scope release stores into the peer's flag block,
//     acquire loads (with s_sleep) on the local one.  A wait gives up after a fixed time (a peer that died must
//     not park a wave forever), raises a sticky status the host sees at the next call, and lets the stream drain.
//
// Mailbox and flags are allocated uncached (fine grained): a peer's writes arrive through the fabric, not through this
// GPU's L2, and the unpack kernel must not find stale lines there.
#ifndef OMEGA_AMD_PEERWIRE_H
#define OMEGA_AMD_PEERWIRE_H

#include "Base.h"

namespace OMEGA {

class PeerWire {
 public:
   static constexpr int MaxPeers    = 32;  ///< neighbours of one rank in one exchange
   static constexpr int MaxRanksGather = 64;  ///< ranks an allGather spans (one lane each)
   static constexpr int MaxGatherVals  = 128; ///< 8-byte values per rank in one allGather (64 double-double pairs)
   static constexpr int HandleBytes = 160; ///< two hipIpcMemHandle_t (64 B each) + sizes

   /// The calling process must already have selected its GPU (deviceInit).  MailboxBytes: capacity of this rank's
   /// receive buffer (the largest exchange it will take part in: Halo::recvRows * K * 8).
   PeerWire(int NRanks, int Rank, size_t MailboxBytes);
   ~PeerWire();
   PeerWire(const PeerWire &)            = delete;
   PeerWire &operator=(const PeerWire &) = delete;

   int NRanks, Rank;
   I8 NExchanges = 0;

   /// this rank's handle block, to be distributed to every rank by any side channel (once)
   void localHandle(char Out[HandleBytes]) const;
   /// All = NRanks * HandleBytes bytes in rank order.  Opens the peers' mailboxes and flag blocks.  Collective in
   /// the sense that every rank must call it before the first exchange; it does not communicate.
   void connect(const char *All);
   bool connected() const { return Connected; }
   /// One wire serves ONE Halo (the exchange counter and the mailbox layout are that Halo's).  The binding is kept from
   /// both ends (Halo::usePeerWire sets it): whichever of the two objects is destroyed first releases the other, so
   /// neither a Halo that outlives its wire nor a wire that outlives its Halo is left with a dangling pointer.  A Halo
   /// whose wire is gone has no wire (its next exchange fails with "no transport"), it does not crash.
   class Halo *BoundTo = nullptr;
   bool bound() const { return BoundTo != nullptr; }

   void *mailbox() const { return Mailbox; }
   size_t mailboxBytes() const { return MailboxBytes; }

   /// Steps 2-4 of an exchange (see above) on stream S: message I goes from SendPtrs[I] (SendBytes[I] bytes, local
   /// device memory, complete in stream order) to Peers[I]'s mailbox at byte offset RemoteOff[I]; returns when the
   /// work is QUEUED.  After it, in stream order, all neighbours' messages are in the local mailbox.
   int put(int N, const int *Peers, void *const *SendPtrs, const size_t *SendBytes, const size_t *RemoteOff,
           hipStream_t S);
   /// Step 6: the mailbox has been read (unpack kernel queued on S before this call): tell the neighbours.
   int release(int N, const int *Peers, hipStream_t S);

   /// All-gather of NVals (<= MaxGatherVals) 8-byte values per rank over ALL ranks of the wire (not only the halo
   /// neighbours): In[NVals] -> Out[NRanks][NVals], both local device memory, on stream S; one one-wavefront kernel, lane
   /// r writes this rank's values into rank r's gather slots (system-scope stores, then a release store of the gather's
   /// sequence number), waits for rank r's sequence number in the local block and copies its values out.  Slots are
   /// double-buffered by the parity of the sequence number: a rank can only be one gather ahead of any other (it needs
   /// everybody's values to finish one).  Collective; gathers must be issued in the same order on every rank and
   /// stream-ordered among themselves.  The reductions' wire (Reductions.h:71-88: MPI_Allreduce there).
   int allGather(const void *In, int NVals, void *Out, hipStream_t S);
   I8 NGathers = 0;

   /// 0, or the sticky failure raised by a wait kernel that gave up (bit 0, value 1: "consumed" wait; bit 1, value 2:
   /// "arrived" wait; bit 2, value 4: an allGather's wait for a rank's values -- the sum it fed is reported as failed).
   /// Once it is raised the unpack kernel of that exchange and of every later one copies nothing and no "consumed"
   /// signal leaves this rank (the neighbours' next exchange then gives up too: the failure spreads instead of a state
   /// with stale halos); the host sees it here after synchronising with the exchange's stream, and the next put() fails.
   int status() const;
   /// the status word as kernels read it (pinned host memory)
   const int *statusWord() const { return Status; }
   const std::string &lastError() const { return LastError; }
   /// how long a wait kernel spins before it gives up [s] (default 60)
   void setTimeout(double Seconds);

 private:
   size_t MailboxBytes;
   void *Mailbox = nullptr;
   /// [2 * NRanks]: arrived[r], consumed[r] written by rank r; then [NRanks] gather sequence numbers and
   /// [2][NRanks][MaxGatherVals] gather slots (allGather)
   unsigned long long *Flags = nullptr;
   size_t flagWords() const { return (size_t)NRanks * (3 + 2 * (size_t)MaxGatherVals); }
   int *Status               = nullptr; ///< pinned host word the wait kernels raise
   bool Connected            = false;
   long long TimeoutTicks;
   std::vector<void *> PeerMailbox;               ///< opened mappings, by rank
   std::vector<unsigned long long *> PeerFlags;
   std::vector<size_t> PeerMailboxBytes;
   std::string LastError;
};

// kernels/PeerKernels.hip
struct PeerFlagPtrs {
   unsigned long long *P[PeerWire::MaxPeers];
   int N;
};
struct PeerFlagIdx {
   int I[PeerWire::MaxPeers];
   int N;
};
/// wait until Local[Idx.I[i]] >= Seq for every i (bounded; raises *Status |= Bit on timeout)
void launchPeerWait(const unsigned long long *Local, const PeerFlagIdx &Idx, unsigned long long Seq, int *Status, int Bit,
                    long long TimeoutTicks, hipStream_t S);
struct PeerBlockPtrs {
   unsigned long long *P[PeerWire::MaxRanksGather]; ///< every rank's flag block as mapped here ([Rank]: the local one)
};
void launchPeerAllGather(const PeerBlockPtrs &Blocks, int NRanks, int Rank, const unsigned long long *In, int NVals,
                         unsigned long long *Out, unsigned long long Seq, int *Status, long long TimeoutTicks,
                         hipStream_t S);
/// *Remote.P[i] = Seq (system-scope release), then optionally wait as above.  Wait == false (the "consumed" signal after
/// the unpack kernel): nothing is signalled if *Status is already raised.
void launchPeerSignalWait(const PeerFlagPtrs &Remote, unsigned long long Seq, const unsigned long long *Local,
                          const PeerFlagIdx &Idx, bool Wait, int *Status, int Bit, long long TimeoutTicks, hipStream_t S);

} // namespace OMEGA
#endif
