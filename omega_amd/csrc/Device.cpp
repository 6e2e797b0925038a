// Device.cpp -- HIP memory helpers and the error sink.
#include "Base.h"
#include "Pacer.h"
#include "Tuning.h"

#include <rocprofiler-sdk-roctx/roctx.h>

#include <atomic>
#include <cctype>
#include <cstdlib>

#include <cstring>
#include <sstream>

namespace OMEGA {

namespace {
struct OptionName {
   const char *Name;
   int TuningOptions::*Field;
};
const OptionName OptionTable[] = {
    {"MergeL1", &TuningOptions::MergeL1},
    {"Pair", &TuningOptions::Pair},
    {"TracerPatch", &TuningOptions::TracerPatch},
    {"SendBand", &TuningOptions::SendBand},
    {"BandOnComm", &TuningOptions::BandOnComm},
    {"ShrinkSweeps", &TuningOptions::ShrinkSweeps},
    {"ForceGeneric", &TuningOptions::ForceGeneric},
    {"KeepMaxEdges", &TuningOptions::KeepMaxEdges},
    {"NarrowTables", &TuningOptions::NarrowTables},
    {"Graphs", &TuningOptions::Graphs},
};
} // namespace

TuningOptions &tuning() {
   static TuningOptions T = [] {
      TuningOptions X;
#ifdef OMEGA_TUNING_ENV
      // measurement builds only: OMEGA_MERGEL1, OMEGA_PAIR, ... (upper-cased option names)
      for (const OptionName &O : OptionTable) {
         std::string Var = "OMEGA_";
         for (const char *C = O.Name; *C; ++C)
            Var += (char)std::toupper((unsigned char)*C);
         if (const char *V = std::getenv(Var.c_str()))
            X.*(O.Field) = std::atoi(V);
      }
#endif
      return X;
   }();
   return T;
}
namespace {
std::atomic<unsigned long long> TuningGen{0};
}
unsigned long long tuningGeneration() { return TuningGen.load(); }
bool setTuningOption(const std::string &Name, int Value) {
   for (const OptionName &O : OptionTable)
      if (Name == O.Name) {
         tuning().*(O.Field) = Value;
         ++TuningGen;
         return true;
      }
   return false;
}
bool getTuningOption(const std::string &Name, int &Value) {
   for (const OptionName &O : OptionTable)
      if (Name == O.Name) {
         Value = tuning().*(O.Field);
         return true;
      }
   return false;
}

namespace Pacer {
int &timingLevel() {
   static int Level = 3;
   return Level;
}
namespace {
constexpr int MaxLevels = 16;
thread_local int PushedAt[MaxLevels] = {}; // open ranges per level of this thread
int slot(int Level) { return Level < 0 ? 0 : (Level >= MaxLevels ? MaxLevels - 1 : Level); }
} // namespace
bool start(const char *Name, int Level) {
   if (Level <= timingLevel()) {
      roctxRangePushA(Name);
      if (Level > -1000000)
         ++PushedAt[slot(Level)];
   }
   return true;
}
bool stop(const char * /*Name*/, int Level) {
   if (Level <= -1000000) { // a Range that knows its push happened
      roctxRangePop();
      return true;
   }
   if (PushedAt[slot(Level)] > 0) { // pop what a start of this level pushed, whatever the level is now
      --PushedAt[slot(Level)];
      roctxRangePop();
   }
   return true;
}
} // namespace Pacer

void abortError(const char *File, int Line, const std::string &Msg) {
   std::ostringstream OS;
   OS << "[omega_amd] " << File << ":" << Line << ": " << Msg;
   throw OmegaError(OS.str());
}

namespace {
std::atomic<I8> ResourceCount{0};
}
I8 deviceResourceCount() { return ResourceCount.load(); }
void noteDeviceResource(int N) { ResourceCount += N; }

DeviceBuffer::DeviceBuffer(size_t B) : Bytes(B) {
   if (B == 0)
      B = 8;
   HIP_CHECK(hipMalloc(&Ptr, B));
   noteDeviceResource();
   HIP_CHECK(hipMemset(Ptr, 0, B)); // Kokkos views are zero-initialised; the sentinel rows rely on it
   // hipMemset of device memory returns before its fill kernel (queued on the null stream) has run, and work on a
   // non-blocking stream is not ordered after the null stream: an array allocated right before its first use -- the
   // stepper's provisional state at the first step, a halo buffer at the first exchange -- could be zeroed AFTER the
   // first kernels had written it (seen as NaNs in one of five 4-rank runs sharing a GPU).  Allocation is set-up work:
   // wait for the fill here.
   HIP_CHECK(hipStreamSynchronize(nullptr));
}
DeviceBuffer::~DeviceBuffer() {
   if (Ptr)
      (void)hipFree(Ptr);
}

void deviceInit(int DeviceId) {
   int N = 0;
   hipError_t E = hipGetDeviceCount(&N);
   if (E != hipSuccess || N <= 0)
      OMEGA_ABORT("no HIP device visible: the omega_amd product path has no CPU fallback");
   OMEGA_REQUIRE(DeviceId >= 0 && DeviceId < N, "device id out of range");
   HIP_CHECK(hipSetDevice(DeviceId));
}

void copyToDevice(void *Dst, const void *Src, size_t Bytes, hipStream_t S) {
   if (!Bytes)
      return;
   if (S) {
      HIP_CHECK(hipMemcpyAsync(Dst, Src, Bytes, hipMemcpyHostToDevice, S));
      HIP_CHECK(hipStreamSynchronize(S));
   } else {
      HIP_CHECK(hipMemcpy(Dst, Src, Bytes, hipMemcpyHostToDevice));
   }
}
void copyToHost(void *Dst, const void *Src, size_t Bytes, hipStream_t S) {
   if (!Bytes)
      return;
   if (S) {
      HIP_CHECK(hipMemcpyAsync(Dst, Src, Bytes, hipMemcpyDeviceToHost, S));
      HIP_CHECK(hipStreamSynchronize(S));
   } else {
      HIP_CHECK(hipMemcpy(Dst, Src, Bytes, hipMemcpyDeviceToHost));
   }
}
// Host copies are not hot-path calls; they synchronise the device first so that they are ordered against work on
// ANY stream, including the non-blocking streams of omg_stream_create (a plain hipMemcpy only orders against the
// null stream).
void copyRowsToDevice(Real *Dst, int Pitch, const Real *Src, size_t Rows, int Width) {
   if (!Rows || !Width)
      return;
   HIP_CHECK(hipDeviceSynchronize());
   if (Pitch == Width)
      HIP_CHECK(hipMemcpy(Dst, Src, Rows * Width * sizeof(Real), hipMemcpyHostToDevice));
   else
      HIP_CHECK(hipMemcpy2D(Dst, (size_t)Pitch * sizeof(Real), Src, (size_t)Width * sizeof(Real),
                            (size_t)Width * sizeof(Real), Rows, hipMemcpyHostToDevice));
}
void copyRowsToHost(Real *Dst, const Real *Src, int Pitch, size_t Rows, int Width) {
   if (!Rows || !Width)
      return;
   HIP_CHECK(hipDeviceSynchronize());
   if (Pitch == Width)
      HIP_CHECK(hipMemcpy(Dst, Src, Rows * Width * sizeof(Real), hipMemcpyDeviceToHost));
   else
      HIP_CHECK(hipMemcpy2D(Dst, (size_t)Width * sizeof(Real), Src, (size_t)Pitch * sizeof(Real),
                            (size_t)Width * sizeof(Real), Rows, hipMemcpyDeviceToHost));
}
void deviceFill0(void *Dst, size_t Bytes, hipStream_t S) {
   if (Bytes)
      HIP_CHECK(hipMemsetAsync(Dst, 0, Bytes, S));
}
void deviceCopy(void *Dst, const void *Src, size_t Bytes, hipStream_t S) {
   if (Bytes)
      HIP_CHECK(hipMemcpyAsync(Dst, Src, Bytes, hipMemcpyDeviceToDevice, S));
}

} // namespace OMEGA
