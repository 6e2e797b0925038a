// Base.h -- basic types, error convention and device-array handles.
//
// Layout contract taken from the reference (no Kokkos here): Real = double,
// I4 = int32, LayoutRight (last index = vertical level, contiguous), every
// mesh-indexed array has NXxSize = NXxAll + 1 rows, the last being the zero
// sentinel row that missing neighbours point to
// (reference: components/omega/src/base/DataTypes.h:19-94,
//  components/omega/src/base/Decomp.cpp:553-574, 1082).
#ifndef OMEGA_AMD_BASE_H
#define OMEGA_AMD_BASE_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace OMEGA {

using I4   = int32_t;
using I8   = int64_t;
using R8   = double;
using Real = double;

/// Mesh element kinds (reference: components/omega/src/base/Halo.h:45)
enum MeshElement { OnCell, OnEdge, OnVertex };

/// Error convention: host code throws OmegaError; the C ABI catches it, stores the
/// message (omg_last_error) and returns a non-zero int.  This mirrors the reference's
/// ABORT_ERROR / int-return-code pair (components/omega/src/infra/Error.h:207-270)
/// without tearing the process down inside a library.
struct OmegaError : std::runtime_error {
   using std::runtime_error::runtime_error;
};

[[noreturn]] void abortError(const char *File, int Line, const std::string &Msg);

#define OMEGA_ABORT(msg) ::OMEGA::abortError(__FILE__, __LINE__, (msg))
#define OMEGA_REQUIRE(cond, msg)                                               \
   do {                                                                        \
      if (!(cond))                                                             \
         ::OMEGA::abortError(__FILE__, __LINE__, (msg));                       \
   } while (0)
#define HIP_CHECK(call)                                                        \
   do {                                                                        \
      hipError_t e_ = (call);                                                  \
      if (e_ != hipSuccess)                                                    \
         ::OMEGA::abortError(__FILE__, __LINE__,                               \
                             std::string(#call) + ": " + hipGetErrorString(e_)); \
   } while (0)

/// Owning device allocation (zero-initialised, like a Kokkos::View).
class DeviceBuffer {
 public:
   DeviceBuffer() = default;
   explicit DeviceBuffer(size_t Bytes);
   ~DeviceBuffer();
   DeviceBuffer(const DeviceBuffer &)            = delete;
   DeviceBuffer &operator=(const DeviceBuffer &) = delete;
   void *Ptr    = nullptr;
   size_t Bytes = 0;
};

/// Row pitch (in values) of a device array whose last index is the vertical level: a column of K levels is
/// padded to whole 128-byte cache lines (16 doubles) once it is at least one line long, so every column starts
/// on a line boundary and the kernels' 16-level chunks never straddle two lines (K = 60 -> 64: the EC30to60 /
/// QU240 configurations).  Host arrays and message buffers stay compact ([rows][K]); only copies see the pitch.
inline int levelPitch(int K) { return (K >= 16 && K % 16 != 0) ? (K + 15) / 16 * 16 : K; }

/// Rank-N device array handle: pointer + extents, shared ownership of the
/// allocation (copies alias, as Kokkos views do).  `Pitch` is the distance in values between consecutive
/// rows of the last index (== Ext[N-1] unless the array was made with levels()).
template <class T, int N> struct DeviceArray {
   T *Ptr = nullptr;
   int Ext[N > 0 ? N : 1] = {};
   int Pitch = 0;
   std::string Label;
   std::shared_ptr<DeviceBuffer> Buf;

   DeviceArray() = default;
   DeviceArray(const std::string &L, int E0, int E1 = 1, int E2 = 1, int Pitch_ = 0) : Label(L) {
      int E[3] = {E0, E1, E2};
      size_t Cnt = 1;
      for (int I = 0; I < N; ++I) {
         Ext[I] = E[I];
         Cnt *= (size_t)(I == N - 1 && Pitch_ > 0 ? Pitch_ : E[I]);
      }
      Pitch = Pitch_ > 0 ? Pitch_ : Ext[N - 1];
      Buf   = std::make_shared<DeviceBuffer>(Cnt * sizeof(T));
      Ptr   = static_cast<T *>(Buf->Ptr);
   }
   /// an array whose last index is the vertical level: rows padded to levelPitch(K) (zero-filled pad)
   static DeviceArray levels(const std::string &L, int E0, int E1 = 1, int E2 = 1) {
      const int K = N == 1 ? E0 : (N == 2 ? E1 : E2);
      return DeviceArray(L, E0, E1, E2, levelPitch(K));
   }
   /// number of rows of the last index (product of the other extents)
   size_t rows() const {
      size_t Cnt = Ptr ? 1 : 0;
      for (int I = 0; I + 1 < N; ++I)
         Cnt *= (size_t)Ext[I];
      return Cnt;
   }
   /// logical number of values (what a compact host copy holds)
   size_t size() const {
      size_t Cnt = Ptr ? 1 : 0;
      for (int I = 0; I < N; ++I)
         Cnt *= (size_t)Ext[I];
      return Cnt;
   }
   size_t bytes() const { return size() * sizeof(T); }
   int extent_int(int I) const { return Ext[I]; }
   T *data() const { return Ptr; }
   const std::string &label() const { return Label; }
};

using Array1DI4   = DeviceArray<I4, 1>;
using Array2DI4   = DeviceArray<I4, 2>;
using Array1DReal = DeviceArray<Real, 1>;
using Array2DReal = DeviceArray<Real, 2>;
using Array3DReal = DeviceArray<Real, 3>;

/// Host arrays are plain vectors with extents.
template <class T> struct HostArray {
   std::vector<T> V;
   int Ext[3] = {0, 1, 1};
   HostArray() = default;
   HostArray(int E0, int E1 = 1, int E2 = 1, T Fill = T()) : V((size_t)E0 * E1 * E2, Fill) {
      Ext[0] = E0;
      Ext[1] = E1;
      Ext[2] = E2;
   }
   T &operator()(int I) { return V[I]; }
   const T &operator()(int I) const { return V[I]; }
   T &operator()(int I, int J) { return V[(size_t)I * Ext[1] + J]; }
   const T &operator()(int I, int J) const { return V[(size_t)I * Ext[1] + J]; }
   T *data() { return V.data(); }
   const T *data() const { return V.data(); }
   size_t size() const { return V.size(); }
};
using HostArrayI4   = HostArray<I4>;
using HostArrayReal = HostArray<Real>;

/// The reference's object registries (e.g. Tendencies::create / get / getDefault / erase / clear,
/// components/omega/src/ocn/Tendencies.h:105-144; the same pattern in OceanState.h:100-149, AuxiliaryState.h:49-65,
/// Halo.h:258-279, HorzMesh.h, Decomp.h, TimeStepper.h:87-139): objects are created by name, owned by a static
/// map<string, unique_ptr<T>>, looked up by name, and the one called "Default" is what getDefault() returns.
/// create() forwards its arguments after the name to the constructor T(Name, ...); a second create with the
/// same name fails (returns nullptr, as the reference logs an error and returns nullptr); get() of a missing name
/// returns nullptr.  `init()` of the reference builds the default object from the YAML configuration, which is out
/// of scope here: the host code creates "Default" itself.
template <class T> class Registry {
 public:
   template <class... A> static T *create(const std::string &Name, A &&...Args) {
      auto &M = all();
      if (M.find(Name) != M.end())
         return nullptr;
      T *Obj = new T(Name, std::forward<A>(Args)...);
      M[Name].reset(Obj);
      return Obj;
   }
   static T *get(const std::string &Name) {
      auto &M  = all();
      auto It = M.find(Name);
      return It == M.end() ? nullptr : It->second.get();
   }
   static T *getDefault() { return get("Default"); }
   static void erase(const std::string &Name) { all().erase(Name); }
   static void clear() { all().clear(); }

 private:
   static std::map<std::string, std::unique_ptr<T>> &all() {
      static std::map<std::string, std::unique_ptr<T>> M;
      return M;
   }
};

// ---- device helpers (Device.cpp) ----
void deviceInit(int DeviceId);                 ///< hipSetDevice + sanity check (gfx950)
/// Number of device resources the library has created so far in this process: device buffers (DeviceBuffer, the peer
/// wire's mailbox and flags), streams and events.  Everything a time step needs is created when its objects are
/// initialised (the reference allocates in the constructors / finalizeInit, RungeKutta4Stepper.cpp:43-64), so this
/// number does not move across doStep calls (tests/test_gpu_properties.py).
I8 deviceResourceCount();
void noteDeviceResource(int N = 1);
void copyToDevice(void *Dst, const void *Src, size_t Bytes, hipStream_t S = nullptr);
void copyToHost(void *Dst, const void *Src, size_t Bytes, hipStream_t S = nullptr);
/// rows x width values between a compact host array and a device array of row pitch `Pitch` (values)
void copyRowsToDevice(Real *Dst, int Pitch, const Real *Src, size_t Rows, int Width);
void copyRowsToHost(Real *Dst, const Real *Src, int Pitch, size_t Rows, int Width);
void deviceFill0(void *Dst, size_t Bytes, hipStream_t S);
void deviceCopy(void *Dst, const void *Src, size_t Bytes, hipStream_t S);

/// compact host data <-> a (possibly level-padded) device array
template <int N> void copyToDevice(const DeviceArray<Real, N> &D, const Real *Host) {
   copyRowsToDevice(D.Ptr, D.Pitch, Host, D.rows(), D.Ext[N - 1]);
}
template <int N> void copyToHost(Real *Host, const DeviceArray<Real, N> &D) {
   copyRowsToHost(Host, D.Ptr, D.Pitch, D.rows(), D.Ext[N - 1]);
}

template <class T, int N>
DeviceArray<T, N> createDeviceMirrorCopy(const std::string &L, const HostArray<T> &H) {
   DeviceArray<T, N> D(L, H.Ext[0], H.Ext[1], H.Ext[2]);
   copyToDevice(D.Ptr, H.data(), D.bytes());
   return D;
}

} // namespace OMEGA
#endif
