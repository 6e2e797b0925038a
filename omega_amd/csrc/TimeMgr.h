// TimeMgr.h -- the slice of the reference's time manager that the hot path's SIGNATURES need, so that a reference call
// site (Tendencies::computeAllTendencies(..., TimeInstant), TimeStepper::doStep(OceanState *, TimeInstant &),
// updateStateByTend(..., TimeInterval), `SimTime + RKC[Stage] * TimeStep`, `Time - ReferenceTime` in a custom tendency)
// compiles unchanged against this backend: TimeFrac (exact integer-fraction seconds), TimeInterval (non-calendar
// intervals only), TimeInstant (fractional seconds since the reference time).  Calendars, alarms, clocks and time strings
// are out of scope (TimeMgr is not on the path); a host model keeps its own Clock and hands the instants
// over.  Arithmetic restated from components/omega/src/infra/TimeMgr.cpp:193-283 (setSeconds: continued fractions),
// :625-679 (+ / - over the least common denominator), :747-767 (Real * fraction), :956-1000 (simplify), :382-391
// (getSeconds), so that `Real * TimeInterval` coefficients carry the reference's bits.
#ifndef OMEGA_AMD_TIMEMGR_H
#define OMEGA_AMD_TIMEMGR_H

#include "Base.h"

namespace OMEGA {

enum class TimeUnits { None = 0, Seconds, Minutes, Hours }; ///< (the reference's calendar units are not supported)

struct TimeFrac {
   I8 Whole = 0, Numer = 0, Denom = 1;
   TimeFrac() = default;
   TimeFrac(I8 W, I8 N, I8 D) : Whole(W), Numer(N), Denom(D) { simplify(); }
   static TimeFrac fromSeconds(R8 Seconds); ///< TimeFrac::setSeconds
   R8 getSeconds() const { return (R8)Whole + (R8)Numer / (R8)Denom; }
   void simplify();
   TimeFrac operator+(const TimeFrac &O) const;
   TimeFrac operator-(const TimeFrac &O) const;
   TimeFrac operator*(R8 Multiplier) const;
   TimeFrac operator*(I4 Multiplier) const;
   bool operator==(const TimeFrac &O) const { return (*this - O).isZero(); }
   bool operator<(const TimeFrac &O) const { return (*this - O).sign() < 0; }
   bool isZero() const { return Whole == 0 && Numer == 0; }
   int sign() const { return Whole != 0 ? (Whole < 0 ? -1 : 1) : (Numer < 0 ? -1 : (Numer > 0 ? 1 : 0)); }
};

class TimeInterval {
 public:
   TimeInterval() = default;
   TimeInterval(I8 Whole, I8 Numer, I8 Denom) : Interval(Whole, Numer, Denom) {}
   TimeInterval(R8 Length, TimeUnits Units) { set(Length, Units); }
   TimeInterval(I4 Length, TimeUnits Units) { set((R8)Length, Units); }
   void set(R8 Length, TimeUnits Units);
   void set(I8 Whole, I8 Numer, I8 Denom) { Interval = TimeFrac(Whole, Numer, Denom); }
   void get(I8 &Whole, I8 &Numer, I8 &Denom) const { Whole = Interval.Whole, Numer = Interval.Numer, Denom = Interval.Denom; }
   void get(R8 &Length, TimeUnits Units) const;
   R8 getSeconds() const { return Interval.getSeconds(); } ///< (extension: what get(Length, TimeUnits::Seconds) returns)
   bool operator==(const TimeInterval &O) const { return Interval == O.Interval; }
   bool operator!=(const TimeInterval &O) const { return !(Interval == O.Interval); }
   bool operator<(const TimeInterval &O) const { return Interval < O.Interval; }
   bool operator>(const TimeInterval &O) const { return O.Interval < Interval; }
   TimeInterval operator+(const TimeInterval &O) const { return wrap(Interval + O.Interval); }
   TimeInterval operator-(const TimeInterval &O) const { return wrap(Interval - O.Interval); }
   TimeInterval &operator+=(const TimeInterval &O) { return *this = *this + O; }
   TimeInterval &operator-=(const TimeInterval &O) { return *this = *this - O; }
   TimeInterval operator*(R8 Multiplier) const { return wrap(Interval * Multiplier); }
   TimeInterval operator*(I4 Multiplier) const { return wrap(Interval * Multiplier); }
   bool isPositive() const { return Interval.sign() > 0; }
   friend TimeInterval operator*(const R8 &Multiplier, const TimeInterval &TI) { return TI * Multiplier; }
   friend TimeInterval operator*(const I4 &Multiplier, const TimeInterval &TI) { return TI * Multiplier; }
   friend class TimeInstant;

 private:
   TimeFrac Interval;
   static TimeInterval wrap(const TimeFrac &F) {
      TimeInterval T;
      T.Interval = F;
      return T;
   }
};

class TimeInstant {
 public:
   TimeInstant() = default;
   /// (extension: the reference builds instants from calendar dates) seconds since the reference time
   static TimeInstant fromSeconds(R8 Seconds) {
      TimeInstant T;
      T.ElapsedTime = TimeFrac::fromSeconds(Seconds);
      return T;
   }
   R8 getSeconds() const { return ElapsedTime.getSeconds(); }
   bool operator==(const TimeInstant &O) const { return ElapsedTime == O.ElapsedTime; }
   bool operator!=(const TimeInstant &O) const { return !(ElapsedTime == O.ElapsedTime); }
   bool operator<(const TimeInstant &O) const { return ElapsedTime < O.ElapsedTime; }
   bool operator>(const TimeInstant &O) const { return O.ElapsedTime < ElapsedTime; }
   TimeInstant operator+(const TimeInterval &I) const { return wrap(ElapsedTime + I.Interval); }
   TimeInstant operator-(const TimeInterval &I) const { return wrap(ElapsedTime - I.Interval); }
   TimeInterval operator-(const TimeInstant &O) const { return TimeInterval::wrap(ElapsedTime - O.ElapsedTime); }
   TimeInstant &operator+=(const TimeInterval &I) { return *this = *this + I; }
   TimeInstant &operator-=(const TimeInterval &I) { return *this = *this - I; }

 private:
   TimeFrac ElapsedTime; ///< fractional seconds since the reference time
   static TimeInstant wrap(const TimeFrac &F) {
      TimeInstant T;
      T.ElapsedTime = F;
      return T;
   }
};

} // namespace OMEGA
#endif
