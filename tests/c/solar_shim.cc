// extern "C" view of the solar-geometry helpers of include/elmk_interface.hpp, for tests/test_host_side.py
#include "elmk_interface.hpp"
extern "C" void elmk_test_solar(long n, const double* lat, const double* lon, const double* dt, const double* jday, double* cosz,
                                double* dayl, double* max_dayl)
{
  for (long i = 0; i < n; i++) {
    cosz[i] = elmk::solar::average_cosz(lat[i], lon[i], dt[i], jday[i]);
    dayl[i] = elmk::solar::daylength(lat[i], elmk::solar::declination_angle_sin(static_cast<int>(jday[i])));
    max_dayl[i] = elmk::solar::max_daylength(lat[i]);
  }
}
