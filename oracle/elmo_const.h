/* elmo_const.h - physical constants and land-unit codes for the oracle.
 * Values restate src/data/elm_constants.h:18-53 and src/data/land_data.h:8-31 of the reference
 * (derived constants are formed by the same expressions so they round identically).
 * TEST INFRASTRUCTURE - see elm_oracle.h. */
#ifndef ELMO_CONST_H
#define ELMO_CONST_H

#define TFRZ 273.15
#define ELM_PI 3.14159265358979323846
#define BOLTZ 1.38065e-23
#define AVOGAD 6.02214e26
#define MWWV 18.016
#define RGAS (AVOGAD * BOLTZ)
#define RWV (RGAS / MWWV)
#define STEBOL 5.67e-8
#define MWDAIR 28.966
#define RAIR (RGAS / MWDAIR)
#define GRAV 9.80616
#define ROVERG (RWV / GRAV * 1000.)
#define O2_MOLAR_CONST 0.209
#define CO2_PPMV 355.0
#define DENICE 0.917e3
#define DENH2O 1.000e3
#define HVAP 2.501e6
#define HFUS 3.337e5
#define HSUB (HVAP + HFUS)
#define VKC 0.4
#define CPAIR 1.00464e3
#define CSOILC 0.004
#define ZLND 0.01
#define ZSNO 0.0024
#define SNW_RDS_MIN 54.526
#define SPVAL 1.0e36

enum {
  istsoil = 1,
  istcrop = 2,
  istice = 3,
  istice_mec = 4,
  istdlak = 5,
  istwet = 6,
  icol_roof = 71,
  icol_sunwall = 72,
  icol_shadewall = 73,
  icol_road_imperv = 74,
  icol_road_perv = 75,
  pft_nsoybean = 23,
  pft_nsoybeanirrig = 24
};


/* std::min / std::max semantics of the reference (<algorithm>): first argument wins ties and NaNs */
static inline double dmin(double a, double b) { return (b < a) ? b : a; }
static inline double dmax(double a, double b) { return (a < b) ? b : a; }

#endif
