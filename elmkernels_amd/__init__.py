"""elmkernels_amd - MI355X-native (gfx950 / HIP) per-gridcell land-surface physics.

A drop-in for the per-column water+energy hot path of CANGA/ELMKernels: the seven L3 wrappers
kokkos_frac_wet, kokkos_albedo_snicar, kokkos_canopy_hydrology, kokkos_surface_radiation,
kokkos_canopy_temperature, kokkos_bareground_fluxes, kokkos_canopy_fluxes (driver/kokkos/*_kokkos.hh of the
reference) as hand-written HIP kernels behind the C ABI of include/elmk.h.  This package is the thin host
mirror of that interface; see DESIGN.md.
"""
from .decomp import all_ranges, block_range  # noqa: F401
from .state import (  # noqa: F401
    KERNEL_NAMES,
    ELMInterface,
    ELMState,
    initialize_kokkos_elm,
    kokkos_albedo_snicar,
    kokkos_bareground_fluxes,
    kokkos_canopy_fluxes,
    kokkos_soil_temperature,
    kokkos_surface_fluxes,
    kokkos_init_timestep,
    get_forcing,
    compute_phenology,
    forcing_time_weights,
    kokkos_evaluate_conservation,
    kokkos_canopy_hydrology,
    kokkos_canopy_temperature,
    kokkos_frac_wet,
    kokkos_surface_radiation,
    timestep7,
    timestep7_fused,
    advance_physics,
    kokkos_snow_hydrology,
)
