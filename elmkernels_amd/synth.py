"""Synthetic column states for tests and the benchmark.

The only physically consistent 20-level column data that ships with the reference are its test fixtures
(an ELM single-column run at US-Brw, test/data/*_IN.txt, committed here as tests/golden/*.npz).  A state for
the whole 7-kernel timestep is assembled from them: fixture step t of every module is the same model time,
so column k of the base block takes step 2+k of each module's _IN record (first module in call order that
carries a field wins, so every field has its start-of-timestep value).

  tier "A" (fixture-tiled)   47 base columns (steps 2..48): vegetated, snl = 0, day/night mix ~50/50.
  tier "B" (branch-mix)      tier A plus, per column and seeded: 30 % bare ground, snow layers snl in 0..5
                             with a consistent snowpack, 5 % capped snow, 10 % ponded surface water,
                             10 % C4 grass (vtype 14), aerosols in snow, cold/warm forcing for all three
                             snowfall-density regimes.

Larger states tile the base block (column c <- base column c mod nbase) with small multiplicative / additive
perturbations of forcing-like fields; the same rule runs on the device (elmk_tile_columns) for big N.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

# call order of ELMInterface::advance (albedo first, canopy fluxes last)
MODULE_ORDER = [
    "SurfaceAlbedo", "CanopyHydrology", "CanopySunShadeFractions", "SurfaceRadiation", "CanopyTemperature",
    "BareGroundFluxes", "CanopyFluxes",
]
RENAME = {
    "forc_t": "forc_tbot", "forc_q": "forc_qbot", "forc_th": "forc_thbot", "z": "zsoi", "zi": "zisoi",
    "mss_cnc_bcphi": "cnc_bcphi", "mss_cnc_bcpho": "cnc_bcpho", "mss_cnc_dst1": "cnc_dst1",
    "mss_cnc_dst2": "cnc_dst2", "mss_cnc_dst3": "cnc_dst3", "mss_cnc_dst4": "cnc_dst4",
    "albsnd_hst": "albsnd", "albsni_hst": "albsni",
}
TFRZ = 273.15
TEST_LAND = dict(ltype=1, ctype=1, vtype=12, urbpoi=0, lakpoi=0)
DTIME = 1800.0

# forcing-like fields perturbed when a base block is tiled: (field, mode, amp); mode 0: *(1+amp*u), 1: +amp*u
TILE_RULES = [
    ("forc_rain", 0, 0.02), ("forc_snow", 0, 0.02), ("forc_lwrad", 0, 0.02), ("forc_u", 0, 0.02),
    ("forc_v", 0, 0.02), ("forc_pbot", 0, 0.002), ("forc_qbot", 0, 0.02), ("forc_solad", 0, 0.02),
    ("forc_solai", 0, 0.02), ("h2ocan", 0, 0.02), ("forc_tbot", 1, 0.5), ("forc_thbot", 1, 0.5),
    ("t_grnd", 1, 0.5), ("t_veg", 1, 0.5),
]


def load_params():
    """-> (pft npz, snow-optics npz): the shared parameter tables (converted reference data files)."""
    return np.load(os.path.join(GOLDEN, "pft_params.npz")), np.load(os.path.join(GOLDEN, "SnowOptics.npz"))


def base_columns(field_table):
    """Tier-A base block. field_table: {name: (id, nlev, dtype)} -> ({name: array [n, nlev] or [n]}, extras)."""
    steps = np.arange(2, 49)
    n = len(steps)
    cols = {}
    extras = {}
    for mod in MODULE_ORDER:
        d = np.load(os.path.join(GOLDEN, mod + ".npz"))
        rows = np.searchsorted(d["steps"], steps)
        assert np.array_equal(d["steps"][rows], steps), mod
        for key in d.files:
            if not key.startswith("in/"):
                continue
            label = key[3:]
            name = RENAME.get(label, label)
            arr = d[key][rows]
            if name in field_table and arr.shape[1] == field_table[name][1]:
                if name not in cols:
                    cols[name] = arr
            elif label not in extras:
                extras[label] = arr
    out = {}
    for name, (_, nlev, dt) in field_table.items():
        if name == "err_flags":
            continue
        if name in cols:
            a = np.nan_to_num(cols[name], nan=0.0)
            a = np.where(np.abs(a) >= 1e30, 0.0, a)  # 1e+36 "unset" sentinels of pure outputs
        else:
            a = np.zeros((n, nlev))
        a = a.astype(dt)
        out[name] = a if nlev > 1 else a[:, 0]
    # fields no fixture carries
    out["vtype"][:] = TEST_LAND["vtype"]
    out["veg_active"][:] = 1
    out["isoicol"][:] = 3
    for k in "utq":  # driver resets the patch heights to the forcing height every step (atm_physics_impl.hh:197-203)
        out[f"forc_hgt_{k}_patch"][:] = extras["forc_hgt_" + k][:, 0]
    out["watdry"][:] = 0.0
    out["watopt"][:] = 0.0
    # soil thermal parameters (soil_texture_hydraulic_model in the reference driver; no fixture carries them):
    # mineral conductivity, dry conductivity, solid heat capacity - plausible loam values varied over level and column
    if "tkmg" in out:
        k = (np.arange(n)[:, None] * 7 % 11 - 5) / 50.0
        j = np.arange(15)[None, :]
        out["tkmg"][:] = (2.0 + 0.08 * j) * (1.0 + k)
        out["tkdry"][:] = (0.20 + 0.005 * j) * (1.0 - k)
        out["csol"][:, 5:] = (2.0e6 + 2.0e4 * j) * (1.0 + 0.5 * k)
    scal = dict(
        dewmx=float(extras["dewmx"][0, 0]), oldfflag=int(extras["oldfflag"][0, 0]),
        dayl=float(extras["dayl"][n // 2, 0]), max_dayl=float(extras["max_dayl"][n // 2, 0]),
    )
    soil = dict(albsat=np.tile(extras["albsat"][0], (20, 1)), albdry=np.tile(extras["albdry"][0], (20, 1)))
    return out, scal, soil


def branch_mix(cols, seed=0x5EEDE1A0):
    """Tier B: rewrite a copy of `cols` so that the branches the fixtures never take are exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = {k: v.copy() for k, v in cols.items()}
    n = c["snl"].shape[0]
    u = lambda: rng.random(n)  # noqa: E731

    # --- forcing temperature across the three snowfall-density regimes (canopy_hydrology_impl.hh:185-191)
    regime = rng.integers(0, 3, n)
    forc_t = np.where(regime == 0, TFRZ + 2.5 + 8 * u(), np.where(regime == 1, TFRZ - 14 + 15 * u(), TFRZ - 30 + 14 * u()))
    dT = forc_t - c["forc_tbot"]
    c["forc_tbot"] = forc_t
    c["forc_thbot"] = c["forc_thbot"] + dT
    qs = 0.622 * 611.0 * np.exp(17.3 * (forc_t - TFRZ) / (forc_t - 35.86)) / c["forc_pbot"]
    c["forc_qbot"] = np.minimum(c["forc_qbot"], 0.9 * qs)
    cold = forc_t < TFRZ
    prec = c["forc_rain"] + c["forc_snow"] + 1e-5 * (u() < 0.5)
    c["forc_snow"] = np.where(cold, prec, 0.2 * prec * (u() < 0.3))
    c["forc_rain"] = prec - c["forc_snow"]
    c["t_veg"] = c["t_veg"] + dT
    c["t_h2osfc"] = np.maximum(c["t_h2osfc"] + dT, TFRZ - 20)
    c["t10"] = c["t10"] + 0.5 * dT

    # --- snow layers with a consistent pack: dz = snow_depth/snl, ice = 250*dz, cold layers, grain radius in table
    snl = np.where(u() < 0.5, 0, rng.integers(1, 6, n)).astype(np.int32)
    snl = np.where(cold | (u() < 0.3), snl, 0).astype(np.int32)
    depth = np.where(snl > 0, snl * (0.03 + 0.2 * u()), 0.0)
    # capped snow only exists on a deep pack (h2osno > 1000 mm): 5 layers, > 4 m at 250 kg/m3
    cap = (snl == 5) & (u() < 0.4)
    depth = np.where(cap, 4.1 + u(), depth)
    for lev in range(5):
        act = lev >= 5 - snl
        dzl = np.where(act, depth / np.maximum(snl, 1), 0.0)
        c["dz"][:, lev] = dzl
        c["h2osoi_ice"][:, lev] = np.where(act, 250.0 * dzl, 0.0)
        c["h2osoi_liq"][:, lev] = np.where(act, 5.0 * dzl * (u() < 0.3), 0.0)
        c["t_soisno"][:, lev] = np.where(act, np.minimum(TFRZ - 1.0, forc_t) - 2 * u(), 0.0)
        c["snw_rds"][:, lev] = np.where(act, 54.526 + (1500.0 - 54.526) * u() ** 2, 0.0)
        c["frac_iceold"][:, lev] = np.where(act, 1.0, 0.0)
        for a in ("cnc_bcphi", "cnc_bcpho", "cnc_dst1", "cnc_dst2", "cnc_dst3", "cnc_dst4"):
            c[a][:, lev] = np.where(act & (u() < 0.5), 1e-7 * u(), 0.0)
    # interfaces / node depths of the snow levels (negative upward from the soil surface)
    zi_run = np.zeros(n)
    for lev in range(4, -1, -1):
        act = lev >= 5 - snl
        c["zsoi"][:, lev] = np.where(act, zi_run - 0.5 * c["dz"][:, lev], 0.0)
        zi_run = np.where(act, zi_run - c["dz"][:, lev], zi_run)
        c["zisoi"][:, lev] = np.where(act, zi_run, 0.0)
    h2osno_layers = (c["h2osoi_ice"][:, :5] + c["h2osoi_liq"][:, :5]).sum(axis=1)
    thin = (snl == 0) & (u() < 0.5)  # snow on the ground without a resolved layer (the fixtures' only case)
    c["h2osno"] = np.where(snl > 0, h2osno_layers, np.where(thin, 0.5 + 5 * u(), 0.0))
    c["snow_depth"] = np.where(snl > 0, depth, np.where(thin, c["h2osno"] / 250.0, 0.0))
    c["frac_sno"] = np.where(c["h2osno"] > 0, np.clip(np.tanh(c["snow_depth"] / 0.05) + 0.05, 0.05, 1.0), 0.0)
    c["frac_sno_eff"] = c["frac_sno"].copy()
    c["int_snow"] = c["h2osno"] * (1.0 + u())
    c["qflx_snow_melt"] = np.where(c["h2osno"] > 0, 2e-5 * (u() < 0.5), 0.0)
    c["snl"] = snl
    c["swe_old"][:] = 0.0

    # --- soil: fully frozen columns (btran = 0), partly frozen profiles with soil ice, dry top soil
    frozen = u() < 0.10
    partly = (~frozen) & (u() < 0.25)
    nfrz = rng.integers(1, 15, n)
    for j in range(15):
        lev = 5 + j
        fr = frozen | (partly & (j >= nfrz))
        c["t_soisno"][:, lev] = np.where(fr, np.minimum(c["t_soisno"][:, lev], TFRZ - 0.5 - 3 * u()), c["t_soisno"][:, lev])
        moved = np.where(fr, 0.7 * c["h2osoi_liq"][:, lev], 0.0)
        c["h2osoi_ice"][:, lev] = c["h2osoi_ice"][:, lev] + moved
        c["h2osoi_liq"][:, lev] = c["h2osoi_liq"][:, lev] - moved
    dry = u() < 0.15
    c["h2osoi_liq"][:, 5] = np.where(dry, 0.15 * c["h2osoi_liq"][:, 5], c["h2osoi_liq"][:, 5])
    c["h2osoi_vol"][:, 0] = np.where(dry, 0.15 * c["h2osoi_vol"][:, 0], c["h2osoi_vol"][:, 0])

    # --- capped snow, ponded surface water, bare ground, C4 grass
    c["do_capsnow"] = cap.astype(np.int32)
    pond = u() < 0.10
    c["h2osfc"] = np.where(pond, 1e-8 + 5.0 * u(), np.where(u() < 0.5, 0.0, 5e-9 * u()))
    c["frac_h2osfc"] = np.where(pond, 0.05 + 0.2 * u(), 0.0)
    c["frac_sno"] = np.minimum(c["frac_sno"], 1.0 - c["frac_h2osfc"])
    c["frac_sno_eff"] = c["frac_sno"].copy()
    bare = u() < 0.30
    c["frac_veg_nosno"] = np.where(bare, 0, 1).astype(np.int32)
    c["vtype"] = np.where(u() < 0.10, 14, 12).astype(np.int32)
    c["isoicol"] = rng.integers(0, 20, n).astype(np.int32)
    # sun angle: keep the fixture's day/night split, add low sun for the SNICAR zenith correction
    low = (c["coszen"] > 0) & (u() < 0.3)
    c["coszen"] = np.where(low, 0.02 + 0.2 * u(), c["coszen"])
    return c


def soil_color_tables(seed=7):
    """20 plausible soil colour classes (saturated / dry albedo, VIS and NIR) for tier B."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sat = 0.05 + 0.2 * rng.random((20, 1)) + np.array([[0.0, 0.1]])
    dry = sat + 0.1 + 0.05 * rng.random((20, 2))
    return sat, dry


def tile(cols, n, seed=0x5EEDE1A0, rules=TILE_RULES, perturb=True):
    """Host version of elmk_tile_columns for small n: column c <- base column c % nbase (+ perturbation)."""
    nbase = next(iter(cols.values())).shape[0]
    idx = np.arange(n) % nbase
    out = {k: np.ascontiguousarray(v[idx]) for k, v in cols.items()}
    if perturb and n > nbase:
        rng = np.random.Generator(np.random.PCG64(seed))
        for name, mode, amp in rules:
            a = out[name]
            uu = rng.uniform(-1.0, 1.0, a.shape)
            uu[:nbase] = 0.0
            out[name] = a * (1.0 + amp * uu) if mode == 0 else a + amp * uu
    return out


def make_state(field_table, n, tier="A", seed=0x5EEDE1A0, perturb=True):
    """-> (columns dict [n,...], scalars, soil-colour tables): ready to upload into an ELMState / oracle state."""
    base, scal, soil = base_columns(field_table)
    if tier == "B":
        reps = max(1, min(64, (n + 46) // 47))
        base = tile(base, 47 * reps, seed=seed, perturb=False)
        base = branch_mix(base, seed=seed)
        sat, dry = soil_color_tables()
        soil = dict(albsat=sat, albdry=dry)
    cols = tile(base, n, seed=seed, perturb=perturb) if n != next(iter(base.values())).shape[0] else base
    if "dtbegin_column_h2o" in cols:  # what the driver records at the start of a step (conservation diagnostics)
        cols["dtbegin_column_h2o"] = (cols["h2ocan"] + cols["h2osno"] + cols["h2osfc"]
                                      + (cols["h2osoi_ice"] + cols["h2osoi_liq"]).sum(axis=1))
        cols["h2osno_old"] = cols["h2osno"].copy()
    if "frac_veg_nosno_alb" in cols:
        cols["frac_veg_nosno_alb"] = cols["frac_veg_nosno"].copy()
    if "atm_tbot" in cols:
        cols.update(forcing_streams(cols, seed))
    if "mss_bcphi" in cols:
        cols.update(snow_aerosols(cols, seed))
    if "pct_sand" in cols:
        cols.update(init_inputs(cols, seed))
    return cols, scal, soil


ORGANIC_MAX = 130.0  # kg/m3, the value of the E3SM parameter file (read at initialize_elm_kokkos.cc:320)


def init_inputs(cols, seed):
    """Inputs of the cold-start initialisation (initialize_elm_kokkos.cc:373-428) no fixture carries: surface-data slope and
    elevation standard deviation, soil texture and organic matter by level (surfdata's PCT_SAND / PCT_CLAY / ORGANIC)."""
    n = cols["snl"].shape[0]
    rng = np.random.default_rng(seed + 8191)
    out = {}
    out["topo_slope"] = np.where(rng.random(n) < 0.2, 0.2 * rng.random(n), 12.0 * rng.random(n) ** 2)  # both sides of the 0.2 floor
    out["topo_std"] = np.where(rng.random(n) < 0.2, 10.0 * rng.random(n), 400.0 * rng.random(n))       # both sides of the 10 m floor
    sand = 5.0 + 85.0 * rng.random((n, 15))
    clay = (95.0 - sand) * rng.random((n, 15)) + 1.0
    out["pct_sand"], out["pct_clay"] = sand, clay
    org = ORGANIC_MAX * rng.random((n, 15)) ** 2                # mostly mineral soil, some peat (om_frac up to 1)
    org[rng.random(n) < 0.05] = ORGANIC_MAX                      # pure organic columns: the om_frac == 1 branch
    org[rng.random((n, 15)) < 0.3] = 0.0
    out["organic"] = org
    return out


def init_snow_depths(n, seed):
    """Snow depths that reach every branch of init_snow_layers (init_snow_state_impl.hh:67-151), its bin edges included."""
    rng = np.random.default_rng(seed + 12289)
    edges = np.array([0.0, 0.01, 0.03, 0.04, 0.07, 0.12, 0.18, 0.29, 0.41, 0.64, 1.5])
    k = rng.integers(0, len(edges) - 1, n)
    d = edges[k] + (edges[k + 1] - edges[k]) * rng.random(n)
    exact = rng.random(n) < 0.1
    d[exact] = edges[1:][rng.integers(0, len(edges) - 1, int(exact.sum()))]
    return d


def snow_aerosols(cols, seed):
    """Inputs of kokkos_snow_hydrology no fixture carries: aerosol masses consistent with the concentrations SNICAR reads
    (mss = cnc * layer water mass, what update_aerosol_mass_and_concen maintains), deposition rates of the eleven
    AerosolFileInput streams (kg/m2/s), and the previous step's dew / sublimation / evaporation on the snow surface."""
    n = cols["snl"].shape[0]
    rng = np.random.default_rng(seed + 4099)
    out = {}
    mass = cols["h2osoi_ice"][:, :5] + cols["h2osoi_liq"][:, :5]
    for a in ("bcphi", "bcpho", "dst1", "dst2", "dst3", "dst4"):
        out["mss_" + a] = cols["cnc_" + a] * mass
    for a in ("bcphi", "bcpho", "bcdep", "dst1_1", "dst1_2", "dst2_1", "dst2_2", "dst3_1", "dst3_2", "dst4_1", "dst4_2"):
        out["aer_" + a] = 1e-12 * rng.random(n) * (rng.random(n) < 0.7)
    snow = cols["h2osno"] > 0
    out["qflx_sub_snow"] = np.where(snow & (rng.random(n) < 0.4), 2e-6 * rng.random(n), 0.0)
    out["qflx_dew_snow"] = np.where(snow & (out["qflx_sub_snow"] == 0) & (rng.random(n) < 0.3), 1e-6 * rng.random(n), 0.0)
    out["qflx_evap_grnd"] = np.where(rng.random(n) < 0.4, 3e-6 * rng.random(n), 0.0)
    out["qflx_dew_grnd"] = np.where((out["qflx_evap_grnd"] == 0) & (rng.random(n) < 0.3), 1e-6 * rng.random(n), 0.0)
    return out


def snow_age_tables(seed=11):
    """SnwRdsTable (snicar_data.h:75-84): snowage_tau / kappa / drdt0 [11, 31, 8].  The reference reads them from
    snicar_drdt_bst_fit_60_c070416.nc, which is not in its repository: synthetic positive values of the right order of
    magnitude (hours, unitless, um/hr), smooth in temperature / gradient / density."""
    rng = np.random.default_rng(seed)
    T, G, R = np.meshgrid(np.arange(11), np.arange(31), np.arange(8), indexing="ij")
    tau = 50.0 + 30.0 * T + 5.0 * G + 10.0 * R + rng.random((11, 31, 8))
    kappa = 1.2 + 0.05 * T + 0.02 * G + 0.01 * R + 0.01 * rng.random((11, 31, 8))
    drdt0 = 0.5 + 0.3 * T + 0.1 * G + 0.02 * R + 0.01 * rng.random((11, 31, 8))
    return np.stack([tau, kappa, drdt0])


def forcing_streams(cols, seed):
    """Raw forcing records (t_idx, t_idx + 1) and monthly phenology (start_idx, + 1) around the column's current forcing, for
    the init_timestep functors: values on both sides of every clamp of atm_physics_impl.hh (tbot > 323, pbot < 4e4,
    qbot < 1e-9, flds outside / inside (50, 600), negative precipitation and shortwave) and of phenology_physics_impl.hh
    (lai below 0.05, snow burial of short and tall vegetation)."""
    n = cols["forc_tbot"].shape[0]
    rng = np.random.default_rng(seed + 977)

    def two(base, rel=0.05, add=0.0):
        a = np.empty((n, 2))
        a[:, 0] = base * (1 + rel * (rng.random(n) - 0.5)) + add * (rng.random(n) - 0.5)
        a[:, 1] = base * (1 + rel * (rng.random(n) - 0.5)) + add * (rng.random(n) - 0.5)
        return a

    out = {}
    tb = two(cols["forc_tbot"], 0.0, 8.0)
    tb[rng.random(n) < 0.02] = 330.0
    out["atm_tbot"] = tb
    pb = two(cols["forc_pbot"])
    pb[rng.random(n) < 0.02] = 3.0e4
    out["atm_pbot"] = pb
    qb = two(cols["forc_qbot"], 0.2)
    qb[rng.random(n) < 0.02] = 0.0
    out["atm_qbot"] = qb
    fl = two(cols["forc_lwrad"], 0.1)
    sel = rng.random(n)
    fl[sel < 0.15] = 20.0
    fl[sel > 0.9] = 700.0
    out["atm_flds"] = fl
    out["atm_fsds"] = two(np.where(cols["coszen"] > 0, 600.0, 0.0), 0.8, 0.0) - 30.0 * (rng.random((n, 2)) < 0.05)
    pr = two((cols["forc_rain"] + cols["forc_snow"]) + 1e-5, 0.5)
    pr[rng.random(n) < 0.05] = -1e-6
    out["atm_prec"] = pr
    out["atm_wind"] = two(np.hypot(cols["forc_u"], cols["forc_v"]), 0.3)
    lai = 3.0 * rng.random(n)
    lai[rng.random(n) < 0.1] = 0.03
    out["mlai"] = two(lai, 0.3)
    out["msai"] = two(0.3 * lai + 0.02, 0.3)
    top = np.where(rng.random(n) < 0.5, 0.5, 17.0) * (0.5 + rng.random(n))
    out["mhtop"] = two(top, 0.05)
    out["mhbot"] = two(0.1 * top, 0.05)
    return out
