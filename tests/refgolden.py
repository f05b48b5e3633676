"""Outputs of the REFERENCE ITSELF on branch-mix columns, as a committed fixture (tests/golden/ref_branch_mix.npz).

/root/reference cannot travel to the GPU box, so what its own functions return on seeded synthetic columns - snow layers 0..5,
bare ground, capped snow, ponded water, C4, frozen soil: the branches the bundled single-site fixtures never take - is
recorded here, in the build container, by oracle/_ref (the reference's headers compiled from where they lie), and committed
as data: for every stage of ELMInterface::advance that the reference's headers can run, the fields the stage changed.
The inputs of a stage are not stored: they are the state of the oracle chain at that point, which any host with the same libm
reproduces from the seed; a hash of every stage's input state is stored instead, and a test that finds another hash skips
(its oracle chain is not the one the fixture was recorded on) instead of comparing unlike with unlike.

    python -m tests.refgolden          # (build container) regenerate tests/golden/ref_branch_mix.npz

walk() is the one definition of the sequence; the generator, the CPU test (oracle vs fixture) and the GPU test (HIP vs
fixture) all iterate it."""
import hashlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "ref_branch_mix.npz")
N, TIER, SEED, DT = 1024, "B", 4242, 1800.0

# (stage, how the reference runs it or None, how the oracle runs it)
STAGES = ("init_timestep", "frac_wet", "albedo_snicar", "canopy_hydrology", "surface_radiation", "canopy_temperature",
          "bareground_fluxes", "canopy_fluxes", "soil_temperature", "snow_hydrology", "surface_fluxes")
REF_STAGES = ("init_timestep", "frac_wet", "albedo_snicar", "canopy_hydrology", "surface_radiation", "canopy_temperature",
              "bareground_fluxes", "canopy_fluxes", "soil_temperature", "surface_fluxes")  # + the conservation diagnostics at the end
# (snow_hydrology: the reference's functions run seven of its ten stages, tests/test_oracle_vs_ref.py; not the whole wrapper)


def start_state():
    from elmkernels_amd import synth
    from tests import helpers as H

    ft = H.field_table_from_oracle()
    cols, scal, soil = synth.make_state(ft, N, tier=TIER, seed=SEED)
    return (cols, scal, soil), H.oracle_state(cols, scal, soil)


def state_hash(S):
    h = hashlib.sha256()
    for k in sorted(S.fields):
        if k != "err_flags":
            h.update(np.ascontiguousarray(S.fields[k]).tobytes())
    return h.hexdigest()[:24]


def run_oracle(S, stage):
    {"init_timestep": S.init_timestep, "frac_wet": S.frac_wet, "albedo_snicar": S.albedo_snicar,
     "canopy_hydrology": lambda: S.canopy_hydrology(DT), "surface_radiation": S.surface_radiation,
     "canopy_temperature": S.canopy_temperature, "bareground_fluxes": S.bareground_fluxes,
     "canopy_fluxes": lambda: S.canopy_fluxes(DT), "soil_temperature": lambda: S.soil_temperature(DT),
     "snow_hydrology": lambda: S.snow_hydrology(DT), "surface_fluxes": lambda: S.surface_fluxes(DT)}[stage]()


def run_reference(S, stage):
    from oracle import oracle as O

    R = O.Reference()
    {"init_timestep": lambda: S.init_timestep(lib=R.R), "frac_wet": lambda: R.frac_wet(S),
     "canopy_hydrology": lambda: R.canopy_hydrology(S, DT), "surface_radiation": lambda: R.surface_radiation(S),
     "canopy_temperature": lambda: R.canopy_temperature(S), "bareground_fluxes": lambda: R.bareground_fluxes(S),
     "albedo_snicar": S.albedo_snicar_ref, "canopy_fluxes": lambda: S.canopy_fluxes_ref(DT),
     "soil_temperature": lambda: S.soil_temperature_ref(DT), "surface_fluxes": lambda: S.surface_fluxes(DT, lib=R.R)}[stage]()


def changed_fields(before, after):
    out = {}
    for k, a in after.fields.items():
        if k == "err_flags":
            continue
        b = before.fields[k]
        same = (a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float)))
        if not same.all():
            out[k] = a.copy()
    return out


def generate():
    from oracle import oracle as O

    _, S = start_state()
    data = {"meta/n": np.int64(N), "meta/seed": np.int64(SEED)}
    R = O.Reference()
    for stage in STAGES:
        data[f"hash/{stage}"] = np.array(state_hash(S))
        if stage in REF_STAGES:
            T = S.clone()
            T["err_flags"][...] = 0
            run_reference(T, stage)
            assert not (T["err_flags"] >> 31).any(), f"the reference threw in {stage}"
            for k, v in changed_fields(S, T).items():
                data[f"out/{stage}/{k}"] = v
        if stage == "albedo_snicar":
            # the SNICAR products of both passes on their own as well (the wrapper drops the per-layer absorbed-flux factors
            # flx_absd_snw / flx_absi_snw after folding them into flx_abs*): on the state in which the wrapper left albsoi / albsod
            T = S.clone()
            T.albedo_snicar()
            T["albsnd"][:] = -1.0
            T["albsni"][:] = -1.0
            fd, fi = R.snicar(T)
            data["out/snicar/albsnd"], data["out/snicar/albsni"] = T["albsnd"].copy(), T["albsni"].copy()
            data["out/snicar/flx_absd_snw"], data["out/snicar/flx_absi_snw"] = fd, fi
        run_oracle(S, stage)
    data["hash/evaluate_conservation"] = np.array(state_hash(S))
    data["out/evaluate_conservation/diag"] = S.clone().evaluate_conservation(DT, lib=R.R)
    np.savez_compressed(PATH, **data)
    print(PATH, os.path.getsize(PATH) // 1024, "KiB;", sum(1 for k in data if k.startswith("out/")), "output arrays")


def load():
    return np.load(PATH, allow_pickle=False)


if __name__ == "__main__":
    generate()
