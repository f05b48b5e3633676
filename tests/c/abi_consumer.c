/* A plain C99 consumer of include/elmk.h (tests/test_host_side.py compiles it with gcc -std=c99 -pedantic -Werror
 * -fsyntax-only): INTEGRATION.md promises the header is plain C, and the only other consumer in the tree is ctypes. */
#include <stddef.h>
#include <stdio.h>

#include "elmk.h"

int drive_one_step(int64_t ncols, const double *t_soisno_cols /* [ncols][20], the reference's layout */, double dt)
{
  elmk_ctx *ctx = NULL;
  uint32_t flags = 0;
  int64_t first_bad = -1;
  int rc = elmk_create(ncols, 0, &ctx);
  if (rc != ELMK_OK) {
    fprintf(stderr, "elmk_create: %s\n", elmk_last_error(NULL));
    return rc;
  }
  rc = elmk_set_land(ctx, 1, 1, 12, 0, 0);
  if (rc == ELMK_OK) rc = elmk_set_scalars(ctx, 0.1, 1, 43200.0, 86400.0);
  if (rc == ELMK_OK) rc = elmk_upload(ctx, ELMK_FIELD_t_soisno, t_soisno_cols, 0, ncols, ELMK_LAYOUT_COL_MAJOR);
  if (rc == ELMK_OK) rc = elmk_initialize_state(ctx);
  if (rc == ELMK_OK) rc = elmk_timestep7(ctx, dt);
  if (rc == ELMK_OK) rc = elmk_timestep7_fused(ctx, dt);
  if (rc == ELMK_OK) rc = elmk_soil_temperature(ctx, dt);
  if (rc == ELMK_OK) rc = elmk_snow_hydrology(ctx, dt);
  if (rc == ELMK_OK) rc = elmk_surface_fluxes(ctx, dt);
  if (rc == ELMK_OK) rc = elmk_advance_physics(ctx, dt);
  if (rc == ELMK_OK) rc = elmk_error_summary(ctx, &flags, &first_bad);
  if (rc == ELMK_OK && (flags & ELMK_ERR_FATAL_MASK)) fprintf(stderr, "column %ld raised %#x\n", (long)first_bad, (unsigned)flags);
  (void)elmk_destroy(ctx);
  return rc;
}
