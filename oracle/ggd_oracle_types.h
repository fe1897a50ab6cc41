/* ggd_oracle_types.h -- TEST INFRASTRUCTURE (CPU oracle): the parameter block shared by ggd_oracle.c and
 * ggd_oracle_bound.cpp.  Mirrors the scalar fields of the reference's GaussianRasterizationSettings
 * (gaussian_splatting/gaussian_renderer/__init__.py:38-51). */
#ifndef GGD_ORACLE_TYPES_H
#define GGD_ORACLE_TYPES_H
#include <stdint.h>
typedef struct ggo_params {
  int32_t P, M, D, W, H;
  int32_t prefiltered;
  double tanfovx, tanfovy, scale_modifier;
} ggo_params;
#endif
