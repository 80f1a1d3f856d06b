// tb_mech_split.hpp — launchers of tb_mech_split.hip
#pragma once
#include "tb_energy.hpp"
#include "tb_mech_common.hpp"

namespace tb {
constexpr int QP_SYM = 45, QP_ENT = QP_SYM + 9, QP_REC = QP_ENT * 27; // entries per quadrature point (symmetric Â + P̂), doubles per cell record
int launch_mech_points(tb_device *dev, const MechMesh &mm, const HOParams &hp, const EnergyParams *ep /*NULL: hand-derived Holzapfel–Ogden; else device AD of that energy*/,
                       const double *d_u, int64_t n_cells, double *d_qp); // cells [mm.cell0, mm.cell0 + n)
int launch_mech_contract(tb_device *dev, const double *d_qp, int64_t cell0, int64_t n_cells, double *d_ke, double *d_re /*nullable*/,
                         const uint8_t *d_rank27 /*non-NULL: Kₑ symmetric-packed by these per-cell ranks (32 bytes per cell); NULL: 81 × 81 in tensor order*/);
} // namespace tb
