// tb_forms.hpp — kernel-argument views of a mesh and of a scalar form (shared by tb_assembly.hip and tb_patch_fused.hip).
#pragma once
#include <cmath>
#include <cstdlib>

#include "tb_internal.h"

namespace tb {

struct MeshView {
    const double *xyz;
    const int32_t *conn;
    const int32_t *cell_dofs;
    int64_t n_cells;
};

struct FormArgs {
    double D[9];      // constant tensor (already divided by Cₘ·χ when wrapped)
    double rho;       // constant density
    double lambda[3]; // eigenvalues for the spectral field coefficient
    double scale;     // 1/(Cₘ·χ) for field coefficients (1 when not wrapped)
    const double *field;
    const double *gtab; // hexahedron patch kernel with a field tensor: 7 doubles per (cell, q): G_q (xx,xy,xz,yy,yz,zz) and detJ_q
    const double *dtab; // diffusion with a fibre field: the tensor at every quadrature point, 6 doubles (xx,xy,xz,yy,yz,zz) per (cell, q)
    // source
    int src_kind;
    double p0;
    const double *table;
    double t, ct;
    const double *tslot; // non-NULL while a graph capture is open: {t, cos 2πt} on the device replace the two scalars above (tb_graph.hip)
    // constant positive definite tensor D = L·Lᵀ, record kernel: coordinates are mapped by L⁻¹ at staging (row-major Linv), iso_scale = −¼·det L
    double Linv[9], iso_scale;
#ifdef TB_ABLATION
    int debug; // profiling builds only (make ABLATION=1 → libtbhip_ablation.so): TB_DEBUG_FLAGS bits 1 skip LDS adds, 2 skip write-out, 4 skip arithmetic
#endif
};

inline FormArgs make_args(const tb_form *f, double t)
{
    FormArgs a{};
    for (int i = 0; i < 9; ++i) a.D[i] = f->Dconst[i];
    a.rho = f->coef.p[0];
    a.lambda[0] = f->coef.p[0]; a.lambda[1] = f->coef.p[1]; a.lambda[2] = f->coef.p[2];
    a.scale = f->coef.wrap ? 1.0 / (f->coef.Cm * f->coef.chi) : 1.0;
    a.field = f->d_field;
    a.dtab = f->d_dtab;
    a.gtab = f->d_gtab;
    a.src_kind = f->coef.kind;
    a.p0 = f->coef.p[0];
    a.table = f->d_table;
    a.t = t;
    a.ct = std::cos(2.0 * 3.141592653589793 * t);
    a.tslot = nullptr;
    if (f->mesh->dev->capturing && f->kind == TB_FORM_SOURCE) { // the one form kind whose integrand reads the time
        a.tslot = f->mesh->dev->d_tslot;
        f->mesh->dev->tslot_used = true;
    }
#ifdef TB_ABLATION
    static const int dbg = getenv("TB_DEBUG_FLAGS") ? atoi(getenv("TB_DEBUG_FLAGS")) : 0;
    a.debug = dbg;
#endif
    return a;
}

inline MeshView make_view(const tb_mesh *m) { return MeshView{m->d_xyz, m->d_conn, m->d_cell_dofs, m->n_cells}; }


} // namespace tb
