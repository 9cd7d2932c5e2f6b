// Kernel argument block shared by the generic (estep.hip) and the fast
// (estep_fast.hip) E-step kernels.
#pragma once
#include "ctx.h"

struct EstepArgs {
    int N, L;
    int mode, n_iter, vb;
    double dmu_bound;
    const int64_t* off;
    const int* unit_prior;
    const double* const* prior_base;
    const int* prior_rl;
    const int64_t* prior_goff;
    const double* y;
    const double* xb;  // (rows, N) or null when x == 1
    double* mu;
    double* v;
    double* w;
    double* dmu;
    const double* a;
    const double* b;
    const double* noise;
    const int* gauss;
    double* scratch;       // long units: 3 * rows * L doubles (ra, ya, u)
    double* lc_global;     // long units whose factors do not fit LDS (else null)
    int64_t lc_stride;     // doubles per unit in lc_global
    int* fail;
    int rg;                // lanes per row in the (T x N) passes (power of two <= 64)
    int lds_T;             // SMALL: row capacity of the LDS tiles
    int lds_gsz, lds_lcsz; // doubles reserved for G tiles / factors in LDS
    int lds_scr;           // fast kernel: doubles of the shared scratch region
    unsigned long long* clk; // optional per-phase cycle counters (thread 0 of every block), or null
    const double* cols_g;    // fast kernel: per-channel records (a_l, a_l^2, b, 1/noise), (N, 2*LT+2), global
    const double* wconst_g;  // fast kernel: Gaussian-channel constant of w per latent (L)
};


// estep_fast.hip: sets *handled = 1 and launches when the fast kernel applies
// (T <= 64, every effective rank <= 32, L <= 8, LDS fits), else leaves 0.
int launch_estep_fast(vlgp_ctx* ctx, UnitSet& us, EstepArgs A, int* handled);

// estep_long.hip: long units (T > 64) with every wave of the workgroup on the per-latent phases;
// declines (leaves *handled = 0) when rank > 50, L > 10 or the LDS budget does not fit.
int launch_estep_long(vlgp_ctx* ctx, UnitSet& us, EstepArgs A, int* handled);

// estep_split.hip: many window-sized units as a sequence of chip-wide launches (passes over rows, one wave per
// (unit, latent) for the factor and mean phases); declines for small sets, T > 64, rank > 32 or L > 10.
int launch_estep_split(vlgp_ctx* ctx, UnitSet& us, EstepArgs A, int* handled);
