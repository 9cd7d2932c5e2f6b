// Wave-synchronous dense factorisation of one small SPD matrix per wavefront,
// with each lane's row (or column) held in REGISTERS and the finished factor
// rows broadcast from LDS.
//
// Why this shape on CDNA4: a T x T (T <= 64) fp64 Cholesky done by one wave out
// of LDS alone needs two LDS reads per FMA (own row + pivot row); the own-row
// reads are 512 B/instruction and saturate the CU's LDS port long before the
// fp64 pipe (measured: 1.15 ms per 20 000 50x50 factorisations).  With the loop
// fully unrolled (T is a template parameter) every register index is static,
// the own row lives in 2T VGPRs, and the only LDS traffic left is the pivot
// row, read with one wave-uniform (broadcast) ds_read_b128 per two FMAs.
//
// Packed storage: row i of a lower-triangular factor holds i+1 entries padded
// to an even count so that every row starts 16-byte aligned.
#pragma once
#include <hip/hip_runtime.h>

__host__ __device__ constexpr int tri_row_off(int i) {
    // sum_{q<i} ((q + 2) & ~1)
    return (i & 1) ? 2 * (i / 2) * (i / 2) + 4 * (i / 2) + 2 : 2 * (i / 2) * (i / 2) + 2 * (i / 2);
}
__host__ __device__ constexpr int tri_packed_size(int T) { return tri_row_off(T); }

__device__ __forceinline__ void tri_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// Intra-wave LDS hand-off: the LDS executes one wavefront's DS instructions in issue order,
// so a lane's ds_write is visible to a later ds_read of any lane of the SAME wave without
// waiting for the write to be acknowledged.  Only the compiler must keep the order.
__device__ __forceinline__ void tri_wave_order() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double tri_readlane(double x, int src_lane) {
    union { double d; int i[2]; } u, o;
    u.d = x;
    o.i[0] = __builtin_amdgcn_readlane(u.i[0], src_lane);
    o.i[1] = __builtin_amdgcn_readlane(u.i[1], src_lane);
    return o.d;
}

// 1 / sqrt(d) and sqrt(d) for d > 0 from v_rsq_f64 + two Newton steps (about a
// third of the dependent-instruction depth of sqrt() followed by a division).
__device__ __forceinline__ void tri_rsqrt(double d, double* inv_out, double* sd_out) {
    double y = __builtin_amdgcn_rsq(d);
    double e = fma(-d * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-d * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    double sd = d * y;
    sd = fma(fma(-sd, sd, d), 0.5 * y, sd);
    *inv_out = y;
    *sd_out = sd;
}

// In: Lp (LDS, packed lower) holds A: row `lane` at tri_row_off(lane).  Out: Lp
// holds chol(A) in place; r[i] = L[lane][i] for i < lane.  Returns false on a
// non-positive pivot.
// Lanes >= T compute garbage, write nothing.
template <int T>
__device__ __forceinline__ bool wave_chol_rows(double (&r)[T], double* Lp, int lane, double* invd = nullptr) {
    const int my_off = tri_row_off(lane < T ? lane : 0);
    int bad = 0;  // eager failure flag (see wave_chol_rows_duo)
#pragma unroll
    for (int k = 0; k < T; ++k) {  // no early exit: a `break` would defeat the full unroll
        const double* Lk = Lp + tri_row_off(k);
        double s0 = Lp[my_off + k], s1 = 0.0;  // A[lane][k]; garbage (in bounds) for lane < k
#pragma unroll
        for (int i = 0; i + 1 < k; i += 2) {
            const double2 v = *reinterpret_cast<const double2*>(Lk + i);
            s0 = fma(-r[i], v.x, s0);
            s1 = fma(-r[i + 1], v.y, s1);
        }
        if (k & 1) s0 = fma(-r[k - 1], Lk[k - 1], s0);
        const double s = s0 + s1;
        const double d = tri_readlane(s, k);
        bad |= (!(d > 0.0) || !(d < 1e300)) ? 1 : 0;  // later columns are garbage; caller discards
        asm volatile("" : "+v"(bad));
        double inv, sd;
        tri_rsqrt(d, &inv, &sd);  // one reciprocal root per column instead of sqrt + a division per lane
        r[k] = (lane == k) ? sd : s * inv;
        if (lane >= k && lane < T) Lp[my_off + k] = r[k];
        if (invd && lane == k) invd[k] = inv;  // reciprocal diagonal for the inverse
        tri_wave_sync();
        __builtin_amdgcn_sched_barrier(0);  // one scheduling region per column: bounded live ranges
    }
    return bad == 0;
}

// In: Lp = packed lower-triangular L.  Out: x[i] = (L^-1)[i][lane], i.e. lane c holds COLUMN c of X = L^-1 (zeros
// above the diagonal).
template <int T>
__device__ __forceinline__ void wave_tri_inverse_cols(const double* Lp, double (&x)[T], int lane,
                                                      const double* invd = nullptr) {
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const double* Li = Lp + tri_row_off(i);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j + 1 < i; j += 2) {
            const double2 v = *reinterpret_cast<const double2*>(Li + j);
            a0 = fma(v.x, x[j], a0);
            a1 = fma(v.y, x[j + 1], a1);
        }
        if (i & 1) a0 = fma(Li[i - 1], x[i - 1], a0);
        // NB: a true division here, not a multiplication by a stored reciprocal:
        // with the cheap form hipcc (ROCm 7.2) hoists far more of the unrolled
        // rows' address/mask setup and spills ~230 VGPRs (measured 3x slower).
        const double rhs = (lane == i) ? 1.0 : 0.0;
        if (invd) x[i] = (rhs - (a0 + a1)) * invd[i];
        else x[i] = (rhs - (a0 + a1)) / Li[i];
        // keep the scheduler from hoisting later rows' LDS loads (register pressure)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Register-only Cholesky + triangular inverse for small matrices (RP <= 16):
// lane j (and every lane congruent to j mod RP) holds the full symmetric row j in
// a[0..RP-1]; pivot-row entries are fetched with v_readlane (static lane), so there
// is no LDS traffic and no wave-level synchronisation at all.  Out: x[i] =
// (L^-1)[i][j], the column j of X.  Returns false on a non-positive pivot.
template <int RP>
__device__ __forceinline__ bool wave_chol_inv_regs(double (&a)[RP], double (&x)[RP], int j, int r) {
    // r (wave-uniform, <= RP) = true dimension; rows/columns >= r are the identity padding and
    // are skipped: the unrolled steps beyond r cost one scalar branch each.
    // the diagonal slot a[k] of lane k keeps 1 / L[k][k]: the recurrence never reads the
    // diagonal again and the inverse wants the reciprocal (fetched with v_readlane)
    bool ok = true;
#pragma unroll
    for (int k = 0; k < RP; ++k) {
        if (k < r) {
            double s0 = a[k], s1 = 0.0;
#pragma unroll
            for (int i = 0; i < k; ++i) {
                const double lki = tri_readlane(a[i], k);  // L[k][i]
                if (i & 1) s1 = fma(-a[i], lki, s1);
                else s0 = fma(-a[i], lki, s0);
            }
            const double s = s0 + s1;
            const double d = tri_readlane(s, k);
            if (!(d > 0.0) || !(d < 1e300)) ok = false;
            double inv, sd;
            tri_rsqrt(d, &inv, &sd);
            a[k] = (j == k) ? inv : s * inv;  // L[j][k] for j > k (garbage above the diagonal)
        }
    }
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        if (i < r) {
            double c0 = 0.0, c1 = 0.0;
#pragma unroll
            for (int q = 0; q < i; ++q) {
                const double liq = tri_readlane(a[q], i);  // L[i][q]
                if (q & 1) c1 = fma(liq, x[q], c1);
                else c0 = fma(liq, x[q], c0);
            }
            x[i] = ((j == i ? 1.0 : 0.0) - (c0 + c1)) * tri_readlane(a[i], i);
        } else {
            x[i] = (j == i) ? 1.0 : 0.0;
        }
    }
    return ok;
}

// Upper-packed storage of X' (row c = column c of X, entries k = c..T-1, padded
// to an even count): offset of row c.
__host__ __device__ constexpr int triu_row_off(int c, int T) {
    // sum_{q<c} ((T - q + 1) & ~1)
    int s = 0;
    for (int q = 0; q < c; ++q) s += (T - q + 1) & ~1;
    return s;
}

// ===========================================================================
// "Lean" duo variants: the task's LDS footprint is the packed triangle and nothing else
// (T (T + 1) / 2 doubles, rows NOT padded), so that four two-wave blocks -- eight waves --
// share a CU's 160 KB at T = 50 (the padded layout plus per-task vectors stops at six;
// measured: the factorisations are latency-bound and gain 18 % from the two extra waves).
//   * A = I + S K S is never stored: column k of A is produced on the fly from the lane's
//     s_row, the pivot's s_k (static-lane pick) and a sliding window of the Toeplitz first
//     column kept in registers: lane q holds kv[q - k] (row q) and kv[q + T/2 - k]; after
//     each step both windows move one lane up (DPP wave_shr:1, no LDS), row T/2's window
//     taking over the value that falls off row T/2 - 1.
//   * 1 / L[k][k] lives in the diagonal slot of the packed factor (the diagonal itself is
//     never read back): no reciprocal-diagonal array.
//   * the dK-weighted trace uses the symmetry of the double sum (2 x the strictly lower
//     part), so its weights dK[row - j] are one more sliding window; X' is stored
//     pre-scaled by s_j.
// Unpadded rows are only 8-byte aligned: pivot-row pairs are read as two adjacent doubles
// (one ds_read2_b64).
// ===========================================================================
__host__ __device__ constexpr int tri_off_u(int i) { return i * (i + 1) / 2; }
__host__ __device__ constexpr int triu_off_u(int c, int T) { return c * T - (c * (c - 1)) / 2; }

// offset of row c in the upper-packed X' storage (rows padded to an even length), T even
__host__ __device__ constexpr int triu_off_even(int c, int T) {
    return (c & 1) ? 2 * T * (c / 2) - 2 * (c / 2) * (c / 2 - 1) + (T - 2 * (c / 2))
                   : 2 * T * (c / 2) - 2 * (c / 2) * (c / 2 - 1);
}
