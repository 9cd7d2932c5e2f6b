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

// sum_k log L[k][k] of a packed factor: one log per lane, then a wave reduction
template <int T>
__device__ __forceinline__ double wave_tri_logdet(const double* Lp, int lane) {
    double v = lane < T ? log(Lp[tri_row_off(lane < T ? lane : 0) + (lane < T ? lane : 0)]) : 0.0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
template <int T>
struct TriuOff {
    int v[T + 1];
    constexpr TriuOff() : v() {
        int s = 0;
        for (int q = 0; q <= T; ++q) {
            v[q] = s;
            s += (T - q + 1) & ~1;
        }
    }
};

// Store column `lane` of X (registers) as packed row `lane` of X' in LDS.
template <int T>
__device__ __forceinline__ void wave_store_cols(const double (&x)[T], double* Xp, int lane) {
    constexpr TriuOff<T> off{};
    int my = 0;
#pragma unroll
    for (int q = 0; q < T; ++q)
        if (q == lane) my = off.v[q];
    if (lane < T) {
#pragma unroll
        for (int k = 0; k < T; ++k)
            if (k >= lane) Xp[my + k - lane] = x[k];
        if ((T - lane) & 1) Xp[my + T - lane] = 0.0;  // zero the pad so pair reads are exact
    }
    tri_wave_sync();
}

// ===========================================================================
// "Duo" variants: ONE wavefront factors TWO matrices (task A in lanes 0..31,
// task B in lanes 32..63) and every lane holds TWO rows of its matrix (row q and
// row q + T/2, q = lane & 31).  Why: a wave-uniform pivot value read from LDS
// serves one FMA per lane in the one-row scheme; the CU's single LDS pipe (one
// broadcast ds_read_b128 = 4 LDS cycles for 2 doubles) then caps the four SIMDs
// at half their fp64 FMA rate (measured: 53 % issue utilisation).  With two rows
// per lane each broadcast value feeds two FMAs, the two halves read two
// addresses per instruction at no extra LDS cost, and the static upper-triangle
// work of rows < T/2 disappears for steps k >= T/2 (762 instead of 1225 FMA
// instructions per matrix and phase).
// ===========================================================================
__device__ __forceinline__ double tri_pick_half(double v, int src_q, int h) {
    // value of lane (32 h + src_q): two static readlanes and a per-half select
    const double a = tri_readlane(v, src_q);
    const double b = tri_readlane(v, 32 + src_q);
    return h ? b : a;
}

// In: Lp = this lane's task buffer (packed lower, holds A).  Out: chol(A) in place,
// r0[i] = L[q][i], r1[i] = L[q + T/2][i]; invd[k] = 1 / L[k][k] (per-task LDS array).
// Returns per-lane ok flag (identical within a half).
template <int T>
__device__ __forceinline__ bool wave_chol_rows_duo(double (&r0)[T / 2], double (&r1)[T], double* Lp, double* invd,
                                                   int q, int h) {
    constexpr int H = T / 2;
    const bool in = q < H;
    const int off0 = tri_row_off(in ? q : 0), off1 = tri_row_off(in ? q + H : H);
    int bad = 0;  // materialised every step (asm below): a lazily evaluated `ok &= d > 0` chain
                  // keeps all T pivots alive to the end (measured: +74 VGPRs)
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const double* Lk = Lp + tri_row_off(k);
        double s0 = (k < H) ? Lp[off0 + (k < H ? k : 0)] : 0.0, s0b = 0.0;  // A[q][k]
        double s1 = Lp[off1 + k], s1b = 0.0;                              // A[q + H][k]
#pragma unroll
        for (int i = 0; i + 1 < k; i += 2) {
            const double2 v = *reinterpret_cast<const double2*>(Lk + i);
            if (k < H) {  // rows < H are finished once k >= H: no work for them (static)
                s0 = fma(-r0[i < H ? i : 0], v.x, s0);
                s0b = fma(-r0[i + 1 < H ? i + 1 : 0], v.y, s0b);
            }
            s1 = fma(-r1[i], v.x, s1);
            s1b = fma(-r1[i + 1], v.y, s1b);
            // bound the number of pivot-row loads in flight (registers): at most 8 b128 ahead
            if ((i & 14) == 14) {
                asm volatile("" : "+v"(s0), "+v"(s0b), "+v"(s1), "+v"(s1b) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (k & 1) {
            const double lv = Lk[k - 1];
            if (k < H) s0 = fma(-r0[k - 1 < H ? k - 1 : 0], lv, s0);
            s1 = fma(-r1[k - 1], lv, s1);
        }
        s0 += s0b;
        s1 += s1b;
        const double d = tri_pick_half(k < H ? s0 : s1, k < H ? k : k - H, h);  // pivot of this half's matrix
        bad |= (!(d > 0.0) || !(d < 1e300)) ? 1 : 0;
        asm volatile("" : "+v"(bad));
        double inv, sd;
        tri_rsqrt(d, &inv, &sd);
        if (k < H) {
            r0[k < H ? k : 0] = (q == k) ? sd : s0 * inv;
            if (in && q >= k) Lp[off0 + k] = r0[k < H ? k : 0];
        }
        r1[k] = (q + H == k) ? sd : s1 * inv;
        if (in && q + H >= k) Lp[off1 + k] = r1[k];
        if (q == 0) invd[k] = inv;
        tri_wave_order();
        __builtin_amdgcn_sched_barrier(0);
    }
    return bad == 0;
}

// Columns q and q + T/2 of X = L^-1 in registers: x0[i] = X[i][q], x1[i - T/2] = X[i][q + T/2]
// (the latter is zero for i < T/2 and not stored).
template <int T>
__device__ __forceinline__ void wave_tri_inverse_cols_duo(const double* Lp, const double* invd, double (&x0)[T],
                                                          double (&x1)[T / 2], int q) {
    constexpr int H = T / 2;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const double* Li = Lp + tri_row_off(i);
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
        for (int j = 0; j + 1 < i; j += 2) {
            const double2 v = *reinterpret_cast<const double2*>(Li + j);
            a0 = fma(v.x, x0[j], a0);
            a1 = fma(v.y, x0[j + 1], a1);
            if (j >= H) b0 = fma(v.x, x1[j - H < 0 ? 0 : j - H], b0);
            if (j + 1 >= H) b1 = fma(v.y, x1[j + 1 - H < 0 ? 0 : j + 1 - H], b1);
            if ((j & 14) == 14) {
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (i & 1) {
            const double lv = Li[i - 1];
            a0 = fma(lv, x0[i - 1], a0);
            if (i - 1 >= H) b0 = fma(lv, x1[i - 1 - H < 0 ? 0 : i - 1 - H], b0);
        }
        const double di = invd[i];
        x0[i] = ((q == i ? 1.0 : 0.0) - (a0 + a1)) * di;
        if (i >= H) x1[i - H < 0 ? 0 : i - H] = ((q + H == i ? 1.0 : 0.0) - (b0 + b1)) * di;
        __builtin_amdgcn_sched_barrier(0);
    }
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

__device__ __forceinline__ double tri_wave_shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);  // wave_shr:1 (lane i <- lane i - 1)
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Moves both sliding windows one step: w0 (row q) <- lane q - 1, w1 (row q + H) <- lane q - 1, and lane
// q = 0 of each half takes row H's next value from row H - 1 (w0 of lane H - 1), `fill0` for row 0.
template <int H>
__device__ __forceinline__ void tri_windows_step(double& w0, double& w1, int q, int h, double fill0) {
    const double carry = tri_pick_half(w0, H - 1, h);
    w0 = tri_wave_shr1(w0);
    w1 = tri_wave_shr1(w1);
    if (q == 0) {
        w0 = fill0;
        w1 = carry;
    }
}

// Factor A = I + S K S without ever storing A.  sw0/sw1: sqrt(w) of rows q, q + T/2; kvw0/kvw1:
// jitter-free first column of K, sigma^2 exp(-omega (d dt)^2), at distances d = q and q + T/2;
// g0/g1 = exp(omega dt^2 (2 d - 1)) at those distances, gc = exp(-2 omega dt^2): the column obeys
// kv0[d-1] = kv0[d] g_d with g_{d-1} = g_d gc, so every lane walks its own two values down to
// distance zero (and symmetrically beyond) with two multiplies a step -- fifty steps cost ~1e-14
// relative, far inside the parity tolerance; eps = the jitter on the diagonal of K.  The diagonal slot of
// row k holds s_k on entry (the caller puts it there) and 1 / L[k][k] on exit; L strictly below the
// diagonal (unpadded packed).  Unpadded rows start at even or odd offsets: the pivot-row pairs are
// formed from index 0 or 1 accordingly (static), so that every pair is one aligned ds_read_b128
// with an immediate offset.
template <int T>
__device__ __forceinline__ bool wave_chol_rows_duo_lean(double (&r0)[T / 2], double (&r1)[T], double* Lp, int q, int h,
                                                        double sw0, double sw1, double kvw0, double kvw1,
                                                        double g0, double g1, double gc, double eps) {
    constexpr int H = T / 2;
    const bool in = q < H;
    const int off0 = tri_off_u(in ? q : 0), off1 = tri_off_u(in ? q + H : H);
    int bad = 0;
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const double* Lk = Lp + tri_off_u(k);
        const int st = tri_off_u(k) & 1;  // first index of the aligned pairs
        const double sk = Lk[k];          // s_k, parked in the diagonal slot
        double s0 = 0.0, s0b = 0.0, s1b = 0.0;
        // A[row][k] = s_row s_k kv0[row - k] + (row == k) (1 + s_k^2 eps)
        if (k < H) s0 = fma(sw0 * sk, kvw0, q == k ? fma(sw0 * sk, eps, 1.0) : 0.0);
        double s1 = fma(sw1 * sk, kvw1, q + H == k ? fma(sw1 * sk, eps, 1.0) : 0.0);
        if (st && k > 0) {
            const double lv = Lk[0];
            if (k < H) s0 = fma(-r0[0], lv, s0);
            s1 = fma(-r1[0], lv, s1);
        }
#pragma unroll
        for (int i = st; i + 1 < k; i += 2) {
            const double2 v = *reinterpret_cast<const double2*>(Lk + i);
            if (k < H) {
                s0 = fma(-r0[i < H ? i : 0], v.x, s0);
                s0b = fma(-r0[i + 1 < H ? i + 1 : 0], v.y, s0b);
            }
            s1 = fma(-r1[i], v.x, s1);
            s1b = fma(-r1[i + 1], v.y, s1b);
            if (((i - st) & 14) == 14) {
                asm volatile("" : "+v"(s0), "+v"(s0b), "+v"(s1), "+v"(s1b) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (k > st && ((k - st) & 1)) {
            const double lv = Lk[k - 1];
            if (k < H) s0 = fma(-r0[k - 1 < H ? k - 1 : 0], lv, s0);
            s1 = fma(-r1[k - 1], lv, s1);
        }
        s0 += s0b;
        s1 += s1b;
        const double d = tri_pick_half(k < H ? s0 : s1, k < H ? k : k - H, h);
        bad |= (!(d > 0.0) || !(d < 1e300)) ? 1 : 0;
        asm volatile("" : "+v"(bad));
        double inv, sd;
        tri_rsqrt(d, &inv, &sd);
        if (k < H) {
            r0[k < H ? k : 0] = s0 * inv;
            if (in && q >= k) Lp[off0 + k] = (q == k) ? inv : r0[k < H ? k : 0];
        }
        r1[k] = s1 * inv;
        if (in && q + H >= k) Lp[off1 + k] = (q + H == k) ? inv : r1[k];
        // next distance: kv0[d - 1] = kv0[d] g_d, g_{d-1} = g_d e^{-2 omega dt^2} (each lane advances its own
        // two window values; no cross-lane traffic)
        if (k < H) {
            kvw0 *= g0;
            g0 *= gc;
        }
        kvw1 *= g1;
        g1 *= gc;
        asm volatile("" : "+v"(kvw0), "+v"(kvw1), "+v"(g0), "+v"(g1));
        tri_wave_order();
        __builtin_amdgcn_sched_barrier(0);
    }
    return bad == 0;
}

// Columns q and q + T/2 of X = L^-1 from the lean factor (reciprocal diagonal in place).
template <int T>
__device__ __forceinline__ void wave_tri_inverse_cols_duo_lean(const double* Lp, double (&x0)[T], double (&x1)[T / 2],
                                                               int q) {
    constexpr int H = T / 2;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const double* Li = Lp + tri_off_u(i);
        const int st = tri_off_u(i) & 1;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        if (st && i > 0) {
            const double lv = Li[0];
            a0 = fma(lv, x0[0], a0);
        }
#pragma unroll
        for (int j = st; j + 1 < i; j += 2) {
            const double2 v = *reinterpret_cast<const double2*>(Li + j);
            a0 = fma(v.x, x0[j], a0);
            a1 = fma(v.y, x0[j + 1], a1);
            if (j >= H) b0 = fma(v.x, x1[j - H < 0 ? 0 : j - H], b0);
            if (j + 1 >= H) b1 = fma(v.y, x1[j + 1 - H < 0 ? 0 : j + 1 - H], b1);
            if (((j - st) & 14) == 14) {
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (i > st && ((i - st) & 1)) {
            const double lv = Li[i - 1];
            a0 = fma(lv, x0[i - 1], a0);
            if (i - 1 >= H) b0 = fma(lv, x1[i - 1 - H < 0 ? 0 : i - 1 - H], b0);
        }
        const double di = Li[i];  // 1 / L[i][i]
        x0[i] = ((q == i ? 1.0 : 0.0) - (a0 + a1)) * di;
        if (i >= H) x1[i - H < 0 ? 0 : i - H] = ((q + H == i ? 1.0 : 0.0) - (b0 + b1)) * di;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// offset of row c in the upper-packed X' storage (rows padded to an even length), T even
__host__ __device__ constexpr int triu_off_even(int c, int T) {
    return (c & 1) ? 2 * T * (c / 2) - 2 * (c / 2) * (c / 2 - 1) + (T - 2 * (c / 2))
                   : 2 * T * (c / 2) - 2 * (c / 2) * (c / 2 - 1);
}
