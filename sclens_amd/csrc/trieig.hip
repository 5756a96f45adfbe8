// Symmetric tridiagonal eigenproblem (fp64) + back-transformation (fp32 MFMA), device-resident.
// Second and third phase of the replacement for the reference's `_get_eigen` (scLENS.jl:375-387).
//
//  * stebz_f64 : all eigenvalues by Sturm-count bisection, one thread per eigenvalue (ascending index;
//                the reference's eigen()/syevd! also return ascending order).
//  * stein_f64 : eigenvectors of selected indices by inverse iteration with a partially pivoted LU of
//                T - lambda I (the dlagtf/dlagts/dstein scheme, restated from the published LAPACK
//                algorithm), one thread per eigenvector, workspaces laid out [row][vector] so every
//                access is coalesced. fp64 keeps eps/gap small enough that no re-orthogonalisation
//                is needed for the simple spectra of this path (near-degenerate pairs are counted
//                and reported).
//  * ormtr_f32 : Z <- Q Z with Q = H_0 H_1 ... from sytrd_f32, as block reflectors (I - V T V^T)
//                applied with three fp32 MFMA GEMMs per panel; T comes from the V^T v products that
//                the tridiagonalisation already formed (no extra pass over V).
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace scl {

constexpr int NB = 128;  // must match tridiag.hip
constexpr double EPS64 = 2.220446049250313e-16;
constexpr double SFMIN64 = 2.2250738585072014e-308;

// info[0]=gl, [1]=gu, [2]=pivmin, [3]=onenrm, [4]=tnorm(max abs entry)
__global__ __launch_bounds__(1024) void tri_bounds(const double* __restrict__ d, const double* __restrict__ e,
                                                   int64_t n, double* __restrict__ e2, double* __restrict__ info) {
  __shared__ double s_lo[16], s_hi[16], s_e2[16], s_one[16], s_tn[16];
  __shared__ int s_nan;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) s_nan = 0;
  __syncthreads();
  double lo = 1e300, hi = -1e300, me2 = 0.0, one = 0.0, tn = 0.0;
  for (int64_t i = tid; i < n; i += 1024) {
    const double el = (i > 0) ? fabs(e[i - 1]) : 0.0, er = (i < n - 1) ? fabs(e[i]) : 0.0;
    const double di = d[i];
    if (!(fabs(di) + er < 1e300)) s_nan = 1;  // NaN / Inf in the tridiagonal: fmin / fmax below would hide it
    lo = fmin(lo, di - el - er);
    hi = fmax(hi, di + el + er);
    one = fmax(one, fabs(di) + el + er);
    tn = fmax(tn, fmax(fabs(di), er));
    if (i < n - 1) {
      const double q = e[i] * e[i];
      e2[i] = q;
      me2 = fmax(me2, q);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o));
    hi = fmax(hi, __shfl_xor(hi, o));
    me2 = fmax(me2, __shfl_xor(me2, o));
    one = fmax(one, __shfl_xor(one, o));
    tn = fmax(tn, __shfl_xor(tn, o));
  }
  if (lane == 0) { s_lo[wid] = lo; s_hi[wid] = hi; s_e2[wid] = me2; s_one[wid] = one; s_tn[wid] = tn; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) {
      lo = fmin(lo, s_lo[w]); hi = fmax(hi, s_hi[w]); me2 = fmax(me2, s_e2[w]);
      one = fmax(one, s_one[w]); tn = fmax(tn, s_tn[w]);
    }
    const double pivmin = SFMIN64 * fmax(1.0, me2);
    const double bn = fmax(fabs(lo), fabs(hi));
    info[0] = lo - 2.0 * bn * EPS64 * (double)n - 2.0 * pivmin;
    info[1] = hi + 2.0 * bn * EPS64 * (double)n + 2.0 * pivmin;
    info[2] = pivmin;
    info[3] = one;
    info[4] = tn;
    info[5] = s_nan ? 1.0 : 0.0;  // -> every eigenvalue is reported as NaN (the check of scLENS.jl:379 then fires)
  }
}

// Eight lanes per eigenvalue index k: 9-section on the Sturm count (#eigenvalues < x). Each sweep shrinks the
// bracket 9x (18 sweeps to full fp64 resolution instead of 57 bisection sweeps); the n-step recurrence is the
// latency chain, so trading 8x more lanes for 3x fewer sweeps is what a mostly idle chip wants.
__global__ __launch_bounds__(64) void tri_bisect(const double* __restrict__ d, const double* __restrict__ e2,
                                                 int64_t n, const double* __restrict__ info,
                                                 double* __restrict__ w, int64_t k_begin, int64_t k_end, int64_t k_extra) {
  // k_extra >= 0: one more block at the end of the grid finds that single index (the largest eigenvalue of a partial
  // spectrum) concurrently with the range -- as a launch of its own it costs a full latency chain of 18 x n steps
  const int q = threadIdx.x & 7;
  const bool extra = k_extra >= 0 && blockIdx.x == gridDim.x - 1;
  const int64_t k = extra ? k_extra + (threadIdx.x >> 3) : k_begin + (int64_t)blockIdx.x * 8 + (threadIdx.x >> 3);
  double lo = info[0], hi = info[1];
  const double pivmin = info[2], atol = EPS64 * info[4];
  const bool live = extra ? (threadIdx.x >> 3) == 0 : k < k_end;
  if (info[5] != 0.0) {  // non-finite input (block-uniform)
    if (live && q == 0) w[k] = __longlong_as_double(0x7ff8000000000000LL);
    return;
  }
  for (int it = 0; it < 40; ++it) {
    const bool done = (hi - lo) <= 2.0 * EPS64 * fmax(fabs(lo), fabs(hi)) + atol;
    if (__all(done || !live)) break;
    const double step = (hi - lo) / 9.0;
    const double xq = lo + (double)(q + 1) * step;
    double p = d[0] - xq;
    if (fabs(p) < pivmin) p = -pivmin;
    int cnt = (p < 0.0) ? 1 : 0;
    // e2 / p as e2 * (1/p): v_rcp_f64 refined by two Newton steps (|p| >= pivmin, so 1/p is finite) -- the full IEEE
    // division sequence is twice as long and this chain is the whole kernel; eight steps per trip so that the (uniform,
    // scalar) loads of d and e2 are issued as two wide loads ahead of the eight dependent steps
    auto step1 = [&](double di, double ei) {
      double r = __builtin_amdgcn_rcp(p);
      r = fma(fma(-p, r, 1.0), r, r);
      r = fma(fma(-p, r, 1.0), r, r);
      p = (di - xq) - ei * r;
      if (fabs(p) < pivmin) p = -pivmin;
      cnt += (p < 0.0) ? 1 : 0;
    };
    int64_t i = 1;
    for (; i + 8 <= n; i += 8) {
      double dd[8], ee[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        dd[u] = d[i + u];
        ee[u] = e2[i - 1 + u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) step1(dd[u], ee[u]);
    }
    for (; i < n; ++i) step1(d[i], e2[i - 1]);
    // m = number of probe points of this group with count <= k (eigenvalue k is not below them)
    int below = (cnt <= k) ? 1 : 0;
    int m = below;
    m += __shfl_xor(m, 1);
    m += __shfl_xor(m, 2);
    m += __shfl_xor(m, 4);
    if (!done) {
      const double nlo = (m == 0) ? lo : lo + (double)m * step;
      const double nhi = (m == 8) ? hi : lo + (double)(m + 1) * step;
      lo = nlo;
      hi = nhi;
    }
  }
  if (live && q == 0) w[k] = 0.5 * (lo + hi);
}

// ---- round 3: the same 9-section on the Sturm count, the count taken from the three-term recurrence of the leading principal
// minors p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2} (sign changes of p_{-1} = 1, p_0, p_1, ...) instead of the ratios
// q_i = p_i / p_{i-1}: no division. The ratio step is a v_rcp_f64, two Newton steps, the recurrence and two clamps -- ~150 issue
// cycles per step, and the kernel is bound by exactly that (two waves per SIMD, no stalls to hide); the product step is three
// fp64 operations, three integer ones for the sign change and a zero guard. Range: the matrix is scaled by a power of two to
// norm < 1 (tri_scale: exact), so |p_i| <= 3 max(|p_{i-1}|, |p_{i-2}|), and both running values are rescaled by a power of two
// every eight steps (exact). A minor that is exactly zero is replaced by a tiny value of the sign that makes it a sign change,
// which is what the ratio form's `pivmin` does (a decoupled block after an exact zero would otherwise see only zeros).
__global__ void tri_scale(const double* __restrict__ d, const double* __restrict__ e2, int64_t n, double* __restrict__ info,
                          double* __restrict__ ds, double* __restrict__ e2s) {
  const double bound = fmax(fabs(info[0]), fabs(info[1]));
  double sc = 1.0;
  if (bound > 0.0 && bound < 1.0e300) {
    int ex = 0;
    (void)frexp(bound, &ex);  // bound = f 2^ex, 1/2 <= f < 1
    sc = ldexp(1.0, -ex);
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ds[i] = d[i] * sc;
  if (i <= n) e2s[i] = e2[i] * sc * sc;
  if (i == 0) info[6] = sc;
}

__global__ __launch_bounds__(64) void tri_bisect_prod(const double* __restrict__ ds, const double* __restrict__ e2s, int64_t n,
                                                      const double* __restrict__ info, double* __restrict__ w, int64_t k_begin,
                                                      int64_t k_end, int64_t k_extra) {
  const int q = threadIdx.x & 7;
  const bool extra = k_extra >= 0 && blockIdx.x == gridDim.x - 1;
  const int64_t k = extra ? k_extra + (threadIdx.x >> 3) : k_begin + (int64_t)blockIdx.x * 8 + (threadIdx.x >> 3);
  const double sc = info[6];
  double lo = info[0] * sc, hi = info[1] * sc;
  const double atol = EPS64 * info[4] * sc;
  const bool live = extra ? (threadIdx.x >> 3) == 0 : k < k_end;
  if (info[5] != 0.0) {  // non-finite input (block-uniform)
    if (live && q == 0) w[k] = __longlong_as_double(0x7ff8000000000000LL);
    return;
  }
  const double tiny = 1.0e-290;
  for (int it = 0; it < 40; ++it) {
    const bool done = (hi - lo) <= 2.0 * EPS64 * fmax(fabs(lo), fabs(hi)) + atol;
    if (__all(done || !live)) break;
    const double step = (hi - lo) / 9.0;
    const double xq = lo + (double)(q + 1) * step;
    double p2 = 1.0, p1 = ds[0] - xq;
    if (p1 == 0.0) p1 = -tiny;
    int cnt = (p1 < 0.0) ? 1 : 0;
    auto step1 = [&](double di, double ei) {
      double pn = fma(di - xq, p1, -(ei * p2));
      if (pn == 0.0) pn = copysign(tiny, -p1);
      cnt += (int)(((unsigned)(__double_as_longlong(pn) >> 32) ^ (unsigned)(__double_as_longlong(p1) >> 32)) >> 31);
      p2 = p1;
      p1 = pn;
    };
    int64_t i = 1;
    for (; i + 8 <= n; i += 8) {
      double dd[8], ee[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        dd[u] = ds[i + u];
        ee[u] = e2s[i - 1 + u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) step1(dd[u], ee[u]);
      int ex = 0;
      (void)frexp(fmax(fabs(p1), fabs(p2)), &ex);
      p1 = ldexp(p1, -ex);
      p2 = ldexp(p2, -ex);
    }
    for (; i < n; ++i) step1(ds[i], e2s[i - 1]);
    int below = (cnt <= k) ? 1 : 0;
    int m = below;
    m += __shfl_xor(m, 1);
    m += __shfl_xor(m, 2);
    m += __shfl_xor(m, 4);
    if (!done) {
      const double nlo = (m == 0) ? lo : lo + (double)m * step;
      const double nhi = (m == 8) ? hi : lo + (double)(m + 1) * step;
      lo = nlo;
      hi = nhi;
    }
  }
  if (live && q == 0) w[k] = 0.5 * (lo + hi) / sc;
}

__device__ __forceinline__ double hash_uniform(uint64_t a, uint64_t b) {  // deterministic U(-1,1)
  uint64_t z = a * 0x9E3779B97F4A7C15ull + b * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

// Inverse iteration, one thread per eigenvector t (eigenvalue index idx0 + t).
// Workspaces are [n][B] (row i, vector t): element (i,t) at i*B + t, so every access is coalesced across threads.
// The recurrences are sequential in i; the loads of the next STEIN_PF steps are issued together to hide memory latency: the
// kernel moves ~230 B per (row, vector) in seven passes (100 GB for 15 008 vectors of order 30 016) from only m / 64 waves, so
// the bytes in flight set its speed (4 steps: 1.4 TB/s, 73 ms).
template <int STEIN_PF>
__global__ __launch_bounds__(64) void tri_stein(const double* __restrict__ d, const double* __restrict__ e,
                                                int64_t n, const double* __restrict__ w, int64_t idx0,
                                                int64_t count, int64_t B, const double* __restrict__ info,
                                                double* __restrict__ wa, double* __restrict__ wb,
                                                double* __restrict__ wc, double* __restrict__ wd,
                                                unsigned char* __restrict__ win, double* __restrict__ wx,
                                                double* __restrict__ inv_norm, int* __restrict__ fail_count, int good_its) {
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= count) return;
  const int64_t gi = idx0 + t;
  double lam = w[gi];
  const double onenrm = info[3];
  // separate numerically coincident shifts a little (dstein's PERTOL idea, index-local form)
  if (gi > 0) {
    const double prev = w[gi - 1];
    const double pertol = 10.0 * fabs(EPS64 * lam);
    if (lam - prev < pertol) lam = prev + pertol;
  }
#define AT(arr, i) arr[(i) * B + t]
  if (n == 1) {
    AT(wx, 0) = 1.0;
    inv_norm[t] = 1.0;
    return;
  }
  // ---- dlagtf: LU of T - lam I with partial pivoting. a=diag, b=super, c=sub(multipliers), d2=2nd super
  double ak = d[0] - lam;
  double bk = e[0];
  double scale1 = fabs(ak) + fabs(bk);
  double tol = 0.0;  // dlagts tolerance: eps * max |a|,|b|,|d2| of the factors
  for (int64_t k = 0; k < n - 1; ++k) {
    double ak1 = d[k + 1] - lam;
    const double ck = e[k];
    const double bk1 = (k < n - 2) ? e[k + 1] : 0.0;
    double scale2 = fabs(ck) + fabs(ak1);
    if (k < n - 2) scale2 += fabs(bk1);
    const double piv1 = (ak == 0.0) ? 0.0 : fabs(ak) / scale1;
    double a_out, b_out, c_out, d_out = 0.0, b_next = bk1;
    unsigned char in_k;
    if (ck == 0.0) {
      in_k = 0; a_out = ak; b_out = bk; c_out = ck;
    } else {
      const double piv2 = fabs(ck) / scale2;
      if (piv2 <= piv1) {
        in_k = 0;
        c_out = ck / ak;
        ak1 -= c_out * bk;
        a_out = ak; b_out = bk;
      } else {
        in_k = 1;
        const double mult = ak / ck;
        a_out = ck;
        const double temp = ak1;
        ak1 = bk - mult * temp;
        if (k < n - 2) { d_out = bk1; b_next = -mult * d_out; }
        b_out = temp;
        c_out = mult;
      }
    }
    scale1 = scale2;
    AT(wa, k) = a_out; AT(wb, k) = b_out; AT(wc, k) = c_out; AT(win, k) = in_k;
    if (k < n - 2) AT(wd, k) = d_out;
    tol = fmax(tol, fmax(fabs(a_out), fmax(fabs(b_out), fabs(d_out))));
    ak = ak1;
    bk = b_next;
  }
  AT(wa, n - 1) = ak;
  const double a_last = ak;
  tol = fmax(tol, fabs(a_last)) * EPS64;
  if (tol == 0.0) tol = EPS64;
  const double dtpcrt = sqrt(0.1 / (double)n);
  const double bignum = 1.0 / SFMIN64;
  int nrmchk = 0, its = 0;
  bool failed = false;
  double xmax = 1.0;  // the start vector is uniform(-1,1), generated on the fly in the first forward sweep
  double s2 = 1.0;
  while (true) {
    ++its;
    if (its > 8) { failed = true; break; }
    const bool first = (its == 1);
    const double scl = (double)n * onenrm * fmax(EPS64, fabs(a_last)) / xmax;
    // ---- forward: y = L^-1 P (scl * x)
    double yprev = (first ? hash_uniform((uint64_t)gi, 0) : AT(wx, 0)) * scl;
    int64_t k = 1;
    for (; k + STEIN_PF <= n; k += STEIN_PF) {
      double xk[STEIN_PF], ck[STEIN_PF];
      unsigned char ik[STEIN_PF];
#pragma unroll
      for (int u = 0; u < STEIN_PF; ++u) {
        xk[u] = first ? hash_uniform((uint64_t)gi, (uint64_t)(k + u)) : AT(wx, k + u);
        ck[u] = AT(wc, k + u - 1);
        ik[u] = AT(win, k + u - 1);
      }
#pragma unroll
      for (int u = 0; u < STEIN_PF; ++u) {
        const double yk = xk[u] * scl;
        double ynew;
        if (ik[u] == 0) {
          ynew = yk - ck[u] * yprev;
          AT(wx, k + u - 1) = yprev;
        } else {
          AT(wx, k + u - 1) = yk;
          ynew = yprev - ck[u] * yk;
        }
        yprev = ynew;
      }
    }
    for (; k < n; ++k) {
      const double yk = (first ? hash_uniform((uint64_t)gi, (uint64_t)k) : AT(wx, k)) * scl;
      const double ck = AT(wc, k - 1);
      double ynew;
      if (AT(win, k - 1) == 0) {
        ynew = yk - ck * yprev;
        AT(wx, k - 1) = yprev;
      } else {
        AT(wx, k - 1) = yk;
        ynew = yprev - ck * yk;
      }
      yprev = ynew;
    }
    AT(wx, n - 1) = yprev;
    // ---- backward with pivot perturbation (dlagts job = -1); tracks max |y| and sum y^2
    double y1 = 0.0, y2 = 0.0, nrm = 0.0;
    s2 = 0.0;
    auto solve_one = [&](int64_t kk, double xv, double av, double bv, double dv) {
      double temp = xv - bv * y1 - dv * y2;
      double akk = av;
      double pert = copysign(tol, akk);
      while (true) {
        const double absak = fabs(akk);
        if (absak < 1.0) {
          if (absak < SFMIN64) {
            if (absak == 0.0 || fabs(temp) * SFMIN64 > absak) { akk += pert; pert *= 2.0; continue; }
            temp *= bignum; akk *= bignum;
          } else if (fabs(temp) > absak * bignum) { akk += pert; pert *= 2.0; continue; }
        }
        break;
      }
      const double yk = temp / akk;
      AT(wx, kk) = yk;
      nrm = fmax(nrm, fabs(yk));
      s2 += yk * yk;
      y2 = y1; y1 = yk;
    };
    int64_t kb = n - 1;
    // the top two rows have no b / d2 entries: handle them singly, then blocks of four
    for (; kb >= n - 2 && kb >= 0; --kb)
      solve_one(kb, AT(wx, kb), AT(wa, kb), (kb <= n - 2) ? AT(wb, kb) : 0.0, 0.0);
    for (; kb >= STEIN_PF - 1; kb -= STEIN_PF) {
      double xv[STEIN_PF], av[STEIN_PF], bv[STEIN_PF], dv[STEIN_PF];
#pragma unroll
      for (int u = 0; u < STEIN_PF; ++u) {
        xv[u] = AT(wx, kb - u); av[u] = AT(wa, kb - u); bv[u] = AT(wb, kb - u); dv[u] = AT(wd, kb - u);
      }
#pragma unroll
      for (int u = 0; u < STEIN_PF; ++u) solve_one(kb - u, xv[u], av[u], bv[u], dv[u]);
    }
    for (; kb >= 0; --kb) solve_one(kb, AT(wx, kb), AT(wa, kb), AT(wb, kb), AT(wd, kb));
    xmax = nrm;
    if (nrm < dtpcrt) continue;
    ++nrmchk;
    if (nrmchk < good_its) continue;
    break;
  }
  inv_norm[t] = 1.0 / sqrt(s2);
  if (failed) atomicAdd(fail_count, 1);
#undef AT
}

// wx [n][B] fp64 (unnormalised) * inv_norm[t] -> Zt rows (t0 + t) fp32: Zt[(t0+t)*ldz + i]
__global__ __launch_bounds__(256) void tri_transpose_out(const double* __restrict__ wx, const double* __restrict__ inv_norm,
                                                         int64_t n, int64_t count, int64_t B, float* __restrict__ Zt,
                                                         int64_t ldz, int64_t t0) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t tb = (int64_t)blockIdx.x * 32, ib = (int64_t)blockIdx.y * 32;
  const double sc = (tb + tx < count) ? inv_norm[tb + tx] : 0.0;
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = ib + r, t = tb + tx;
    tile[r][tx] = (i < n && t < count) ? (float)(wx[i * B + t] * sc) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t t = tb + r, i = ib + tx;
    if (t < count && i < n) Zt[(t0 + t) * ldz + i] = tile[tx][r];
  }
}

__global__ void tri_fill_nan(double* __restrict__ w, int64_t lo, int64_t hi) {
  const int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < hi) w[i] = __longlong_as_double(0x7ff8000000000000LL);
}

// n_low < 0: all eigenvalues. n_low >= 0: only the n_low smallest and the largest one (the sparsity search consumes the lower
// half of the spectrum, scLENS.jl:742, and the largest value for the positivity floor); the others are left as NaN.
int stebz_f64(Ctx* ctx, const double* d_dev, const double* e_dev, int64_t n, double* w_dev, int64_t n_low, int64_t k_top) {
  if (k_top < 0) k_top = n - 1;
  StageTimer tm(ctx, "stebz");
  SCL_WS(ctx, e2, double, "tri.e2", n + 1);
  SCL_WS(ctx, info, double, "tri.info", 8);
  hipLaunchKernelGGL(tri_bounds, dim3(1), dim3(1024), 0, ctx->stream, d_dev, e_dev, n, e2, info);
  const bool ratio_form = ctx->opt.bisect_div != 0;  // 1: the ratio form (round 2), for comparison
  const double* dd = d_dev;
  const double* ee = e2;
  if (!ratio_form) {  // division-free Sturm counts on a copy scaled to norm < 1
    SCL_WS(ctx, ds, double, "tri.ds", n + 1);
    SCL_WS(ctx, e2s, double, "tri.e2s", n + 1);
    hipLaunchKernelGGL(tri_scale, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, ctx->stream, d_dev, e2, n, info, ds, e2s);
    dd = ds;
    ee = e2s;
  }
  auto kern = ratio_form ? tri_bisect : tri_bisect_prod;
  if (n_low < 0 || n_low >= k_top) {
    hipLaunchKernelGGL(kern, dim3((unsigned)((n + 7) / 8)), dim3(64), 0, ctx->stream, dd, ee, n, info, w_dev, (int64_t)0, n, (int64_t)-1);
  } else {
    hipLaunchKernelGGL(tri_fill_nan, dim3((unsigned)((n - n_low + 255) / 256)), dim3(256), 0, ctx->stream, w_dev, n_low, n);
    hipLaunchKernelGGL(kern, dim3((unsigned)((n_low + 7) / 8 + 1)), dim3(64), 0, ctx->stream, dd, ee, n, info, w_dev, (int64_t)0, n_low, k_top);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

namespace {
struct SteinShared {
  void* p = nullptr;
  size_t bytes = 0;
  hipEvent_t last = nullptr;  // recorded after the last launch that used the block
  bool used = false;
};
std::mutex g_stein_mu;
std::map<int, SteinShared> g_stein;  // by device, under g_stein_mu
}  // namespace

int stein_f64(Ctx* ctx, const double* d_dev, const double* e_dev, int64_t n, const double* w_dev, int64_t lo,
              int64_t hi, float* Zt, int64_t ldz) {
  const int64_t m = hi - lo;
  if (m <= 0) return SCLENS_OK;
  StageTimer tm(ctx, "stein");
  double* info = static_cast<double*>(ctx->workspace("tri.info", 8 * sizeof(double)));  // filled by stebz_f64
  if (!info) return SCLENS_ERR_OOM;
  // batch so that the six [n][B] workspaces stay below ~24 GB (one thread per eigenvector and a sequential recurrence of
  // length n: the kernel is latency-bound, so the batch should cover all requested vectors -- 15 008 vectors of order
  // 30 016 in four batches of 76 waves took 280 ms, in one batch of 235 waves 70 ms)
  int64_t B = (int64_t)(24.0e9 / (41.0 * (double)n));
  B = B / 64 * 64;
  if (B < 64) B = 64;
  if (B > round_up(m, 64)) B = round_up(m, 64);
  // The six [n][B] workspaces: per context (default), or ONE set per device used in turn by every context of the process (stein_shared =
  // 1). With two streams the two contexts of a call each hold 24 GB of them at 30 000 genes -- 48 of the 172 GB at the peak of the search
  // -- for a 42 ms stage. Turn-taking is stream-ordered: a user waits on the event the previous user recorded after its last launch and
  // records its own; the host lock covers the enqueue only. Measured at 100 000 x 30 000, two streams (profiles/r06_stein_shared.md):
  // peak 172.0 -> 153.8 GB, call 41.07 -> 41.43 s (the two streams' inverse iterations no longer overlap): off by default, for hosts
  // that share the device.
  const size_t nb8 = (sizeof(double) * (size_t)n * (size_t)B + 255) & ~(size_t)255;
  const size_t nb1 = ((size_t)n * (size_t)B + 255) & ~(size_t)255;
  double *wa, *wb, *wc, *wd, *wx;
  unsigned char* win;
  std::unique_lock<std::mutex> turn(g_stein_mu, std::defer_lock);
  SteinShared* shared = nullptr;
  if (ctx->opt.stein_shared) {
    turn.lock();
    shared = &g_stein[ctx->device];
    if (!shared->last) SCL_HIP(ctx, hipEventCreateWithFlags(&shared->last, hipEventDisableTiming));
    const size_t need = 5 * nb8 + nb1;
    if (shared->bytes < need) {
      if (shared->p) {
        if (shared->used) SCL_HIP(ctx, hipEventSynchronize(shared->last));
        pool_free(shared->p, ctx->stream);
        shared->p = nullptr;
        shared->bytes = 0;
      }
      if (pool_malloc(&shared->p, need) != hipSuccess) return ctx->fail(SCLENS_ERR_OOM, "stein: shared workspaces");
      shared->bytes = need;
    }
    if (shared->used) SCL_HIP(ctx, hipStreamWaitEvent(ctx->stream, shared->last, 0));
    char* q = static_cast<char*>(shared->p);
    wa = reinterpret_cast<double*>(q);
    wb = reinterpret_cast<double*>(q + nb8);
    wc = reinterpret_cast<double*>(q + 2 * nb8);
    wd = reinterpret_cast<double*>(q + 3 * nb8);
    wx = reinterpret_cast<double*>(q + 4 * nb8);
    win = reinterpret_cast<unsigned char*>(q + 5 * nb8);
  } else {
    wa = static_cast<double*>(ctx->workspace("stein.a", nb8));
    wb = static_cast<double*>(ctx->workspace("stein.b", nb8));
    wc = static_cast<double*>(ctx->workspace("stein.c", nb8));
    wd = static_cast<double*>(ctx->workspace("stein.d", nb8));
    wx = static_cast<double*>(ctx->workspace("stein.x", nb8));
    win = static_cast<unsigned char*>(ctx->workspace("stein.in", nb1));
    if (!wa || !wb || !wc || !wd || !wx || !win) return SCLENS_ERR_OOM;
  }
  SCL_WS(ctx, failc, int, "stein.fail", 4);
  SCL_WS(ctx, invn, double, "stein.invn", B);
  SCL_HIP(ctx, hipMemsetAsync(failc, 0, sizeof(int), ctx->stream));
  const int pf = (int)ctx->opt.stein_pf;  // steps of loads in flight: 4 (round 2), 16 (default), 32
  // Iterations with sufficient growth (dstein's criterion |x|max >= sqrt(0.1 / n)) before a vector is accepted. dstein runs 1 + EXTRA
  // = 3; here 2: the vectors leave this solver as fp32 and go through fp32 back-transformations, and the fp64 residual after the
  // first such iteration is already at working precision (scripts/stein_its.py: residual and orthogonality of the final vectors
  // are the same to four digits for 1, 2 and 3 at n = 1 000 .. 8 192, planted near-degenerate pairs included). -20 % of the stage.
  const int good_its = (int)std::max<int64_t>(1, std::min<int64_t>(5, ctx->opt.stein_its));
  for (int64_t t0 = 0; t0 < m; t0 += B) {
    const int64_t cnt = (m - t0 < B) ? m - t0 : B;
    if (pf == 32)
      hipLaunchKernelGGL(tri_stein<32>, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, ctx->stream, d_dev, e_dev, n, w_dev, lo + t0, cnt, B,
                         info, wa, wb, wc, wd, win, wx, invn, failc, good_its);
    else if (pf == 4)
      hipLaunchKernelGGL(tri_stein<4>, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, ctx->stream, d_dev, e_dev, n, w_dev, lo + t0, cnt, B,
                         info, wa, wb, wc, wd, win, wx, invn, failc, good_its);
    else
      hipLaunchKernelGGL(tri_stein<16>, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, ctx->stream, d_dev, e_dev, n, w_dev, lo + t0, cnt, B,
                         info, wa, wb, wc, wd, win, wx, invn, failc, good_its);
    hipLaunchKernelGGL(tri_transpose_out, dim3((unsigned)((cnt + 31) / 32), (unsigned)((n + 31) / 32)), dim3(256),
                       0, ctx->stream, wx, invn, n, cnt, B, Zt, ldz, t0);
  }
  SCL_HIP(ctx, hipGetLastError());
  if (shared) {
    SCL_HIP(ctx, hipEventRecord(shared->last, ctx->stream));
    shared->used = true;
  }
  return SCLENS_OK;
}

// the device's shared inverse-iteration workspaces back to the pool (release_scratch: eigensolver / all / everything)
void stein_shared_release(int device, hipStream_t stream) {  // device < 0: every device
  std::lock_guard<std::mutex> lk(g_stein_mu);
  for (auto& kv : g_stein) {
    if ((device >= 0 && kv.first != device) || !kv.second.p) continue;
    if (kv.second.used) (void)hipEventSynchronize(kv.second.last);  // the last user's kernels have left the block
    pool_free(kv.second.p, stream);
    kv.second.p = nullptr;
    kv.second.bytes = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// Block-reflector T factors for all panels at once (forward, columnwise; LAPACK dlarft recurrence):
//   T[c][c] = tau_c,  T[0:c, c] = -tau_c * T[0:c, 0:c] * G[0:c, c],  G[cc][c] = V[:,cc]^T V[:,c] = Gst[p+c][cc]
// Trep[pnl] is [NB][S*NB] with Trep[r][s*NB + k] = T[r][k] (S copies, see the split-K use in ormtr_f32).
__global__ __launch_bounds__(NB) void build_T(const float* __restrict__ Gst, const float* __restrict__ tau,
                                              int64_t nref, float* __restrict__ Trep, int S) {
  __shared__ float T[NB][NB + 1];
  __shared__ float g[NB];
  const int64_t p = (int64_t)blockIdx.x * NB;
  const int k = (int)((nref - p < NB) ? nref - p : NB);
  const int tid = threadIdx.x;
  for (int c = 0; c < NB; ++c) T[tid][c] = 0.f;
  __syncthreads();
  for (int c = 0; c < k; ++c) {
    const float tc = tau[p + c];
    if (tid < c) g[tid] = Gst[(p + c) * NB + tid];
    __syncthreads();
    if (tid < c) {
      float s = 0.f;
      for (int q = tid; q < c; ++q) s += T[tid][q] * g[q];  // T upper triangular: T[tid][q] = 0 for q < tid
      T[tid][c] = -tc * s;
    }
    if (tid == c) T[c][c] = tc;
    __syncthreads();
  }
  float* out = Trep + (int64_t)blockIdx.x * NB * ((int64_t)S * NB);
  for (int r = 0; r < NB; ++r)
    for (int s = 0; s < S; ++s) out[(int64_t)r * S * NB + s * NB + tid] = T[r][tid];
}

// Clean reflector panel: Vp[c][i] = v_{p+c}[i] (0 for i <= p+c, 1 at i = p+c+1), rows c >= k are zero.
__global__ __launch_bounds__(256) void build_Vp(const float* __restrict__ A, int64_t n, int64_t lda, int64_t p,
                                                int k, int64_t k_al, float* __restrict__ Vp, int64_t ldvp) {
  const int c = blockIdx.y;
  const int64_t i = k_al + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  if (c < k) {
    const int64_t j = p + c;
    if (i == j + 1) v = 1.f;
    else if (i > j + 1) v = A[j * lda + i];
  }
  Vp[(int64_t)c * ldvp + i] = v;
}

int ormtr_f32(Ctx* ctx, const float* A, int64_t n, int64_t lda, const float* tau_dev, float* Zt, int64_t m,
              int64_t ldz) {
  if (m <= 0 || n <= 1) return SCLENS_OK;
  if (ldz % 4 != 0 || (reinterpret_cast<uintptr_t>(Zt) & 15u))
    return ctx->fail(SCLENS_ERR_ARG, "ormtr_f32: Zt must be 16-byte aligned with ldz a multiple of 4");
  StageTimer tm(ctx, "ormtr");
  const int64_t nref = n - 1;  // reflectors j = 0 .. n-2 (the last ones have tau = 0)
  const int64_t npanel = (nref + NB - 1) / NB;
  float* Gst = static_cast<float*>(ctx->workspace("trd.Gst", sizeof(float) * n * NB));
  if (!Gst) return SCLENS_ERR_OOM;
  // split-K factor of the skinny first GEMM (m x NB output): fill the chip
  const int64_t tiles_m = (m + 127) / 128;
  int S = (int)((512 + tiles_m - 1) / tiles_m);
  if (S < 1) S = 1;
  if (S > 16) S = 16;
  const int64_t ldvp = round_up(n, 4);
  SCL_WS(ctx, Trep, float, "orm.Trep", npanel * NB * (int64_t)S * NB);
  SCL_WS(ctx, Vp, float, "orm.Vp", NB * ldvp);
  SCL_WS(ctx, W1, float, "orm.W1", m * (int64_t)S * NB);
  SCL_WS(ctx, W2, float, "orm.W2", m * NB);
  hipLaunchKernelGGL(build_T, dim3((unsigned)npanel), dim3(NB), 0, ctx->stream, Gst, tau_dev, nref, Trep, S);
  for (int64_t pi = npanel - 1; pi >= 0; --pi) {
    const int64_t p = pi * NB;
    const int k = (int)((nref - p < NB) ? nref - p : NB);
    const int64_t k_al = (p + 1) & ~(int64_t)15;
    const int64_t Kr = n - k_al;
    hipLaunchKernelGGL(build_Vp, dim3((unsigned)((Kr + 255) / 256), NB), dim3(256), 0, ctx->stream, A, n, lda, p, k,
                       k_al, Vp, ldvp);
    // W1[m][S][NB] : split-K partials of Zt[:, k_al:] * Vp[:, k_al:]^T, one launch (grid.y = slice)
    {
      GemmArgs g{};
      g.P = Zt + k_al; g.Q = Vp + k_al; g.C = W1;
      g.M = m; g.N = NB; g.K = Kr;
      g.ldp = ldz; g.ldq = ldvp; g.ldc = (int64_t)S * NB;
      g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = nullptr;
      g.splits = S; g.k_chunk = round_up((Kr + S - 1) / S, 16); g.c_split_off = NB;
      SCL_TRY(gemm_f32(ctx, g));
    }
    {  // W2 = (sum_s W1_s) * T^T   (NT with the S-fold replicated T)
      GemmArgs g{};
      g.P = W1; g.Q = Trep + pi * NB * ((int64_t)S * NB); g.C = W2;
      g.M = m; g.N = NB; g.K = (int64_t)S * NB;
      g.ldp = (int64_t)S * NB; g.ldq = (int64_t)S * NB; g.ldc = NB;
      g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = nullptr;
      SCL_TRY(gemm_f32(ctx, g));
    }
    {  // Zt[:, k_al:] -= W2 * Vp[:, k_al:]
      GemmArgs g{};
      g.P = W2; g.Q = Vp + k_al; g.C = Zt + k_al;
      g.M = m; g.N = Kr; g.K = NB;
      g.ldp = NB; g.ldq = ldvp; g.ldc = ldz;
      g.alpha = -1.f; g.beta = 1.f; g.q_kcontig = 0; g.lower = 0; g.colabsmax = nullptr;
      SCL_TRY(gemm_f32(ctx, g));
    }
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

// Modified Gram-Schmidt over `cnt` consecutive rows of Zt (one block). Inverse iteration runs every eigenvector
// independently; for eigenvalues that coincide to fp32 resolution it returns independent but not orthogonal vectors of the
// shared eigenspace, so those (rare, small) clusters are orthonormalised afterwards, like LAPACK's dstein does inside.
__global__ __launch_bounds__(256) void k_mgs_rows(float* __restrict__ Zt, int64_t ldz, int64_t n, int64_t r0, int cnt) {
  __shared__ double sw[4];
  __shared__ double bc;
  for (int i = 0; i < cnt; ++i) {
    float* ri = Zt + (r0 + i) * ldz;
    for (int j = 0; j <= i; ++j) {  // j < i: project out row j; j == i: normalise
      const float* rj = Zt + (r0 + j) * ldz;
      double s = 0.0;
      for (int64_t c = threadIdx.x; c < n; c += 256) s += (double)ri[c] * (double)rj[c];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) bc = (sw[0] + sw[1]) + (sw[2] + sw[3]);
      __syncthreads();
      const double dot = bc;
      if (j < i) {
        for (int64_t c = threadIdx.x; c < n; c += 256) ri[c] = (float)((double)ri[c] - dot * (double)rj[c]);
      } else {
        const double inv = 1.0 / sqrt(dot);
        for (int64_t c = threadIdx.x; c < n; c += 256) ri[c] = (float)((double)ri[c] * inv);
      }
      __syncthreads();
    }
  }
}

int eig_values(Ctx* ctx, float* A, int64_t n, int64_t lda, double* w64_dev, int64_t n_low) {
  if (n <= 0) return SCLENS_OK;
  ctx->last_two_stage = false;
  // two_stage: 1 = always (orders >= 128), 0 = never, -1 (default) = from the order: the one-stage reduction streams the
  // trailing matrix once per column (2/3 n^3 bytes). A single decomposition of order 10^4 takes the same time either way
  // (0.42 s vs 0.41 s with n/2 vectors; 5.3 s vs 2.0 s at 3 * 10^4), but three concurrent ones (the host's default below
  // n = 16 000) overlap better in the two-stage form: a whole sclens() call at 10 000 x 20 000 takes 4.76 s instead of 6.09 s
  // (scripts/sweep_cfg2_paths.sh), so the two-stage solver is the default from n = 8 192
  const int ts = ctx->opt.eff_two_stage();
  if (ts == 1 || (ts < 0 && n >= ctx->opt.two_stage_min_n)) {
    int used = 0;
    SCL_TRY(eig_values_two_stage(ctx, A, n, lda, w64_dev, &used, n_low));
    if (used) {
      ctx->last_two_stage = true;
      return SCLENS_OK;
    }
  }
  SCL_WS(ctx, d, double, "eig.d", n);
  SCL_WS(ctx, e, double, "eig.e", n);
  SCL_WS(ctx, tau, float, "eig.tau", n);
  SCL_TRY(sytrd_f32(ctx, A, n, lda, d, e, tau));
  SCL_TRY(stebz_f64(ctx, d, e, n, w64_dev, n_low));
  return SCLENS_OK;
}

// every eigenvalue from the tridiagonal matrix the last eig_values call of this context left behind (after a partial call)
int eig_values_redo_all(Ctx* ctx, int64_t n, double* w64_dev) {
  if (ctx->last_two_stage) return eig_values_two_stage_redo(ctx, n, w64_dev);
  SCL_WS(ctx, d, double, "eig.d", n);
  SCL_WS(ctx, e, double, "eig.e", n);
  return stebz_f64(ctx, d, e, n, w64_dev);
}

int eig_vectors(Ctx* ctx, const float* A, int64_t n, int64_t lda, const double* w64_dev, int64_t vec_lo,
                int64_t vec_hi, float* Zt, int64_t ldz) {
  if (vec_hi <= vec_lo) return SCLENS_OK;
  if (vec_lo < 0 || vec_hi > n || !Zt) return ctx->fail(SCLENS_ERR_ARG, "eig_vectors: bad eigenvector range");
  if (ctx->last_two_stage) {
    SCL_TRY(eig_vectors_two_stage(ctx, n, vec_lo, vec_hi, Zt, ldz));
  } else {
    SCL_WS(ctx, d, double, "eig.d", n);
    SCL_WS(ctx, e, double, "eig.e", n);
    SCL_WS(ctx, tau, float, "eig.tau", n);
    SCL_TRY(stein_f64(ctx, d, e, n, w64_dev, vec_lo, vec_hi, Zt, ldz));
    SCL_TRY(ormtr_f32(ctx, A, n, lda, tau, Zt, vec_hi - vec_lo, ldz));
  }
  // clusters of eigenvalues that coincide to fp32 resolution: orthonormalise their vectors
  const int64_t m = vec_hi - vec_lo;
  if (m > 1) {
    std::vector<double> w(m);
    SCL_HIP(ctx, hipMemcpyAsync(w.data(), w64_dev + vec_lo, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double wmax = 0.0;
    for (double v : w) wmax = std::max(wmax, std::fabs(v));
    const double tol = 2.4e-7 * wmax;
    int64_t i = 0;
    while (i + 1 < m) {
      int64_t jn = i;
      while (jn + 1 < m && w[jn + 1] - w[jn] <= tol) ++jn;
      if (jn > i) {
        // one workgroup, O(cnt^2 n): meant for the handful of numerically coincident pairs of this path. A cluster of more
        // than 128 (a rank-deficient input, e.g. thousands of exact null eigenvalues) is left as inverse iteration returned
        // it -- unit vectors spanning the eigenspace, not orthogonalised -- and counted.
        if (jn - i + 1 <= 128) {
          hipLaunchKernelGGL(k_mgs_rows, dim3(1), dim3(256), 0, ctx->stream, Zt, ldz, n, i, (int)(jn - i + 1));
          ctx->t_calls["degenerate_clusters"] += 1;
        } else {
          ctx->t_calls["degenerate_clusters_skipped"] += 1;
        }
      }
      i = jn + 1;
    }
    SCL_HIP(ctx, hipGetLastError());
  }
  return SCLENS_OK;
}

int eigh_f32(Ctx* ctx, float* A, int64_t n, int64_t lda, double* w64_dev, int64_t vec_lo, int64_t vec_hi,
             float* Zt, int64_t ldz) {
  SCL_TRY(eig_values(ctx, A, n, lda, w64_dev));
  SCL_TRY(eig_vectors(ctx, A, n, lda, w64_dev, vec_lo, vec_hi, Zt, ldz));
  return SCLENS_OK;
}

}  // namespace scl
