// Host-side RMT statistics of the sclens() path (O(n), never on the device in the reference either):
// _mp_parameters scLENS.jl:390-408, _marchenko_pastur :411-418, _mp_calculation :424-459, _tw :461-467,
// mp_check :469-487, and the Tukey-fence / median scoring of :797-806. Plain C++ (no HIP).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/sclens_hip.h"

namespace {

struct MPP { double m1, m2, gamma, b_plus, b_minus; };

MPP mp_parameters(const double* L, int64_t n, const uint8_t* mask) {  // scLENS.jl:390-408
  double s1 = 0, s2 = 0;
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i)
    if (!mask || mask[i]) { s1 += L[i]; s2 += L[i] * L[i]; ++c; }
  const double nan = std::numeric_limits<double>::quiet_NaN();
  MPP r;
  r.m1 = c ? s1 / (double)c : nan;  // mean of an empty vector is NaN (Appendix A12)
  r.m2 = c ? s2 / (double)c : nan;
  r.gamma = r.m2 / (r.m1 * r.m1) - 1.0;
  const double sg = std::sqrt(r.gamma);
  r.b_plus = r.m1 * (1 + sg) * (1 + sg);
  r.b_minus = r.m1 * (1 - sg) * (1 - sg);
  return r;
}

double mp_pdf(double x, const MPP& y) {  // scLENS.jl:411-418
  if (y.b_minus < x && x < y.b_plus)
    return std::sqrt((y.b_plus - x) * (x - y.b_minus)) / (2 * y.m1 * M_PI * y.gamma * x);
  return 0.0;
}

double quantile7(std::vector<double>& v, double q) {  // Julia quantile default (type 7); v sorted
  const size_t n = v.size();
  const double h = (double)(n - 1) * q;
  const size_t lo = (size_t)std::floor(h);
  const size_t hi = std::min(lo + 1, n - 1);
  return v[lo] + (h - (double)lo) * (v[hi] - v[lo]);
}

}  // namespace

extern "C" {

int sclens_mp_calculation(const double* L, int64_t n, const double* Lr, int64_t nr, double* b_plus_out,
                          double* b_minus_out, uint8_t* L_mp_mask) {
  if (!L || !Lr || n <= 0 || nr <= 0 || !L_mp_mask) return SCLENS_ERR_ARG;
  const double eta = 1.0, eps = 1e-6;
  const int max_iter = 10000;
  MPP r = mp_parameters(Lr, nr, nullptr);
  double b_plus = r.b_plus, b_minus = r.b_minus;
  std::vector<uint8_t> mask(n);
  auto select = [&](double lo, double hi) {
    for (int64_t i = 0; i < n; ++i) mask[i] = (lo < L[i] && L[i] < hi) ? 1 : 0;  // strict (Appendix A11)
  };
  select(b_minus, b_plus);
  MPP nw = mp_parameters(L, n, mask.data());
  double new_b_plus = nw.b_plus, new_b_minus = nw.b_minus;
  int iter = 0;
  while (true) {
    const double loss = (1 - new_b_plus / b_plus) * (1 - new_b_plus / b_plus);
    ++iter;
    if (loss <= eps) break;
    if (iter == max_iter) break;
    const double gradient = new_b_plus - b_plus;
    new_b_plus = b_plus + eta * gradient;
    select(new_b_minus, new_b_plus);
    b_plus = new_b_plus;
    b_minus = new_b_minus;
    MPP up = mp_parameters(L, n, mask.data());
    new_b_plus = up.b_plus;
    new_b_minus = up.b_minus;
  }
  select(new_b_minus, new_b_plus);
  std::copy(mask.begin(), mask.end(), L_mp_mask);
  if (b_plus_out) *b_plus_out = new_b_plus;
  if (b_minus_out) *b_minus_out = new_b_minus;
  return SCLENS_OK;
}

int sclens_tw(int64_t n_all, const double* L_mp, int64_t n_mp, double* lambda_c, double* gamma_out, double* p_out,
              double* sigma_out) {
  if (!L_mp || n_mp <= 0 || !lambda_c) return SCLENS_ERR_ARG;
  MPP r = mp_parameters(L_mp, n_mp, nullptr);
  const double gamma = r.gamma;
  const double p = (double)n_all / gamma;  // length of ALL eigenvalues (Appendix A13)
  const double sigma = 1.0 / std::pow(p, 2.0 / 3.0) * std::pow(gamma, 5.0 / 6.0) * std::pow(1 + std::sqrt(gamma), 4.0 / 3.0);
  *lambda_c = r.m1 * (1 + std::sqrt(gamma)) * (1 + std::sqrt(gamma)) + sigma;
  if (gamma_out) *gamma_out = gamma;
  if (p_out) *p_out = p;
  if (sigma_out) *sigma_out = sigma;
  return SCLENS_OK;
}

int sclens_mp_check(const double* L_mp, int64_t n_mp, double p_val, double* ks_static, int* pass) {
  if (!L_mp || n_mp <= 0) return SCLENS_ERR_ARG;
  double mn = L_mp[0], mx = L_mp[0];
  for (int64_t i = 1; i < n_mp; ++i) { mn = std::min(mn, L_mp[i]); mx = std::max(mx, L_mp[i]); }
  const double lo = mn - 1, hi = mx + 1, step = (hi - lo) / 99.0;  // LinRange(lo, hi, 100): 99 bins (Appendix A14)
  std::vector<double> count(99, 0.0);
  double total = 0;
  for (int64_t i = 0; i < n_mp; ++i) {
    const int64_t b = (int64_t)std::floor((L_mp[i] - lo) / step);
    if (b >= 0 && b < 99) { count[b] += 1; total += 1; }
  }
  MPP par = mp_parameters(L_mp, n_mp, nullptr);
  std::vector<double> c2(99);
  double run = 0, cmax = 0;
  for (int b = 0; b < 99; ++b) {
    const double e0 = lo + (hi - lo) * (double)b / 99.0, e1 = lo + (hi - lo) * (double)(b + 1) / 99.0;
    run += mp_pdf(0.5 * (e0 + e1), par);
    c2[b] = run;
    cmax = std::max(cmax, run);
  }
  double D = 0, cdf = 0;
  for (int b = 0; b < 99; ++b) {
    cdf += count[b] / total;
    D = std::max(D, std::fabs(cdf - c2[b] / cmax));
  }
  const double c_alpha = std::sqrt(-0.5 * std::log(p_val));
  const double m = 99, nn = 99;
  if (ks_static) *ks_static = D;
  if (pass) *pass = (D <= c_alpha * std::sqrt((m + nn) / m / nn)) ? 1 : 0;
  return SCLENS_OK;
}

int sclens_robust_scores(const double* b, int64_t k, int64_t npairs, double* m_score, double* sd_score) {
  if (!b || k <= 0 || npairs <= 0 || !m_score) return SCLENS_ERR_ARG;
  std::vector<double> row(npairs), f;
  for (int64_t s = 0; s < k; ++s) {
    std::copy(b + s * npairs, b + (s + 1) * npairs, row.begin());
    std::sort(row.begin(), row.end());
    const double q1 = quantile7(row, 0.25), q3 = quantile7(row, 0.75), iqr = q3 - q1;
    f.clear();
    for (double v : row)
      if (q1 - 1.5 * iqr <= v && v <= q3 + 1.5 * iqr) f.push_back(v);  // inclusive fence (Appendix A27)
    const size_t c = f.size();
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (c == 0) { m_score[s] = nan; if (sd_score) sd_score[s] = nan; continue; }
    m_score[s] = (c % 2) ? f[c / 2] : 0.5 * (f[c / 2 - 1] + f[c / 2]);  // f is sorted
    if (sd_score) {
      double mean = 0;
      for (double v : f) mean += v;
      mean /= (double)c;
      double ss = 0;
      for (double v : f) ss += (v - mean) * (v - mean);
      sd_score[s] = (c > 1) ? std::sqrt(ss / (double)(c - 1)) : nan;
    }
  }
  return SCLENS_OK;
}

double sclens_noise_baseline_exact(int64_t n) {
  // E max_{i<=n} |g_i| / sqrt(n), g ~ N(0,1):  E max = int_0^inf 1 - erf(x/sqrt2)^n dx  (Simpson)
  if (n <= 0) return 0.0;
  const int steps = 40000;
  const double hi = 14.0, h = hi / steps;
  auto f = [&](double x) { return 1.0 - std::pow(std::erf(x / std::sqrt(2.0)), (double)n); };
  double s = f(0) + f(hi);
  for (int i = 1; i < steps; ++i) s += f(i * h) * ((i & 1) ? 4.0 : 2.0);
  return s * h / 3.0 / std::sqrt((double)n);
}

}  // extern "C"
