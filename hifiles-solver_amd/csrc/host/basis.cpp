#include "basis.hpp"

#include <cmath>

double eval_legendre(double r, int mode)
{
  if (mode == 0) return 1.0;
  if (mode == 1) return r;
  // three-term recurrence, evaluated top-down exactly like the recursive definition
  return ((2 * mode - 1) * r * eval_legendre(r, mode - 1) - (mode - 1) * eval_legendre(r, mode - 2)) / mode;
}

double eval_d_legendre(double r, int mode)
{
  if (mode == 0) return 0.0;
  if (r > -1.0 && r < 1.0) return (mode * ((r * eval_legendre(r, mode)) - eval_legendre(r, mode - 1))) / ((r * r) - 1.0);
  if (r == -1.0) return std::pow(-1.0, mode - 1.0) * 0.5 * mode * (mode + 1.0);
  return 0.5 * mode * (mode + 1.0);
}

void cubature_1d_nodes(int rule, int n, hf_array<double> &locs, hf_array<double> &weights)
{
  locs.setup(n);
  weights.setup(n);
  const double pi = 3.14159265358979323846;
  if (rule == 0)
  {
    for (int i = 0; i < n; i++)
    {
      // Newton on P_n from the Chebyshev guess; nodes ascending
      long double x = -std::cos(pi * (i + 0.75) / (n + 0.5));
      for (int it = 0; it < 100; it++)
      {
        long double p0 = 1.0L, p1 = x;
        for (int k = 2; k <= n; k++)
        {
          long double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
          p0 = p1;
          p1 = p2;
        }
        if (n == 1) { p0 = 1.0L; p1 = x; }
        long double dp = n * (x * p1 - p0) / (x * x - 1.0L);
        long double dx = p1 / dp;
        x -= dx;
        if (std::fabs((double)dx) < 1e-19) break;
      }
      long double p0 = 1.0L, p1 = x;
      for (int k = 2; k <= n; k++)
      {
        long double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
        p0 = p1;
        p1 = p2;
      }
      long double dp = n * (x * p1 - p0) / (x * x - 1.0L);
      locs(i) = (double)x;
      weights(i) = (double)(2.0L / ((1.0L - x * x) * dp * dp));
    }
    if (n % 2 == 1) locs(n / 2) = 0.0;
  }
  else
  {
    // Gauss-Lobatto: +-1 and the roots of P'_{n-1}
    const int N = n - 1;
    locs(0) = -1.0;
    locs(n - 1) = 1.0;
    for (int i = 1; i < n - 1; i++)
    {
      long double x = -std::cos(pi * i / N);
      for (int it = 0; it < 100; it++)
      {
        // f = (1-x^2) P'_N = N (P_{N-1} - x P_N); Newton with f' = -N(N+1) P_N
        long double p0 = 1.0L, p1 = x;
        for (int k = 2; k <= N; k++)
        {
          long double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
          p0 = p1;
          p1 = p2;
        }
        long double f = N * (p0 - x * p1);
        long double df = -(long double)N * (N + 1) * p1;
        long double dx = f / df;
        x -= dx;
        if (std::fabs((double)dx) < 1e-19) break;
      }
      locs(i) = (double)x;
    }
    if (n % 2 == 1) locs(n / 2) = 0.0;
    for (int i = 0; i < n; i++)
    {
      double pn = eval_legendre(locs(i), N);
      weights(i) = 2.0 / (N * (N + 1.0) * pn * pn);
    }
  }
}

double eval_lagrange(double r, int mode, const hf_array<double> &x)
{
  double v = 1.0;
  const int n = x.get_dim(0);
  for (int i = 0; i < n; i++)
    if (i != mode) v = v * ((r - x(i)) / (x(mode) - x(i)));
  return v;
}

double eval_d_lagrange(double r, int mode, const hf_array<double> &x)
{
  const int n = x.get_dim(0);
  double s = 0.0;
  for (int i = 0; i < n; i++)
  {
    if (i == mode) continue;
    double num = 1.0, den = 1.0;
    for (int j = 0; j < n; j++)
    {
      if (j != mode && j != i) num = num * (r - x(j));
      if (j != mode) den = den * (x(mode) - x(j));
    }
    s = s + (num / den);
  }
  return s;
}

double eval_d_vcjh_1d(double r, int mode, int order, double eta)
{
  // g'_L = 1/2 (-1)^p ( L'_p - (eta L'_{p-1} + L'_{p+1}) / (1+eta) ),
  // g'_R = 1/2        ( L'_p + (eta L'_{p-1} + L'_{p+1}) / (1+eta) )     
  const double lp = eval_d_legendre(r, order);
  const double lp1 = eval_d_legendre(r, order + 1);
  double mix;
  if (order == 0)
    mix = lp1 / (1.0 + eta);
  else
    mix = ((eta * eval_d_legendre(r, order - 1)) + lp1) / (1.0 + eta);
  if (mode == 0) return 0.5 * std::pow(-1.0, order) * (lp - mix);
  return 0.5 * (lp + mix);
}

static double factorial(int n)
{
  double f = 1.0;
  for (int i = 2; i <= n; i++) f *= i;
  return f;
}

bool compute_eta(int scheme, int order, double &eta)
{
  if (order == 0 && scheme != 1) return false;
  if (scheme == 1)
    eta = 0.0;
  else if (scheme == 2)
    eta = (1.0 * order) / (1.0 * (order + 1));
  else if (scheme == 3)
    eta = (1.0 * (order + 1)) / (1.0 * order);
  else if (scheme == 4)
  {
    double c_1d;
    if (order == 2) c_1d = 0.206;
    else if (order == 3) c_1d = 3.80e-3;
    else if (order == 4) c_1d = 4.67e-5;
    else if (order == 5) c_1d = 4.28e-7;
    else return false;
    double ap = 1. / std::pow(2.0, order) * factorial(2 * order) / (factorial(order) * factorial(order));
    eta = c_1d * (2 * order + 1) / 2 * (factorial(order) * ap) * (factorial(order) * ap);
  }
  else
    return false;
  return true;
}
