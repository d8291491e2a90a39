// qpc_qp_prep.h -- host side of the PREPARED L0 problem (qpc_qp_setup / qpc_qp_solve, include/qpc.h): the optimizer's life cycle of
// the reference -- `MOI.copy_to` once with a fixed sparsity pattern, then coefficient updates and `MOI.optimize!` every tick
// (Parametron `solve!` -> `update!`, /root/reference/src/lowlevel/momentum.jl:58) -- split the way OSQP splits it (osqp_setup /
// osqp_update_* / osqp_solve): everything that depends on the PATTERN only is done once, here, on the host:
//   * the compressed columns and rows of A (orders of the caller's value array),
//   * the diagonal of P (the workgroup solver takes a diagonal P; a pattern with off-diagonal entries goes to the dense path),
//   * the structure of G = A H^-1 A': for every structurally nonzero G_rs (r >= s) the list of (a_rc, a_sc, c) it sums over, so that
//     the kernel builds S with one thread per entry, in a fixed order, without atomics.
// Plain C++ (no HIP): used by the runtime (which uploads `ints` once) and by the test emulation.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "qpc_qp_view.h"

namespace qpc {

struct QpPrepHost {
  int n = 0, m = 0;
  long long nnzP = 0, nnzA = 0;
  bool p_offdiag = false;
  bool wg_tables = false; // the term tables of the dense workgroup kernel are there
  int npairs = 0, ntrip = 0;
  std::vector<int> ints; // [pdiag | cptr | crow | cperm | rptr | rcol | ro | pptr | pr | ps | ta | tb | tc]
  std::vector<int> Prow, Pcol, Arow, Acol; // the caller's pattern (the dense path scatters with it)
  size_t o_pdiag = 0, o_cptr = 0, o_crow = 0, o_cperm = 0, o_rptr = 0, o_rcol = 0, o_ro = 0, o_pptr = 0, o_pr = 0, o_ps = 0, o_ta = 0, o_tb = 0,
         o_tc = 0;
  QpPrepView view(const int *base) const {
    QpPrepView V;
    V.n = n; V.m = m; V.nnzA = (int)nnzA; V.npairs = npairs;
    V.pdiag = base + o_pdiag; V.cptr = base + o_cptr; V.crow = base + o_crow; V.cperm = base + o_cperm; V.rptr = base + o_rptr;
    V.rcol = base + o_rcol; V.ro = base + o_ro; V.pptr = base + o_pptr; V.pr = base + o_pr; V.ps = base + o_ps; V.ta = base + o_ta;
    V.tb = base + o_tb; V.tc = base + o_tc;
    return V;
  }
};

// returns "" or what is wrong with the pattern
inline std::string qp_prep_build(QpPrepHost &H, int n, int m, long long nnzP, const int32_t *Prow, const int32_t *Pcol, long long nnzA,
                                 const int32_t *Arow, const int32_t *Acol) {
  if (n < 1 || m < 0 || nnzP < 0 || nnzA < 0) return "bad dimensions";
  if (nnzA > (long long)n * m || nnzP > (long long)n * n) return "more pattern entries than matrix entries";
  H.n = n; H.m = m; H.nnzP = nnzP; H.nnzA = nnzA;
  H.Prow.assign(Prow, Prow + nnzP); H.Pcol.assign(Pcol, Pcol + nnzP); H.Arow.assign(Arow, Arow + nnzA); H.Acol.assign(Acol, Acol + nnzA);
  std::vector<int> pdiag(n, -1);
  H.p_offdiag = false;
  {
    std::vector<long long> seen;
    seen.reserve(nnzP);
    for (long long k = 0; k < nnzP; k++) {
      const int r = Prow[k], c = Pcol[k];
      if (r < 0 || r >= n || c < 0 || c >= n) return "P pattern entry out of range";
      seen.push_back(r <= c ? (long long)r * n + c : (long long)c * n + r); // (i, j) and (j, i) are the same entry of a symmetric matrix
      if (r == c) pdiag[r] = (int)k; else H.p_offdiag = true;
    }
    std::sort(seen.begin(), seen.end());
    if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return "duplicate entry in the P pattern";
  }
  // columns of A: entries ordered by (column, row)
  std::vector<int> cperm(nnzA);
  for (long long k = 0; k < nnzA; k++) {
    if (Arow[k] < 0 || Arow[k] >= m || Acol[k] < 0 || Acol[k] >= n) return "A pattern entry out of range";
    cperm[k] = (int)k;
  }
  std::sort(cperm.begin(), cperm.end(), [&](int a, int b) { return Acol[a] != Acol[b] ? Acol[a] < Acol[b] : Arow[a] < Arow[b]; });
  for (long long o = 1; o < nnzA; o++)
    if (Acol[cperm[o]] == Acol[cperm[o - 1]] && Arow[cperm[o]] == Arow[cperm[o - 1]]) return "duplicate entry in the A pattern";
  std::vector<int> cptr(n + 1, 0), crow(nnzA), ccol(nnzA);
  for (long long o = 0; o < nnzA; o++) { crow[o] = Arow[cperm[o]]; ccol[o] = Acol[cperm[o]]; cptr[ccol[o] + 1]++; }
  for (int c = 0; c < n; c++) cptr[c + 1] += cptr[c];
  // rows of A: column-order indices ordered by (row, column)
  std::vector<int> ro(nnzA), rptr(m + 1, 0), rcol(nnzA);
  for (long long o = 0; o < nnzA; o++) ro[o] = (int)o;
  std::sort(ro.begin(), ro.end(), [&](int a, int b) { return crow[a] != crow[b] ? crow[a] < crow[b] : ccol[a] < ccol[b]; });
  for (long long k = 0; k < nnzA; k++) { rcol[k] = ccol[ro[k]]; rptr[crow[ro[k]] + 1]++; }
  for (int r = 0; r < m; r++) rptr[r + 1] += rptr[r];
  // structure of G = A H^-1 A' (lower triangle by row index): terms (r, s, c, o_r, o_s), sorted by (r, s, c) -- only when the dense
  // workgroup kernel (qpc_qp_wg.h: diagonal P, n, m <= 256) can run on this pattern: nobody else reads these tables, and for a
  // dense 300 x 400 pattern they take seconds and gigabytes
  struct Term { int r, s, c, a, b; };
  std::vector<Term> terms;
  H.wg_tables = !H.p_offdiag && n <= 256 && m <= 256;
  if (H.wg_tables) {
    long long nt = 0;
    for (int c = 0; c < n; c++) { const long long k = cptr[c + 1] - cptr[c]; nt += k * (k + 1) / 2; }
    if (nt > (1LL << 26)) return "pattern too dense for the workgroup solver's tables";
    terms.reserve((size_t)nt);
  }
  if (H.wg_tables)
  for (int c = 0; c < n; c++)
    for (int o1 = cptr[c]; o1 < cptr[c + 1]; o1++)
      for (int o2 = cptr[c]; o2 <= o1; o2++) terms.push_back(Term{crow[o1], crow[o2], c, o1, o2}); // (rows ascend inside a column: r >= s)
  std::sort(terms.begin(), terms.end(), [](const Term &x, const Term &y) {
    return x.r != y.r ? x.r < y.r : (x.s != y.s ? x.s < y.s : x.c < y.c);
  });
  std::vector<int> pptr, pr, ps, ta, tb, tc;
  for (size_t t = 0; t < terms.size(); t++) {
    if (t == 0 || terms[t].r != terms[t - 1].r || terms[t].s != terms[t - 1].s) { pptr.push_back((int)t); pr.push_back(terms[t].r); ps.push_back(terms[t].s); }
    ta.push_back(terms[t].a); tb.push_back(terms[t].b); tc.push_back(terms[t].c);
  }
  pptr.push_back((int)terms.size());
  H.npairs = (int)pr.size(); H.ntrip = (int)terms.size();
  H.ints.clear();
  auto put = [&](const std::vector<int> &v) { const size_t o = H.ints.size(); H.ints.insert(H.ints.end(), v.begin(), v.end()); if (v.empty()) H.ints.push_back(0); return o; };
  H.o_pdiag = put(pdiag); H.o_cptr = put(cptr); H.o_crow = put(crow); H.o_cperm = put(cperm); H.o_rptr = put(rptr); H.o_rcol = put(rcol);
  H.o_ro = put(ro); H.o_pptr = put(pptr); H.o_pr = put(pr); H.o_ps = put(ps); H.o_ta = put(ta); H.o_tb = put(tb); H.o_tc = put(tc);
  return "";
}

} // namespace qpc
