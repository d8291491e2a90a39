// qpc_qp_sym.h -- symbolic analysis of a PREPARED L0 problem, on the host, once per sparsity pattern and set of equality rows: what
// QDLDL's elimination tree buys the reference's OSQP (`Parametron.solve!` -> `MOI.optimize!`, /root/reference/src/lowlevel/momentum.jl:58;
// SURVEY.md 2.1 row `libosqp + QDLDL`), here for the range-space method of qpc_qp_wg.h: the equality rows of G = A H^-1 A' are
// eliminated in a fill-reducing order, as FRONTS (supernodes: a few adjacent columns with one row structure, stored as a small dense
// rectangle), level by level of the elimination tree -- the fronts of a level are independent of each other and are factored
// concurrently, one wavefront each; their updates of later fronts and of the Schur complement C on the inequality rows are 16 x 16
// tile jobs on the matrix core with a fixed summation order.  For the Atlas standing controller (143 x 178, 114 equality rows): 24 k
// multiply-adds in 3 levels instead of 924 k in 114 dependent columns.
//
// H = P + eps I may be BLOCK diagonal with blocks of at most QPC_SP_MAXBLK variables -- the e'We of a matrix-weighted task on its
// slack variables (`addtask!(controller, task, W::AbstractMatrix)`, momentum.jl:113-117; test/controller.jl:232-285).
//
// Output: the plan (qpc_qp_spview.h), one int array.  Plain C++ (no HIP): used by the runtime and by the test emulation.
#pragma once
#include <algorithm>
#include <bitset>
#include <cstdint>
#include <functional>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "qpc_qp_prep.h"
#include "qpc_qp_spview.h"

namespace qpc {

#define QPC_SP_DIM 256    // n, m <= this
#define QPC_SP_MAXBLK 8   // variables of a diagonal block of P
#define QPC_SP_WMAX 32    // columns of a front
#define QPC_SP_WSMALL 12  // fronts up to this width are merged freely (a level of the tree costs more than their padding)

struct QpSpPlanHost {
  bool valid = false;
  std::string why; // (when not valid)
  std::vector<uint8_t> eq;
  std::vector<int> ints;
  int n = 0, m = 0, ne = 0, mi = 0, nval = 0, nhv = 0, nfront = 0, nlevel = 0, njob = 0, lds_ints = 0, maxw = 0, maxr = 0;
  long long flops = 0; // multiply-adds of the elimination (diagnostic)
};

namespace sp_detail {
typedef std::bitset<QPC_SP_DIM + 64> Bits;
struct Front {
  int c0 = 0, w = 0, level = 0, parent = -1, base = 0, rowoff = 0;
  std::vector<int> rows; // positions: own columns first, then the rows below, ascending
  int r() const { return (int)rows.size(); }
  long long nnz() const { return (long long)w * r() - (long long)w * (w - 1) / 2; }
};
} // namespace sp_detail

// eq[r] != 0: row r is an equality row (l == u).  Returns out.valid.
inline bool qp_sp_plan_build(const QpPrepHost &H, const uint8_t *eq, QpSpPlanHost &out) {
  using namespace sp_detail;
  out = QpSpPlanHost();
  const int n = H.n, m = H.m;
  auto fail = [&](const char *w) { out.valid = false; out.why = w; return false; };
  if (n < 1 || m < 1 || n > QPC_SP_DIM || m > QPC_SP_DIM) return fail("dimensions outside the sparse workgroup solver");
  if (H.nnzA > 65535 || H.nnzP > 65535) return fail("too many pattern entries");
  const long long nnzA = H.nnzA, nnzP = H.nnzP;
  const int *cptr = H.ints.data() + H.o_cptr, *crow = H.ints.data() + H.o_crow, *cperm = H.ints.data() + H.o_cperm;
  const int *rptr = H.ints.data() + H.o_rptr, *rcol = H.ints.data() + H.o_rcol, *ro = H.ints.data() + H.o_ro;
  out.n = n; out.m = m;
  out.eq.assign(eq, eq + m);
  // ---- the diagonal blocks of P: connected components of its off-diagonal entries
  std::vector<int> comp(n);
  std::iota(comp.begin(), comp.end(), 0);
  std::function<int(int)> find = [&](int a) { while (comp[a] != a) { comp[a] = comp[comp[a]]; a = comp[a]; } return a; };
  for (long long k = 0; k < nnzP; k++) if (H.Prow[k] != H.Pcol[k]) { const int a = find(H.Prow[k]), b = find(H.Pcol[k]); if (a != b) comp[std::max(a, b)] = std::min(a, b); }
  std::vector<int> blk_of(n, -1), bptr(1, 0), bvar, bbase, lidx(n, 0);
  int nblk = 0, nhv = 0;
  for (int i = 0; i < n; i++) {
    if (find(i) != i) continue;
    const int b = nblk++;
    int bs = 0;
    for (int j = i; j < n; j++) if (find(j) == i) { blk_of[j] = b; lidx[j] = bs++; bvar.push_back(j); }
    if (bs > QPC_SP_MAXBLK) return fail("a diagonal block of P has more than QPC_SP_MAXBLK variables");
    bptr.push_back((int)bvar.size());
    bbase.push_back(nhv);
    nhv += bs * bs;
  }
  std::map<std::pair<int, int>, int> pent; // (row, col) of P, both orders -> index of the caller's value
  for (long long k = 0; k < nnzP; k++) { pent[{H.Prow[k], H.Pcol[k]}] = (int)k; pent[{H.Pcol[k], H.Prow[k]}] = (int)k; }
  std::vector<int> hptr(n + 1, 0), hcol, hidx, pidx;
  for (int i = 0; i < n; i++) {
    const int b = blk_of[i], bs = bptr[b + 1] - bptr[b];
    for (int lj = 0; lj < bs; lj++) {
      const int j = bvar[bptr[b] + lj];
      hcol.push_back(j);
      hidx.push_back(bbase[b] + lidx[i] * bs + lj);
      auto it = pent.find({i, j});
      pidx.push_back(it == pent.end() ? -1 : it->second);
    }
    hptr[i + 1] = (int)hcol.size();
  }
  // ---- which rows of G = A H^-1 A' are coupled: a_r and a_s touch the same block of H
  std::vector<Bits> cols(m), closure(m);
  for (int r = 0; r < m; r++)
    for (int k = rptr[r]; k < rptr[r + 1]; k++) {
      const int c = rcol[k], b = blk_of[c];
      cols[r].set(c);
      for (int t = bptr[b]; t < bptr[b + 1]; t++) closure[r].set(bvar[t]);
    }
  const int N1 = m + 1;
  std::vector<Bits> adj(N1); // node m: the right-hand-side row (coupled to every row)
  for (int r = 0; r < m; r++) {
    for (int s = 0; s < r; s++) if ((cols[r] & closure[s]).any()) { adj[r].set(s); adj[s].set(r); }
    adj[r].set(m); adj[m].set(r);
  }
  // ---- minimum-degree order of the equality rows (the other rows and the right-hand side stay to the end), the structure of
  //      every eliminated column, the elimination tree
  std::vector<int> erows;
  for (int r = 0; r < m; r++) if (eq[r]) erows.push_back(r);
  const int ne = (int)erows.size(), mi = m - ne;
  out.ne = ne; out.mi = mi;
  std::vector<char> alive(N1, 0), is_e(N1, 0);
  for (int r : erows) { alive[r] = 1; is_e[r] = 1; }
  std::vector<int> order, estep(N1, -1);
  std::vector<Bits> cstruct(N1);
  {
    std::vector<Bits> a = adj;
    for (int step = 0; step < ne; step++) {
      int best = -1;
      size_t bd = 0;
      for (int r : erows) if (alive[r]) { const size_t d = a[r].count(); if (best < 0 || d < bd) { best = r; bd = d; } }
      const Bits nb = a[best];
      cstruct[best] = nb;
      for (int v = 0; v < N1; v++) if (nb.test(v)) { a[v].reset(best); a[v] |= nb; a[v].reset(v); }
      alive[best] = 0;
      estep[best] = step;
      order.push_back(best);
    }
  }
  std::vector<int> parent(N1, -1); // by row: the first later-eliminated equality row of the column's structure
  std::vector<std::vector<int>> kids(N1);
  std::vector<int> roots;
  for (int r : order) {
    int p = -1;
    for (int v = 0; v < m; v++) if (cstruct[r].test(v) && is_e[v] && (p < 0 || estep[v] < estep[p])) p = v;
    parent[r] = p;
    if (p >= 0) kids[p].push_back(r); else roots.push_back(r);
  }
  // postorder (a subtree becomes a contiguous range of positions); the child with the largest subtree last: next to its parent
  std::vector<int> sub(N1, 1);
  for (int r : order) if (parent[r] >= 0) sub[parent[r]] += sub[r]; // (children are eliminated before their parents)
  std::vector<int> post;
  {
    std::vector<std::pair<int, size_t>> stack;
    auto order_kids = [&](std::vector<int> &k) { std::stable_sort(k.begin(), k.end(), [&](int a, int b) { return sub[a] != sub[b] ? sub[a] < sub[b] : a < b; }); };
    order_kids(roots);
    for (int r : order) order_kids(kids[r]);
    for (int root : roots) {
      stack.push_back({root, 0});
      while (!stack.empty()) {
        auto &top = stack.back();
        if (top.second < kids[top.first].size()) { const int c = kids[top.first][top.second++]; stack.push_back({c, 0}); }
        else { post.push_back(top.first); stack.pop_back(); }
      }
    }
  }
  std::vector<int> perm(N1), ipos(N1);
  for (int i = 0; i < ne; i++) perm[i] = post[i];
  { int p = ne; for (int r = 0; r < m; r++) if (!eq[r]) perm[p++] = r; }
  perm[m] = m;
  for (int p = 0; p < N1; p++) ipos[perm[p]] = p;
  // column structures by position
  std::vector<std::vector<int>> cs(ne);
  std::vector<int> ppar(ne, -1);
  for (int j = 0; j < ne; j++) {
    const int r = perm[j];
    for (int v = 0; v < N1; v++) if (cstruct[r].test(v)) cs[j].push_back(ipos[v]);
    std::sort(cs[j].begin(), cs[j].end());
    ppar[j] = parent[r] >= 0 ? ipos[parent[r]] : -1;
    out.flops += (long long)cs[j].size() * (cs[j].size() + 1) / 2;
  }
  // ---- fronts: every column by itself, then a child is merged into its parent when it sits right before it
  std::vector<Front> fr;
  for (int j = 0; j < ne; j++) {
    Front f;
    f.c0 = j; f.w = 1;
    f.rows.push_back(j);
    f.rows.insert(f.rows.end(), cs[j].begin(), cs[j].end());
    fr.push_back(f);
  }
  std::vector<int> nkids(ne, 0); // (of a column, in the tree of columns)
  for (int j = 0; j < ne; j++) if (ppar[j] >= 0) nkids[ppar[j]]++;
  for (bool changed = true; changed;) {
    changed = false;
    for (size_t i = 0; i + 1 < fr.size(); i++) {
      Front &s = fr[i], &p = fr[i + 1];
      const int last = s.c0 + s.w - 1;
      if (ppar[last] < p.c0 || ppar[last] >= p.c0 + p.w) continue; // not its parent
      const int w = s.w + p.w;
      if (w > QPC_SP_WMAX) continue;
      const long long r = (long long)s.w + p.r(), merged = (long long)w * r - (long long)w * (w - 1) / 2, sum = s.nnz() + p.nnz(), extra = merged - sum;
      bool ok;
      if (w <= QPC_SP_WSMALL) ok = merged <= 2 * sum + 64;
      else ok = extra == 0 || (nkids[ppar[last]] == 1 && ppar[last] == p.c0 && extra * 10 <= sum);
      if (!ok) continue;
      Front f;
      f.c0 = s.c0; f.w = w;
      for (int c = 0; c < s.w; c++) f.rows.push_back(s.c0 + c);
      f.rows.insert(f.rows.end(), p.rows.begin(), p.rows.end());
      fr[i] = f;
      fr.erase(fr.begin() + i + 1);
      changed = true;
      break;
    }
  }
  const int nfront = (int)fr.size();
  std::vector<int> colfront(ne > 0 ? ne : 1, 0);
  for (int f = 0; f < nfront; f++) for (int c = 0; c < fr[f].w; c++) colfront[fr[f].c0 + c] = f;
  // levels: a front after all the fronts it receives updates from
  for (int f = 0; f < nfront; f++) {
    const int last = fr[f].c0 + fr[f].w - 1;
    fr[f].parent = ppar[last] >= 0 ? colfront[ppar[last]] : -1;
  }
  for (int f = 0; f < nfront; f++) if (fr[f].parent >= 0) fr[fr[f].parent].level = std::max(fr[fr[f].parent].level, fr[f].level + 1); // (postorder: children first)
  int nlevel = 0;
  for (auto &f : fr) nlevel = std::max(nlevel, f.level + 1);
  std::vector<int> forder(nfront);
  std::iota(forder.begin(), forder.end(), 0);
  std::stable_sort(forder.begin(), forder.end(), [&](int a, int b) { // by level; the wide fronts first, the costly ones first; single columns last
    if (fr[a].level != fr[b].level) return fr[a].level < fr[b].level;
    if ((fr[a].w == 1) != (fr[b].w == 1)) return fr[b].w == 1;
    return fr[a].nnz() * fr[a].w > fr[b].nnz() * fr[b].w;
  });
  { // renumber the fronts by (level, cost descending)
    std::vector<Front> g(nfront);
    std::vector<int> newid(nfront);
    for (int k = 0; k < nfront; k++) { g[k] = fr[forder[k]]; newid[forder[k]] = k; }
    for (auto &f : g) if (f.parent >= 0) f.parent = newid[f.parent];
    fr.swap(g);
    for (int f = 0; f < nfront; f++) for (int c = 0; c < fr[f].w; c++) colfront[fr[f].c0 + c] = f;
  }
  std::vector<int> level_ptr(nlevel + 1, 0);
  for (auto &f : fr) level_ptr[f.level + 1]++;
  for (int l = 0; l < nlevel; l++) level_ptr[l + 1] += level_ptr[l];
  // ---- the value store: the fronts (column-major rectangles), then C = the trailing (mi + 1) x (mi + 1) triangle, packed
  int nval = 0, nrows_total = 0, maxw = 0, maxr = 0;
  for (auto &f : fr) { f.base = nval; nval += f.w * f.r(); f.rowoff = nrows_total; nrows_total += f.r(); maxw = std::max(maxw, f.w); maxr = std::max(maxr, f.r()); }
  const int Cb = nval, M1 = mi + 1;
  nval += M1 * (M1 + 1) / 2;
  auto c_addr = [&](int pr, int pc) { const int i = pr - ne, c = pc - ne; return Cb + c * M1 - (c * (c - 1)) / 2 + i - c; }; // pr >= pc >= ne
  auto f_local = [&](const Front &f, int pos) { // local row of a position in a front (-1: not there)
    if (pos >= f.c0 && pos < f.c0 + f.w) return pos - f.c0;
    auto it = std::lower_bound(f.rows.begin() + f.w, f.rows.end(), pos);
    return (it != f.rows.end() && *it == pos) ? (int)(it - f.rows.begin()) : -1;
  };
  auto s_addr = [&](int pr, int pc) -> int { // address of S[pr][pc], pr >= pc (-1: not in the structure)
    if (pc >= ne) return c_addr(pr, pc);
    const Front &f = fr[colfront[pc]];
    const int lr = f_local(f, pr);
    return lr < 0 ? -1 : f.base + (pc - f.c0) * f.r() + lr;
  };
  // ---- update jobs: target tiles, each with the source columns (slots) it sums over, in a fixed order
  struct Job { int tbase, tstride, trows, tcols, r0, c0; std::vector<int> srcs; std::vector<std::vector<int>> maps; long long cost = 0; bool urgent = false; };
  std::vector<std::vector<Job>> jobs(nlevel);
  for (int l = 0; l < nlevel; l++) {
    std::map<std::tuple<int, int, int>, int> index; // (target, tile row, tile column) -> job
    for (int f = level_ptr[l]; f < level_ptr[l + 1]; f++) {
      const Front &F = fr[f];
      for (int jb = F.w; jb < F.r(); jb++) {
        const int pc = F.rows[jb];
        if (pc >= m) continue; // (the right-hand side is a row, not a column)
        const int tg = pc < ne ? colfront[pc] : -1;
        const int lc = pc < ne ? pc - fr[tg].c0 : pc - ne;
        for (int ib = jb; ib < F.r(); ib++) {
          const int pr = F.rows[ib];
          const int lr = pc < ne ? f_local(fr[tg], pr) : pr - ne;
          if (lr < 0) return fail("internal: an update falls outside the structure of its target");
          auto key = std::make_tuple(tg, lr / 16, lc / 16);
          auto it = index.find(key);
          if (it == index.end()) {
            Job J;
            if (tg >= 0) { J.tbase = fr[tg].base; J.tstride = fr[tg].r(); J.trows = fr[tg].r(); J.tcols = fr[tg].w; J.urgent = fr[tg].level == l + 1; }
            else { J.tbase = Cb; J.tstride = 0; J.trows = M1; J.tcols = M1; }
            J.r0 = 16 * (lr / 16); J.c0 = 16 * (lc / 16);
            it = index.emplace(key, (int)jobs[l].size()).first;
            jobs[l].push_back(J);
          }
          Job &J = jobs[l][it->second];
          if (J.srcs.empty() || J.srcs.back() != f) { J.srcs.push_back(f); J.maps.push_back(std::vector<int>(32, -1)); }
          J.maps.back()[lr % 16] = ib;
          J.maps.back()[16 + lc % 16] = jb;
        }
      }
    }
    for (auto &J : jobs[l]) for (int f : J.srcs) J.cost += (fr[f].w + 3) / 4;
    // the jobs into the NEXT level's fronts first (those fronts wait for them; the others run beside them), the long ones first
    std::stable_sort(jobs[l].begin(), jobs[l].end(), [](const Job &a, const Job &b) { return a.urgent != b.urgent ? a.urgent : a.cost > b.cost; });
  }
  // ---- the sparse products of the solves (qpc_qp_spview.h): targets with their terms, cut into segments
  struct Op { std::vector<int> tpos; std::vector<std::vector<int>> terms; };
  auto pack = [](int addr, int idx) { return (addr << 9) | idx; };
  if (nval >= (1 << 20) || N1 > 511 || nnzA >= (1 << 22)) return fail("too large for the packed terms");
  std::vector<Op> ops(SP_OP_FWD0 + 2 * nlevel);
  for (int p = 0; p < m; p++) { // A w by row position
    const int r = perm[p];
    std::vector<int> t;
    for (int k = rptr[r]; k < rptr[r + 1]; k++) t.push_back(pack(cperm[ro[k]], rcol[k])); // (values are indexed in the CALLER's order: no permutation in the kernel)
    ops[SP_OP_ROWDOT].tpos.push_back(p); ops[SP_OP_ROWDOT].terms.push_back(t);
  }
  for (int c = 0; c < n; c++) { // A' y by column: (value, row position)
    std::vector<int> t;
    for (int o = cptr[c]; o < cptr[c + 1]; o++) t.push_back(pack(cperm[o], ipos[crow[o]]));
    ops[SP_OP_COLDOT].tpos.push_back(c); ops[SP_OP_COLDOT].terms.push_back(t);
  }
  auto below_terms_of_row = [&](int p) { // the entries of L in row p left of its own front: (address, column position), fronts in order
    std::vector<int> t;
    for (int f = 0; f < nfront; f++) {
      if (p >= fr[f].c0 && p < fr[f].c0 + fr[f].w) continue;
      const int lr = f_local(fr[f], p);
      if (lr >= 0) for (int c = 0; c < fr[f].w; c++) t.push_back(pack(fr[f].base + c * fr[f].r() + lr, fr[f].c0 + c));
    }
    return t;
  };
  for (int p = ne; p < m; p++) { ops[SP_OP_FWDROWS].tpos.push_back(p); ops[SP_OP_FWDROWS].terms.push_back(below_terms_of_row(p)); }
  for (int l = 0; l < nlevel; l++)
    for (int f = level_ptr[l]; f < level_ptr[l + 1]; f++)
      for (int c = 0; c < fr[f].w; c++) {
        const int p = fr[f].c0 + c;
        Op &F = ops[SP_OP_FWD0 + l], &B = ops[SP_OP_FWD0 + nlevel + l];
        F.tpos.push_back(p); F.terms.push_back(below_terms_of_row(p));
        std::vector<int> t; // the column's entries below its front's block: (address, row position)
        for (int i = fr[f].w; i < fr[f].r(); i++) if (fr[f].rows[i] < m) t.push_back(pack(fr[f].base + c * fr[f].r() + i, fr[f].rows[i])); // (the right-hand-side row is no unknown)
        B.tpos.push_back(p); B.terms.push_back(t);
      }
  std::vector<int> optab, opdata;
  int maxseg = 1;
  for (auto &O : ops) {
    const int nt = (int)O.tpos.size();
    std::vector<int> sptr(1, 0), segs(1, 0), terms;
    for (int t = 0; t < nt; t++) {
      const int len = (int)O.terms[t].size();
      for (int k = 0; k < len; k += SP_SEG) { terms.insert(terms.end(), O.terms[t].begin() + k, O.terms[t].begin() + std::min(len, k + SP_SEG)); segs.push_back((int)terms.size()); }
      sptr.push_back((int)segs.size() - 1);
    }
    const int nseg = (int)segs.size() - 1;
    maxseg = std::max(maxseg, nseg);
    optab.push_back(nt); optab.push_back(nseg);
    optab.push_back((int)opdata.size()); opdata.insert(opdata.end(), O.tpos.begin(), O.tpos.end());
    optab.push_back((int)opdata.size()); opdata.insert(opdata.end(), sptr.begin(), sptr.end());
    optab.push_back((int)opdata.size()); opdata.insert(opdata.end(), segs.begin(), segs.end());
    optab.push_back((int)opdata.size()); opdata.insert(opdata.end(), terms.begin(), terms.end());
  }
  // ---- G: its structural entries (r >= s by position) with the products they sum over, the longest lists first
  struct Pair { int addr; std::vector<int> t; };
  std::vector<Pair> pairs;
  {
    std::map<int, int> index; // address -> pair
    // a_rc (H^-1)_cd a_sd over the entries (c, d) of the blocks; entries of A in column order
    for (int c = 0; c < n; c++)
      for (int k = hptr[c]; k < hptr[c + 1]; k++) {
        const int d = hcol[k], hv = hidx[k];
        for (int o1 = cptr[c]; o1 < cptr[c + 1]; o1++)
          for (int o2 = cptr[d]; o2 < cptr[d + 1]; o2++) {
            const int pr = ipos[crow[o1]], ps = ipos[crow[o2]];
            if (pr < ps) continue; // (its mirror image is the term (d, c) of the entry (ps, pr); on the diagonal every ordered (c, d) is a term of its own)
            const int addr = s_addr(pr, ps);
            if (addr < 0) return fail("internal: an entry of G falls outside the structure");
            auto it = index.find(addr);
            if (it == index.end()) { it = index.emplace(addr, (int)pairs.size()).first; pairs.push_back(Pair{addr, {}}); }
            std::vector<int> &t = pairs[it->second].t;
            t.push_back(cperm[o1]); t.push_back(cperm[o2]); t.push_back(hv);
          }
      }
    std::stable_sort(pairs.begin(), pairs.end(), [](const Pair &a, const Pair &b) { return a.t.size() > b.t.size(); });
  }
  // ---- assemble
  std::vector<int> &I = out.ints;
  I.assign(SP_HEADER_INTS, 0);
  auto put = [&](int key, const std::vector<int> &v) { I[key] = (int)I.size(); I.insert(I.end(), v.begin(), v.end()); if (v.empty()) I.push_back(0); };
  put(SP_O_PERM, perm);
  put(SP_O_IPOS, std::vector<int>(ipos.begin(), ipos.begin() + m));
  {
    std::vector<int> ft, rows;
    int lev = -1, coloff = 0;
    for (auto &f : fr) { // (the last entry: where the front's columns start among its level's targets of the solves' sparse products)
      if (f.level != lev) { lev = f.level; coloff = 0; }
      ft.insert(ft.end(), {f.base, f.r(), f.w, f.c0, f.rowoff, coloff});
      coloff += f.w;
      rows.insert(rows.end(), f.rows.begin(), f.rows.end());
    }
    put(SP_O_FRONT, ft);
    put(SP_O_ROWS, rows);
  }
  {
    std::vector<int> lv2; // per level: its first front, its first single-column front; then the end
    for (int l = 0; l < nlevel; l++) {
      int fs = level_ptr[l];
      while (fs < level_ptr[l + 1] && fr[fs].w > 1) fs++;
      lv2.push_back(level_ptr[l]); lv2.push_back(fs);
    }
    lv2.push_back(nfront);
    put(SP_O_LEVEL, lv2);
  }
  {
    std::vector<int> joblv(1, 0), jt, slots, maps;
    std::map<std::vector<int>, int> map_ids;
    int njob = 0;
    for (int l = 0; l < nlevel; l++) {
      for (auto &J : jobs[l]) {
        const int s0 = (int)slots.size() / SP_SLOT_INTS;
        for (size_t k = 0; k < J.srcs.size(); k++) {
          const Front &F = fr[J.srcs[k]];
          // (fronts with the same row structure -- the joint rows of a humanoid, its contacts -- map a tile alike: a map is stored once)
          auto seen = map_ids.find(J.maps[k]);
          if (seen == map_ids.end()) {
            seen = map_ids.emplace(J.maps[k], (int)maps.size() / 16).first;
            for (int e = 0; e < 32; e += 2) maps.push_back((J.maps[k][e] & 0xffff) | (J.maps[k][e + 1] << 16)); // (16 bits each: a front has at most 2 QPC_SP_DIM rows; -1 stays -1)
          }
          const int mp = seen->second;
          for (int c = 0; c < F.w; c++) slots.insert(slots.end(), {(F.base + c * F.r()) | ((F.c0 + c) << 20), mp});
        }
        while ((slots.size() / SP_SLOT_INTS - s0) % 4) slots.insert(slots.end(), {0, -1});
        jt.insert(jt.end(), {J.tbase, J.tstride, J.trows, J.tcols, J.r0, J.c0, s0, (int)(slots.size() / SP_SLOT_INTS - s0) / 4});
        njob++;
      }
      int nurgent = 0;
      for (auto &J : jobs[l]) nurgent += J.urgent;
      joblv.back() = njob - (int)jobs[l].size(); // (start of the level's jobs)
      joblv.push_back(njob - (int)jobs[l].size() + nurgent); // start of its jobs that nobody waits for
      joblv.push_back(njob);
    }
    out.njob = njob;
    put(SP_O_JOBLV, joblv);
    put(SP_O_JOB, jt);
    put(SP_O_SLOT, slots);
    put(SP_O_MAP, maps);
  }
  put(SP_O_OPS, optab);
  put(SP_O_OPDATA, opdata);
  put(SP_O_COLFRONT, colfront);
  put(SP_O_HPTR, hptr); put(SP_O_HCOL, hcol); put(SP_O_HIDX, hidx); put(SP_O_PIDX, pidx);
  put(SP_O_BPTR, bptr); put(SP_O_BVAR, bvar); put(SP_O_BBASE, bbase);
  I[SP_LDS_INTS] = (int)I.size();
  { // the entries' term lists cut into segments of at most SP_SEG terms (a thread each): an entry with one segment is stored by that
    // thread, an entry with several is summed from the segments' sums by a second pass
    std::vector<int> gseg, gsum, tt;
    int nterm = 0, npart = 0;
    for (auto &p : pairs) {
      const int len = (int)p.t.size() / 3, nsg = (len + SP_SEG - 1) / SP_SEG;
      if (nsg > 1) { gsum.push_back(p.addr); gsum.push_back(npart); gsum.push_back(npart + nsg); }
      for (int k = 0; k < len; k += SP_SEG) {
        gseg.push_back(nsg > 1 ? -(npart++) - 1 : p.addr);
        gseg.push_back(nterm + k);
      }
      for (int k = 0; k < len; k++) { tt.push_back(p.t[3 * k] | (p.t[3 * k + 1] << 16)); tt.push_back(p.t[3 * k + 2]); }
      nterm += len;
    }
    gseg.push_back(0); gseg.push_back(nterm);
    I[SP_NPAIR] = (int)pairs.size(); I[SP_NTERM] = nterm; I[SP_NGSEG] = (int)gseg.size() / 2 - 1; I[SP_NGSUM] = (int)gsum.size() / 3;
    maxseg = std::max(maxseg, npart);
    put(SP_O_GSEG, gseg);
    put(SP_O_GSUM, gsum);
    put(SP_O_TERM, tt);
  }
  {
    std::vector<int> ra(m);
    for (int p = 0; p < m; p++) { ra[p] = s_addr(m, p); if (ra[p] < 0) return fail("internal: the right-hand-side row is not in a column's structure"); }
    put(SP_O_RTADDR, ra);
  }
  I[SP_MAGIC] = QPC_SP_MAGIC;
  I[SP_N] = n; I[SP_M] = m; I[SP_NE] = ne; I[SP_MI] = mi; I[SP_NNZA] = (int)nnzA; I[SP_NNZP] = (int)nnzP;
  I[SP_NFRONT] = nfront; I[SP_NLEVEL] = nlevel; I[SP_NJOB] = out.njob; I[SP_NVAL] = nval; I[SP_CB] = Cb; I[SP_NHV] = nhv; I[SP_NBLK] = nblk;
  I[SP_NOPS] = (int)ops.size(); I[SP_MAXSEG] = maxseg;
  I[SP_NRT] = (int)nnzA; I[SP_TOTAL_INTS] = (int)I.size(); I[SP_MAXW] = maxw; I[SP_MAXR] = maxr;
  out.nval = nval; out.nhv = nhv; out.nfront = nfront; out.nlevel = nlevel; out.lds_ints = I[SP_LDS_INTS]; out.maxw = maxw; out.maxr = maxr;
  out.valid = true;
  return true;
}

} // namespace qpc
