// qpc_qp_view.h -- the pattern of a prepared L0 problem as the kernel sees it (built by qpc_qp_prep.h on the host)
#pragma once

namespace qpc {

// what the kernel sees (device pointers in the runtime, host pointers in the emulation): sections of one int array
struct QpPrepView {
  int n, m, nnzA, npairs;
  const int *pdiag;  // n: index of P_ii in the caller's P values, or -1
  const int *cptr;   // n + 1: column starts
  const int *crow;   // nnzA: row of the entry (column order, rows ascending inside a column)
  const int *cperm;  // nnzA: its index in the caller's A values
  const int *rptr;   // m + 1: row starts
  const int *rcol;   // nnzA: column of the entry (row order)
  const int *ro;     // nnzA: its index in column order
  const int *pptr;   // npairs + 1
  const int *pr, *ps; // npairs: rows r >= s of a structurally nonzero G_rs
  const int *ta, *tb, *tc; // the terms of a pair: column-order indices of a_rc and a_sc, and c
};

} // namespace qpc
