// qpc_qp_spview.h -- layout of the PLAN of a prepared L0 problem: what the symbolic analysis on the host (qpc_qp_sym.h) hands to the
// sparse workgroup solver (qpc_qp_sp.h).  One int array: a header of dimensions and section offsets, then the sections.  The first
// plan[SP_LDS_INTS] ints are what the kernel keeps in LDS for the life of a workgroup; the rest (the term lists of G = A H^-1 A',
// streamed once per solve) stay in global memory.
#pragma once

namespace qpc {

enum QpSpHeader {
  SP_MAGIC = 0,
  SP_N, SP_M, SP_NE, SP_MI,   // variables, rows, equality rows, the others
  SP_NNZA, SP_NNZP,           // entries of the caller's value arrays
  SP_NFRONT, SP_NLEVEL, SP_NJOB,
  SP_NVAL,                    // doubles of the value store: the fronts, then C (packed lower (mi + 1) x (mi + 1))
  SP_CB,                      // offset of C in the value store
  SP_NHV,                     // values of H^-1 (and of H): one per structural entry of the block-diagonal P, full blocks
  SP_NBLK,                    // diagonal blocks of P
  SP_NPAIR, SP_NTERM,         // structural entries of G on and below the diagonal; the products they sum over
  SP_NRT,                     // products of the right-hand-side row (= nnzA)
  SP_LDS_INTS, SP_TOTAL_INTS, // what goes to LDS; the whole array
  SP_MAXW, SP_MAXR,           // widest front, tallest front
  // sections kept in LDS
  SP_O_PERM,    // m + 1: position -> row (position m: the right-hand-side row)
  SP_O_IPOS,    // m: row -> position
  SP_O_FRONT,   // nfront x SP_FRONT_INTS
  SP_O_ROWS,    // the fronts' row positions (a front's own columns first)
  SP_O_LEVEL,   // nlevel + 1: fronts of a level (the fronts are sorted by level)
  SP_O_JOBLV,   // nlevel + 1: update jobs of a level
  SP_O_JOB,     // njob x SP_JOB_INTS
  SP_O_SLOT,    // x SP_SLOT_INTS: the columns an update job sums over, padded to multiples of four per job
  SP_O_MAP,     // x 32: target row / column of a job's tile -> row of the source front (or -1)
  SP_O_PULLPTR, // m + 2: for a position, the fronts that hold it among their rows below the diagonal block ...
  SP_O_PULL,    // x 2: ... (front, local row)
  SP_O_COLFRONT,// ne: the front of an equality position
  SP_O_HPTR,    // n + 1: the block row of a variable in H^-1 ...
  SP_O_HCOL,    // ... column
  SP_O_HIDX,    // ... index of the value (in hv; the caller's P value: SP_O_PIDX)
  SP_O_PIDX,
  SP_O_BPTR,    // nblk + 1: variables of a block (SP_O_BVAR), its values start at hv[SP_O_BBASE[b]], row-major bs x bs
  SP_O_BVAR,
  SP_O_BBASE,
  SP_O_CPTR,    // n + 1: compressed columns of A ...
  SP_O_CROW,    // ... POSITION of the entry's row
  SP_O_RPTR,    // m + 1: compressed rows of A by POSITION ...
  SP_O_RCOL,    // ... column
  SP_O_RO,      // ... index of the value in column order
  // sections left in global memory
  SP_O_CPERM,   // nnzA: column-order entry -> index in the caller's A values
  SP_O_PAIR,    // npair x 2: address in the value store, end of its terms (the first pair's terms start at 0); longest first
  SP_O_TERM,    // nterm x 3: column-order indices of the two A entries, index of the H^-1 value
  SP_O_RTADDR,  // m: address of the right-hand-side row's entry at a position
  SP_HEADER_INTS = 64
};
enum { SP_FRONT_INTS = 6 }; // base, rows, width, first position, offset of its row positions, (spare)
enum { SP_JOB_INTS = 8 };   // target base, target stride (0: packed C), target rows, target columns, first row, first column, first slot, chunks
enum { SP_SLOT_INTS = 3 };  // address of the source column, position of its pivot, map (-1: a padding slot)
#define QPC_SP_MAGIC 0x51505350

} // namespace qpc
