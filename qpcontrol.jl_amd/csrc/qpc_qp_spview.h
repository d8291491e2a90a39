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
  SP_NOPS, SP_MAXSEG,         // sparse products; the most segments one of them has
  SP_NGSEG, SP_NGSUM,         // segments of G's term lists; entries of G with more than one
  // sections kept in LDS
  SP_O_PERM,    // m + 1: position -> row (position m: the right-hand-side row)
  SP_O_IPOS,    // m: row -> position
  SP_O_FRONT,   // nfront x SP_FRONT_INTS
  SP_O_ROWS,    // the fronts' row positions (a front's own columns first)
  SP_O_LEVEL,   // 2 nlevel + 1: fronts of level l (sorted by level): [2 l] .. [2 l + 1] of several columns, [2 l + 1] .. [2 l + 2] of one
  SP_O_JOBLV,   // 2 nlevel + 1: update jobs of level l: [2 l] .. [2 l + 1] into the fronts of level l + 1, [2 l + 1] .. [2 l + 2] the others
  SP_O_JOB,     // njob x SP_JOB_INTS
  SP_O_SLOT,    // x SP_SLOT_INTS: the columns an update job sums over, padded to multiples of four per job
  SP_O_MAP,     // x 16 (32 x 16 bits): target row / column of a job's tile -> row of the source front (or -1)
  SP_O_OPS,     // SP_NOPS x SP_OP_INTS: the sparse products (below)
  SP_O_OPDATA,  // their tables
  SP_O_COLFRONT,// ne: the front of an equality position
  SP_O_HPTR,    // n + 1: the block row of a variable in H^-1 ...
  SP_O_HCOL,    // ... column
  SP_O_HIDX,    // ... index of the value (in hv; the caller's P value: SP_O_PIDX)
  SP_O_PIDX,
  SP_O_BPTR,    // nblk + 1: variables of a block (SP_O_BVAR), its values start at hv[SP_O_BBASE[b]], row-major bs x bs
  SP_O_BVAR,
  SP_O_BBASE,
  // sections left in global memory
  SP_O_GSEG,    // (SP_NGSEG + 1) x 2: a segment of at most SP_SEG products of an entry of G: address in the value store (or -1 - the slot of
                // its partial sum: an entry with several segments), its first term (the next segment's: its end); longest entries first
  SP_O_GSUM,    // SP_NGSUM x 3: an entry with several segments: address, its partial sums' slots [first, end)
  SP_O_TERM,    // nterm x 2: the indices of the two A entries in the caller's values (16 bits each), the index of the H^-1 value
  SP_O_RTADDR,  // m: address of the right-hand-side row's entry at a position
  SP_HEADER_INTS = 64
};
enum { SP_FRONT_INTS = 6 }; // base, rows, width, first position, offset of its row positions, first target among its level's in the solves' products
enum { SP_JOB_INTS = 8 };   // target base, target stride (0: packed C), target rows, target columns, first row, first column, first slot, chunks
enum { SP_SLOT_INTS = 2 };  // address of the source column | position of its pivot << 20, map (-1: a padding slot)
// A sparse product: out[target] = sum of val[addr] * vec[idx] over the target's terms, a term packed as (addr << 9) | idx.  What bounds
// such a sum on one thread is the latency of its dependent LDS loads (index -> value), so a target's terms are cut into SEGMENTS of at
// most SP_SEG terms: a thread per segment, then a thread per target adds its segments' sums in order -- the 76 terms of a wrench
// balance row take as long as the 16 of any other.
enum { SP_OP_INTS = 6 };    // targets, segments, offset of: the targets' positions | their segment ranges (targets + 1) | the segments' term ranges (segments + 1) | the terms
enum { SP_SEG = 16 };
enum QpSpOp { // operations, in this order; per level L of the tree (0 .. nlevel - 1): SP_OP_FWD0 + L, then SP_OP_FWD0 + nlevel + L
  SP_OP_ROWDOT = 0, // A w at every row position: val = A's values (the caller's order), vec = w by column
  SP_OP_COLDOT,     // A' y at every column: vec = y by position
  SP_OP_FWDROWS,    // the inequality rows' part of a forward solve: val = the store, vec = the scaled solution of the equality columns
  SP_OP_FWD0        // ... the level's columns pulled along before their own fronts' solves; then the back solves' gathers: vec = y by position
};
#define QPC_SP_MAGIC 0x51505350

} // namespace qpc
