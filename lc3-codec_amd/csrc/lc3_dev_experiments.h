// Timing experiments -- the one home of every switch that exists only to PRICE a piece of a kernel.  None of them is set in a
// production build (all default to 0, every use folds away at compile time); a build with one of them set is made with
//   LC3_HIPCC_EXTRA="-DLC3_ENC_KO=512" LC3GPU_LIB=liblc3gpu_x.so python -c "...build_native(force=True)"     (tools/exp_ko.sh)
// and is only ever timed: the output of a knock-out build is garbage, the output of a repeat build is the production output.
//
// Two kinds:
//   knock-out  LC3_KO(set, bit)        true in a build whose `set` has `bit`: the piece is skipped.  Cheap to write, but whatever follows
//                                      the piece sees other data (a skipped gain search changes what the quantiser does next).
//   repeat     LC3_EXP_REPS(set, bit)  the number of times a piece runs: 1, or 2 in a build whose `set` has `bit` (the 2 is opaque to the
//                                      compiler, so the second run is neither folded into the first nor hoisted).  Only for pieces that
//                                      are idempotent -- same inputs, same stores --: the data every later piece sees is the production
//                                      data, and the time added is the piece's price beside everything else the kernel does.
#pragma once

// ---- encoder, wave-per-stream halves (lc3_dev_enc.h) ----
// LC3_ENC_KO (stage level, at the stages' call sites): 1 no MDCT, 2 no bandwidth detector, 4 no attack detector, 8 no SNS targets,
//   16 no LTPF analysis, 256 no TNS, 512 no quantiser, 1024 no residual / noise stage.
// (Settled and removed in round 5, their results are in profiles/r04_knockout_*.txt and DESIGN.md: the switches inside the LTPF stage,
//   the synthesis kernel's LC3_DEC_KO, the wave-per-frame reconstruction's LC3_RECON_KO / LC3_TNS_KO and the "one half only" builds of the
//   producer / consumer pairs, LC3_PCPARSE_KO / LC3_PKPC_KO, which sat inside the unrolled symbol loops.)
#ifndef LC3_ENC_KO
#define LC3_ENC_KO 0
#endif
// LC3_ENC_DUP (repeat builds of the back half): 1 the quantiser's group energies (100 four-line sums + log10), 2 its gain bisection (8 steps),
//   4 its first quantise + bit-count pass, 8 its second pass (in the frames that take one), 16 the TNS partial autocorrelations + quotients,
//   32 the TNS Levinson recursion / reflection coefficients, 64 the residual-bit / noise-level stage, 128 the plane pick-up's shaping pass
//   (LDS side only: the loads are not repeated)
#ifndef LC3_ENC_DUP
#define LC3_ENC_DUP 0
#endif
#define LC3_KO(set, bit) (((set) & (bit)) != 0)
#ifndef LC3_EXP_TWO  // lc3gpu.hip: a 2 the compiler cannot see through
#define LC3_EXP_TWO() 2
#endif
#ifndef LC3_EXP_CLOBBER
#define LC3_EXP_CLOBBER() ((void)0)
#endif
#define LC3_EXP_REPS(set, bit) ((((set) & (bit)) != 0) ? LC3_EXP_TWO() : 1)
// in front of a statement: run it once more per set bit.  A production build sees NOTHING here (not even a loop of one trip: the
// headline kernels sit at the edge of their register budgets and a different statement structure moves the allocation)
#if LC3_ENC_DUP
// (the memory clobber after every run: what the piece loads is loaded again, what it stores is stored again -- without it the compiler
// hoists a piece without side effects out of the loop and the repeat costs nothing)
#define LC3_ENC_REPEAT(bit) for (int rep_ = LC3_EXP_REPS(LC3_ENC_DUP, bit); rep_ > 0; rep_--, LC3_EXP_CLOBBER())
#define LC3_ENC_REPEAT_MORE(bit) for (int rep_ = LC3_EXP_REPS(LC3_ENC_DUP, bit); rep_ > 1; rep_--, LC3_EXP_CLOBBER())  // the statement is a second copy
#else
#define LC3_ENC_REPEAT(bit)
#define LC3_ENC_REPEAT_MORE(bit) if (false)
#endif
