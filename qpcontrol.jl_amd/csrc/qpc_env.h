// qpc_env.h -- which environment variables the library reads (host code only)
#pragma once
#include <cstdlib>
// Experiment switches (A/B toggles of algorithmic choices: QPC_NO_BASE_PIN, QPC_NO_KTAB, QPC_K_CLOSED, QPC_DFS_SCAN, QPC_NO_PRESOLVE_FOLD,
// QPC_MAX_PDAS, QPC_SIM_RPW, QPC_F32_LDS, QPC_EXP_*, QPC_JIT_OPT / _SCHED / _EXTRA_FLAGS, QPC_STAGE_ZEROCOPY) are read from the
// environment ONLY in a build with -DQPC_EXPERIMENTS (libqpcontrol_hip_exp.so, the emulator of tests/emul): the product library
// ignores them.  What the product library does read is listed in include/qpc.h ("environment").
#ifdef QPC_EXPERIMENTS
#define QPC_EXP_ENV(name) getenv(name)
#else
#define QPC_EXP_ENV(name) ((const char *)nullptr)
#endif

