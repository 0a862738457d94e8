// TEST-ONLY: counters of the fibre shim (hip/hip_runtime.h), one definition per shim-built library.
#include <hip/hip_runtime.h>

// out[0..8]: fibre switches, cross-lane operations (per wave), workgroup barriers, cross-lane operations released with a
// subset of the wave (divergent control flow), kernel launches, launches abandoned because lock step could not be set up;
// lock-step mode: page faults taken, intervals merged, bytes two lanes left with different values
extern "C" void snf_simt_counters(unsigned long long* out) {
  out[0] = simt::g_n_switch; out[1] = simt::g_n_wave_ops; out[2] = simt::g_n_block_syncs; out[3] = simt::g_n_divergent_ops;
  out[4] = simt::g_n_launches; out[5] = simt::g_n_unmodelled;
  out[6] = simt::g_n_lockstep_faults; out[7] = simt::g_n_lockstep_merges; out[8] = simt::g_n_lockstep_conflicts;
}
extern "C" unsigned long long snf_simt_unmodelled() { return simt::g_n_unmodelled; }
