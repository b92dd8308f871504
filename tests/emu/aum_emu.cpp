// TEST INFRASTRUCTURE ONLY.  Lane-array build of the kernel sources in audio-mamba-aum_amd/csrc: every
// wavefront is stepped on the host with gfx950 lane semantics (csrc/wave.h, AUM_EMU) so the kernels' index
// arithmetic, tails, carries and reductions can be checked against the oracle on a machine with no GPU.
// Exports the same C ABI as libaum_hip.so but takes HOST pointers.  Never loaded by the product path.
#define AUM_EMU 1
#include "../../audio-mamba-aum_amd/csrc/aum_api.inc"
