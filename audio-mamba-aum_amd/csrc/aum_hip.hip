// aum_hip.hip -- device translation unit of libaum_hip.so (gfx950).  The library is compiled as several
// objects from this one file (build.py): -DAUM_API_PART=1/2 with -DAUM_DTYPE_ONLY=0/1/2 give the scan
// forward / backward kernels per activation dtype, -DAUM_API_PART=3 gives the extern "C" surface with the
// conv and norm kernels.  No torch headers, no host framework: plain HIP + the C ABI of include/aum_hip.h.
#include "aum_api.inc"
