// LDS-ring backward of the SVAE E-step, Student-t theta (kernel: vmp_svae_ring.h; reference models/svae.py:265-322,
// distributions/student_t.py:7-39).  Built with -fno-slp-vectorize (Makefile): see vmp_svae_ring.hip.
#include "vmp_svae_ring.h"

namespace vmp {
int svae_bwd_ring_launch_t(const EBwdArgs& a, int L, int nblk_abi, void* stream) {
    const bool k16 = a.K == 16;
    switch (L) {
        case 4: return k16 ? launch<4, true, true>(a, nblk_abi, stream) : launch<4, false, true>(a, nblk_abi, stream);
        case 6: return k16 ? launch<6, true, true>(a, nblk_abi, stream) : launch<6, false, true>(a, nblk_abi, stream);
        case 8: return k16 ? launch<8, true, true>(a, nblk_abi, stream) : launch<8, false, true>(a, nblk_abi, stream);
        default: return -2;
    }
}
}  // namespace vmp
