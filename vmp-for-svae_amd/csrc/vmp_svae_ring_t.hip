// LDS-ring backward of the SVAE E-step, Student-t theta (kernel: vmp_svae_ring.h; reference models/svae.py:265-322,
// distributions/student_t.py:7-39).  Built with -fno-slp-vectorize (Makefile): see vmp_svae_ring.hip.
#include "vmp_svae_ring.h"

namespace vmp {
int svae_bwd_ring_launch_t(const EBwdArgs& a, int L, int nblk_abi, void* stream) {
    // K = 16 and the reference's K = 10 (experiments.py: nb_components of the Auto / pinwheel schedules) have their own instances
    // (compile-time lane maps); any other 8 <= K <= 15 runs the run-time-K form
    const int ks = a.K == 16 ? 16 : (a.K == 10 && L == 8) ? 10 : 0;
    switch (L) {
        case 4: return ks == 16 ? launch<4, 16, true>(a, nblk_abi, stream) : launch<4, 0, true>(a, nblk_abi, stream);
        case 6: return ks == 16 ? launch<6, 16, true>(a, nblk_abi, stream) : launch<6, 0, true>(a, nblk_abi, stream);
        case 8: return ks == 16 ? launch<8, 16, true>(a, nblk_abi, stream) : ks == 10 ? launch<8, 10, true>(a, nblk_abi, stream) : launch<8, 0, true>(a, nblk_abi, stream);
        default: return -2;
    }
}
}  // namespace vmp
