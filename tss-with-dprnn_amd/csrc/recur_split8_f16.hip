// Eight-wave split recurrence, f16 pieces (gate-GEMM mode 3, the default): 32- to 96-sequence tiles.
#include "recur_split8.h"

int recur::launch_split8_f16(const RecurArgs& a, int nsr, bool stag, bool fast, hipStream_t st) {
    using namespace rsplit;
#ifdef DPRNN_STAG_FORMS        // staggered forms: 17 spilled registers at 64 sequences and four exposed HBM waits per step, not built
    if (stag) return nsr == 48 ? launch_split8_form<48, true, true>(a, 2, fast, st) : launch_split8_form<64, true, true>(a, 2, fast, st);
#endif
    (void)stag;
    switch (nsr) {
        case 16: return launch_split8_form<16, false, true>(a, 2, fast, st);
        case 32: return launch_split8_form<32, false, true>(a, 2, fast, st);
        case 48: return launch_split8_form<48, false, true>(a, 2, fast, st);
        case 80: return launch_split8_form<80, false, true>(a, 2, fast, st);
        case 96: return launch_split8_form<96, false, true>(a, 2, fast, st);
        default: return launch_split8_form<64, false, true>(a, 2, fast, st);
    }
}

// training forward (dprnn_rnn_fwd_train in mode 3): 64-sequence tiles, gate tape in the fp32 training kernel's layout
int recur::launch_split8_f16_train(const RecurArgs& a, bool fast, hipStream_t st) {
    using namespace rsplit;
    const bool fuse = a.Z != nullptr;
    if (fast) return fuse ? launch_split8<64, 3, true, true, false, false, true, true>(a, st) : launch_split8<64, 3, true, false, false, false, true, true>(a, st);
    return fuse ? launch_split8<64, 3, false, true, false, false, true, true>(a, st) : launch_split8<64, 3, false, false, false, false, true, true>(a, st);
}
