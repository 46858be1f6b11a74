// Eight-wave split recurrence, two f16 pieces / three products (gate-GEMM mode 3, the default): 16- to 96-sequence tiles.
#include "recur_split8.h"

int recur::launch_split8_f16(const RecurArgs& a, int nsr, bool fast, hipStream_t st) {
    using namespace rsplit;
    switch (nsr) {
        case 16: return launch_split8_form<16, true>(a, fast, st);
        case 32: return launch_split8_form<32, true>(a, fast, st);
        case 48: return launch_split8_form<48, true>(a, fast, st);
        case 80: return launch_split8_form<80, true>(a, fast, st);
        case 96: return launch_split8_form<96, true>(a, fast, st);
        default: return launch_split8_form<64, true>(a, fast, st);
    }
}

// training forward (dprnn_rnn_fwd_train in mode 3): 64-sequence tiles, gate tape in the fp32 training kernel's layout
int recur::launch_split8_f16_train(const RecurArgs& a, bool fast, hipStream_t st) {
    using namespace rsplit;
    const bool fuse = a.Z != nullptr;
    if (fast) return fuse ? launch_split8<64, 3, true, true, false, true, true>(a, st) : launch_split8<64, 3, true, false, false, true, true>(a, st);
    return fuse ? launch_split8<64, 3, false, true, false, true, true>(a, st) : launch_split8<64, 3, false, false, false, true, true>(a, st);
}
