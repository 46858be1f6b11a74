// Eight-wave split recurrence, bf16 pieces (gate-GEMM modes 1 and 2): 64- and 80-sequence tiles.
#include "recur_split8.h"

int recur::launch_split8_bf16(const RecurArgs& a, int products, int nsr, bool stag, bool fast, hipStream_t st) {
    using namespace rsplit;
#ifdef DPRNN_STAG_FORMS        // staggered forms (cell update of one wave group beside the x half of the other): slower, not built
    if (stag) return nsr == 48 ? launch_split8_form<48, true, false>(a, products, fast, st) : launch_split8_form<64, true, false>(a, products, fast, st);
#endif
    (void)stag;
    switch (nsr) {
        case 80: return launch_split8_form<80, false, false>(a, products, fast, st);
        case 96: return launch_split8_form<96, false, false>(a, products, fast, st);
        default: return launch_split8_form<64, false, false>(a, products, fast, st);
    }
}
