// Eight-wave split recurrence, three bf16 pieces / six products (gate-GEMM mode 1): 64-, 80- and 96-sequence tiles.
#include "recur_split8.h"

int recur::launch_split8_bf16(const RecurArgs& a, int nsr, bool fast, hipStream_t st) {
    using namespace rsplit;
    switch (nsr) {
        case 80: return launch_split8_form<80, false>(a, fast, st);
        case 96: return launch_split8_form<96, false>(a, fast, st);
        default: return launch_split8_form<64, false>(a, fast, st);
    }
}
