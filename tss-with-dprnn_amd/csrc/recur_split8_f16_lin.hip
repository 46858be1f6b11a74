// Direction-serial BiLSTM with the following Linear folded in (gate-GEMM mode 3, recur_split8.h: PJ): instantiations, the
// Linear's pack kernel and the C ABI (dprnn_lin_pack_f16, dprnn_rnn_lin_fwd).
//
// Why (profiles/r04_pmc_traffic_f16.json): with both directions in one launch and the Linear as its own kernel a
// DPRNN path moves 10 slab units of 512 B per row through HBM -- x and z read by each direction (4), x' written (1), h written by
// both directions (2) and read back by the projection (2), z written (1) -- and three of them exist only because h_fwd[t] and
// h_bwd[t] are formed in different workgroups at different times.  Run the directions one after the other and let each multiply
// its own h with its half of the Linear while h is still in LDS:
//     launch 1 (forward direction):  reads x, z (pending norm), writes x' and the partial product p = W[:, :H] h_fwd      2 R + 2 W
//     launch 2 (reverse direction):  reads x' and p, writes z = p + W[:, H:] h_bwd + b, adds the GroupNorm sums           2 R + 1 W
// 7 units, no projection kernel, no cross-workgroup synchronisation (the kernel boundary orders the two), and p lives in the
// buffer that held the pending z and will hold the new one (launch 1 overwrites a row of z after its own step t + 1 has read it;
// launch 2 rewrites every element it reads from the same lane).  reference: dprnn.py:84-99 (SingleRNN + Linear per path).
#include "recur_split8.h"
#include "dprnn_mi355x.h"
using namespace recur;
using namespace rsplit;

namespace {

// One launch: every block finds max|W| itself (128 KB from L2), then packs its share.  Fragment of (dir, wave, slab, piece,
// lane (r, q)): the 8 f16 pieces of S_w W[16 wave + r][dir H + 32 slab + 8 q + j] -- the ROW operand of v_mfma_f32_16x16x32_f16,
// output columns as rows, like the gate weights of the TRANS forms.
__global__ __launch_bounds__(256)
void lin_pack_f16_kernel(const float* __restrict__ W, int ndir, unsigned* __restrict__ packed) {
    __shared__ float part[4];
    const int kd = ndir * HD;
    float m = 0.f;
    const f32x4* p = reinterpret_cast<const f32x4*>(W);
    for (int i = threadIdx.x; i < F * kd / 4; i += 256) {
        const f32x4 v = p[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    int ex = 0;
    if (m > 0.f && m < 3.0e38f) frexpf(m, &ex);
    ex = ex < -100 ? -100 : ex > 100 ? 100 : ex;
    const float sw = ldexpf(1.f, 14 - ex);
    const int idx = blockIdx.x * 256 + threadIdx.x;        // one thread per (dir, wave, slab, lane, dword)
    if (idx == 0) {
        reinterpret_cast<float*>(packed)[LINP_SCALE_WORD] = sw;
        reinterpret_cast<float*>(packed)[LINP_SCALE_WORD + 1] = 1.f / (F16_XS * sw);
    }
    if (idx >= ndir * 8 * 4 * 64 * 4) return;
    const int dw = idx & 3, lane = (idx >> 2) & 63, slab = (idx >> 8) & 3, wave = (idx >> 10) & 7, dir = idx >> 13;
    const int r = lane & 15, q = lane >> 4;
    const int k = dir * HD + 32 * slab + 8 * q + 2 * dw;
    const float* src = W + (size_t)(16 * wave + r) * kd + k;
    unsigned w1, w2;
    split2_pair_f16(sw * src[0], sw * src[1], w1, w2);
    const size_t base = ((((size_t)(dir * 8 + wave) * 4 + slab) * 2) * 64 + lane) * 4 + dw;
    packed[base] = w1;
    packed[base + 256] = w2;
}

template <int NSR>
int launch_lin(const RecurArgs& a, bool fast, hipStream_t st) {
    const bool fuse = a.Z != nullptr;
    (void)fast;                                            // hardware activations only (the libm forms keep the separate projection)
    if (a.proj == 1) return fuse ? launch_split8<NSR, 3, true, true, false, true, false, 1>(a, st) : launch_split8<NSR, 3, true, false, false, true, false, 1>(a, st);
    return launch_split8<NSR, 3, true, false, false, true, false, 2>(a, st);
}

}  // namespace

extern "C" int64_t dprnn_lin_pack_floats(void) { return LINP_FLOATS; }

extern "C" int dprnn_lin_pack_f16(const float* W, int ndir, void* packed, void* stream) {
    if (!W || !packed || (ndir != 1 && ndir != 2)) { dprnn_set_error("dprnn_lin_pack_f16: null pointer or ndir=%d", ndir); return -1; }
    const int n = ndir * 8 * 4 * 64 * 4;
    hipLaunchKernelGGL(lin_pack_f16_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, ndir, (unsigned*)packed);
    return dprnn_check_launch("lin_pack_f16_kernel");
}

extern "C" int dprnn_rnn_lin_fwd(int phase, int dir, const float* X, const float* Z, const double* stats, const float* gamma,
                                 const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                                 const float* packed, const float* bsum, int ndir, const void* lin_packed, const float* lin_bias,
                                 float* P, float* Zout, double* stats_out,
                                 int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride, int64_t t_stride,
                                 int seq_tile, void* stream) {
    if (!X || !packed || !bsum || !lin_packed) { dprnn_set_error("dprnn_rnn_lin_fwd: null pointer"); return -1; }
    if ((phase != 1 && phase != 2) || (ndir != 1 && ndir != 2) || dir < 0 || dir >= ndir) {
        dprnn_set_error("dprnn_rnn_lin_fwd: phase %d (1 partial products, 2 finished rows), direction %d of %d", phase, dir, ndir);
        return -1;
    }
    if (phase == 1 && !P) { dprnn_set_error("dprnn_rnn_lin_fwd: phase 1 writes its partial products to P"); return -1; }
    if (phase == 2 && (!Zout || !stats_out || !lin_bias)) { dprnn_set_error("dprnn_rnn_lin_fwd: phase 2 needs Zout, stats_out and the Linear's bias"); return -1; }
    if (nseq <= 0 || T <= 0 || inner_n <= 0 || seqs_per_utt <= 0 || rows_per_utt <= 0) {
        dprnn_set_error("dprnn_rnn_lin_fwd: bad sizes nseq=%d T=%d inner_n=%d seqs_per_utt=%d rows_per_utt=%lld", nseq, T, inner_n, seqs_per_utt, (long long)rows_per_utt);
        return -1;
    }
    if (Z && (!stats || !gamma || !beta || !Xout || Xout == X)) {
        dprnn_set_error("dprnn_rnn_lin_fwd: the pending norm needs stats, gamma, beta and an Xout different from X");
        return -1;
    }
    if (Z && phase != 1) { dprnn_set_error("dprnn_rnn_lin_fwd: the pending norm belongs to phase 1 (direction 0 writes x'; a unidirectional LSTM keeps dprnn_rnn_fwd_ex + dprnn_proj_stats_fwd)"); return -1; }
    const int nsr = seq_tile % 1000;
    if (seq_tile != 0 && (seq_tile < 2000 || seq_tile >= 3000 || (nsr != 32 && nsr != 48 && nsr != 64))) {
        dprnn_set_error("dprnn_rnn_lin_fwd: seq_tile 0, 2032, 2048 or 2064 (got %d)", seq_tile);
        return -1;
    }
    RecurArgs a{};
    a.X = X; a.Wp = packed; a.bsum = bsum; a.H = nullptr; a.nseq = nseq; a.T = T; a.inner_n = inner_n; a.ndir = ndir;
    a.outer_stride = outer_stride; a.inner_stride = inner_stride; a.t_stride = t_stride; a.cell = 0;
    a.Z = Z; a.stats = stats; a.gamma = gamma; a.beta = beta; a.eps = eps; a.seqs_per_utt = seqs_per_utt;
    a.rows_per_utt = rows_per_utt; a.Xout = Xout;
    a.proj_dirs = 1;
    a.split_m = -1;
    a.proj = phase; a.dir0 = dir; a.Lp = lin_packed; a.lbias = lin_bias; a.P = P; a.Zout = Zout; a.stats_out = stats_out;
    if (!tile_offsets_fit(inner_n, outer_stride, inner_stride, t_stride, F)) {
        dprnn_set_error("dprnn_rnn_lin_fwd: a sequence tile spans more than 2 GiB (outer_stride=%lld rows)", (long long)outer_stride);
        return -1;
    }
    if (int rc = dprnn_pack_expect(packed, DPRNN_PACK_FWD, 0, 3, ndir, "dprnn_rnn_lin_fwd")) return rc;
    a.sat = dprnn_sat_ptr();
    { const ScaleRef sr = dprnn_scale_ref(); a.xs_src = sr.src; a.xmax_dst = sr.dst; }
    hipStream_t st = (hipStream_t)stream;
    switch (seq_tile == 0 ? 64 : nsr) {
        case 32: return launch_lin<32>(a, true, st);
        case 48: return launch_lin<48>(a, true, st);
        default: return launch_lin<64>(a, true, st);
    }
}

// ---- the same fold, both directions in every launch: the path cut in TIME (recur.h: split_m).  Why: a direction-serial launch has
// one workgroup per sequence tile -- 194 / 250 for 64 x 3 s on 256 CUs, 85 % of the chip on average (profiles/r05_serial_vs_concurrent.txt) --
// while launches of both directions (388 / 500 workgroups, two pipelines) pack the chip.  Cutting every sequence at its middle keeps
// both: in phase 1 the forward direction walks rows [0, M) and the reverse direction rows [M, T), each leaving x' and its partial
// product for the rows it meets; in phase 2 they walk the other halves and finish z.  7 slab units per row, twice the tiles.
extern "C" int64_t dprnn_rnn_lin_state_floats(int nseq, int seq_tile) {
    const int nsr = seq_tile ? seq_tile % 1000 : 64;
    if (nsr != 32 && nsr != 48 && nsr != 64) return -1;
    const int64_t tiles = (nseq + nsr - 1) / nsr;
    return 2 * tiles * (int64_t)(512 * (nsr / 16) * 4 + 2 * nsr * 256 / 4);
}

extern "C" int dprnn_rnn_lin_split_fwd(int phase, const float* X, const float* Z, const double* stats, const float* gamma,
                                       const float* beta, float eps, int seqs_per_utt, int64_t rows_per_utt, float* Xout,
                                       const float* packed, const float* bsum, const void* lin_packed, const float* lin_bias,
                                       float* P, double* stats_out, float* state,
                                       int nseq, int T, int inner_n, int64_t outer_stride, int64_t inner_stride, int64_t t_stride,
                                       int seq_tile, void* stream) {
    if (!X || !packed || !bsum || !lin_packed || !P || !state) { dprnn_set_error("dprnn_rnn_lin_split_fwd: null pointer"); return -1; }
    if (phase != 1 && phase != 2) { dprnn_set_error("dprnn_rnn_lin_split_fwd: phase %d (1: first halves, 2: second halves)", phase); return -1; }
    if (phase == 2 && (!stats_out || !lin_bias)) { dprnn_set_error("dprnn_rnn_lin_split_fwd: phase 2 needs stats_out and the Linear's bias"); return -1; }
    if (nseq <= 0 || T <= 0 || inner_n <= 0 || seqs_per_utt <= 0 || rows_per_utt <= 0) {
        dprnn_set_error("dprnn_rnn_lin_split_fwd: bad sizes nseq=%d T=%d inner_n=%d seqs_per_utt=%d rows_per_utt=%lld", nseq, T, inner_n, seqs_per_utt, (long long)rows_per_utt);
        return -1;
    }
    if (Z && (phase != 1 || !stats || !gamma || !beta || !Xout || Xout == X || Z == P)) {
        dprnn_set_error("dprnn_rnn_lin_split_fwd: the pending norm belongs to phase 1 and needs stats, gamma, beta, an Xout different from X and a P different from Z");
        return -1;
    }
    const int nsr = seq_tile % 1000;
    if (seq_tile != 0 && (seq_tile < 2000 || seq_tile >= 3000 || (nsr != 32 && nsr != 48 && nsr != 64))) {
        dprnn_set_error("dprnn_rnn_lin_split_fwd: seq_tile 0, 2032, 2048 or 2064 (got %d)", seq_tile);
        return -1;
    }
    RecurArgs a{};
    a.X = X; a.Wp = packed; a.bsum = bsum; a.H = nullptr; a.nseq = nseq; a.T = T; a.inner_n = inner_n; a.ndir = 2;
    a.outer_stride = outer_stride; a.inner_stride = inner_stride; a.t_stride = t_stride; a.cell = 0;
    a.Z = Z; a.stats = stats; a.gamma = gamma; a.beta = beta; a.eps = eps; a.seqs_per_utt = seqs_per_utt;
    a.rows_per_utt = rows_per_utt; a.Xout = Xout;
    a.proj = phase; a.dir0 = 0; a.proj_dirs = 2; a.split_m = 0; a.state = state;
    a.Lp = lin_packed; a.lbias = lin_bias; a.P = P; a.Zout = P; a.stats_out = stats_out;
    if (!tile_offsets_fit(inner_n, outer_stride, inner_stride, t_stride, F)) {
        dprnn_set_error("dprnn_rnn_lin_split_fwd: a sequence tile spans more than 2 GiB (outer_stride=%lld rows)", (long long)outer_stride);
        return -1;
    }
    if (int rc = dprnn_pack_expect(packed, DPRNN_PACK_FWD, 0, 3, 2, "dprnn_rnn_lin_split_fwd")) return rc;
    a.sat = dprnn_sat_ptr();
    { const ScaleRef sr = dprnn_scale_ref(); a.xs_src = sr.src; a.xmax_dst = sr.dst; }
    hipStream_t st = (hipStream_t)stream;
    switch (seq_tile == 0 ? 64 : nsr) {
        case 32: return launch_lin<32>(a, true, st);
        case 48: return launch_lin<48>(a, true, st);
        default: return launch_lin<64>(a, true, st);
    }
}

// ---- ragged batches (utterances of different lengths in one launch; addressing as dprnn_rnn_fwd_ragged): intra-chunk (wg_desc == NULL):
// n_items chunks of T = K steps, seq_utt names their utterances; inter-chunk: n_items workgroup descriptors {row0 lo, row0 hi, T = S_u,
// count <= 64, utterance, 0, 0, 0}, every workgroup cut at the middle of ITS OWN T.  64-sequence tiles.  state:
// dprnn_rnn_lin_state_floats(64 * tiles, 2064) floats, tiles = ceil(n_items / 64) (intra-chunk) or n_items (inter-chunk).
extern "C" int dprnn_rnn_lin_split_ragged_fwd(int phase, const float* X, const float* Z, const double* stats, const float* gamma,
                                              const float* beta, float eps, const int64_t* utt_rows, const int* seq_utt, const int* wg_desc,
                                              int n_items, float* Xout, const float* packed, const float* bsum, const void* lin_packed,
                                              const float* lin_bias, float* P, double* stats_out, float* state, int T, int K, void* stream) {
    if (!X || !packed || !bsum || !lin_packed || !P || !state || !utt_rows || (!wg_desc && !seq_utt)) {
        dprnn_set_error("dprnn_rnn_lin_split_ragged_fwd: null pointer");
        return -1;
    }
    if (phase != 1 && phase != 2) { dprnn_set_error("dprnn_rnn_lin_split_ragged_fwd: phase %d (1: first halves, 2: second halves)", phase); return -1; }
    if (phase == 2 && (!stats_out || !lin_bias)) { dprnn_set_error("dprnn_rnn_lin_split_ragged_fwd: phase 2 needs stats_out and the Linear's bias"); return -1; }
    if (n_items <= 0 || K <= 0 || (!wg_desc && T <= 0)) { dprnn_set_error("dprnn_rnn_lin_split_ragged_fwd: bad sizes n_items=%d T=%d K=%d", n_items, T, K); return -1; }
    if (Z && (phase != 1 || !stats || !gamma || !beta || !Xout || Xout == X || Z == P)) {
        dprnn_set_error("dprnn_rnn_lin_split_ragged_fwd: the pending norm belongs to phase 1 and needs stats, gamma, beta, an Xout different from X and a P different from Z");
        return -1;
    }
    RecurArgs a{};
    a.X = X; a.Wp = packed; a.bsum = bsum; a.H = nullptr; a.nseq = n_items; a.T = T; a.ndir = 2; a.cell = 0;
    a.Z = Z; a.stats = stats; a.gamma = gamma; a.beta = beta; a.eps = eps; a.Xout = Xout;
    a.seqs_per_utt = 1; a.rows_per_utt = 1; a.utt_rows = reinterpret_cast<const long long*>(utt_rows);
    a.seq_utt = seq_utt; a.wg_desc = wg_desc;
    if (wg_desc) { a.inner_n = 1 << 30; a.outer_stride = 0; a.inner_stride = 1; a.t_stride = K; }
    else { a.inner_n = n_items; a.outer_stride = 0; a.inner_stride = K; a.t_stride = 1; }
    if ((long long)64 * K * F * 4 >= 0x7fffffffLL) { dprnn_set_error("dprnn_rnn_lin_split_ragged_fwd: chunk_length too large"); return -1; }
    a.proj = phase; a.dir0 = 0; a.proj_dirs = 2; a.split_m = 0; a.state = state;
    a.Lp = lin_packed; a.lbias = lin_bias; a.P = P; a.Zout = P; a.stats_out = stats_out;
    if (int rc = dprnn_pack_expect(packed, DPRNN_PACK_FWD, 0, 3, 2, "dprnn_rnn_lin_split_ragged_fwd")) return rc;
    a.sat = dprnn_sat_ptr();
    { const ScaleRef sr = dprnn_scale_ref(); a.xs_src = sr.src; a.xmax_dst = sr.dst; }
    return launch_lin<64>(a, true, (hipStream_t)stream);
}
