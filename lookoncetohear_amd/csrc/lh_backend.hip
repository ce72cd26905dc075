// Back end of the separator: causal ConvTranspose2d(64->4, 3x3) + re/im re-pack + iSTFT synthesis with
// overlap-add and carried tails, persistent workgroups of 8 waves.
//
// The transposed conv is factored so that every input frame is read from HBM once and multiplied once:
//     P[fr][f'][(kt,kf,o)] = sum_c Y[fr][f'][c] * Wd[c][o][kt][kf]          ([97 x 64] x [64 x 36] per frame)
//     D[t][o][f]           = b[o] + sum_{kt,kf} P[t-kt][f+1-kf][(kt,kf,o)]   (9-term gather-add from a 3-frame ring)
// A tile = 15 output frames of one utterance: 18 input frames (3 halo) stream through a register-staged LDS image,
// the 16 spectra Sx[t0..t0+15] (Sx[0] = carried istft_buf) are assembled in LDS as a second A image, and the
// synthesis filterbank ([194 x 192], resident in VGPRs as B fragments for the whole kernel) turns them into 16 frames
// per source that are overlap-added into 15 x 128 output samples.
// Both contractions run on split-precision fp16 MFMA (v = hi + lo, three v_mfma_f32_16x16x32_f16 per product,
// lh_split.h): the exact-fp32 MFMA version of this kernel spent 5x the matrix cycles and, at one wave per SIMD
// (386 registers), could not hide them: 0.46 ms per call at B = 32 against an HBM floor of 0.08 ms.
#include "lh_split.h"

namespace lh {

constexpr int BE_NT = 512;                // threads per workgroup (8 waves, two per SIMD)
constexpr int BE_TT = 15;                 // output frames per tile
constexpr int BE_NJ = BE_TT + 1;          // Sx frames t0 .. t0+15 (one MFMA row tile per source)
constexpr int BE_NP = 36;                 // partial-product columns (kt,kf,o); the B image pads them to 48
constexpr int BE_PP = 40;                 // P row stride: 16-byte aligned rows, the gather reads the 4 outputs of a tap as one float4
constexpr int BE_SK = 7;                  // synthesis k-steps: 194 spectrum rows -> 224
constexpr int BE_SA = BE_SK * 4 * BE_NJ * 8;   // halves per source in the Sx A image
constexpr int BE_FP = NFFT + 4;           // synthesis frame staging row
constexpr int BE_NLD = (NF * 16 + BE_NT - 1) / BE_NT;   // float4 per thread and frame (4)
#if !defined(BE_RING_N)
#define BE_RING_N 2
#endif
constexpr int BE_RING = BE_RING_N;        // input frames in flight per workgroup (2: no spills and 0.21 ms per call at B = 32; 4 = rounds 2-3: 22 spilled VGPRs with the range-safe staging, 0.23 ms — profiles/r04c)
#if defined(LH_PROBE_TRACE)              // timing probe build only (scripts/probe_trace.py --backend): stamps of workgroup 3, tile 2 of its run
__device__ unsigned long long lh_be_trace_buf[32];
#if defined(LH_PROBE_TRACE_T1)           // streaming shape (B = 1, T = 1): the only workgroup, its only tile; 20 / 21 = kernel entry / prologue done
#define BE_STAMP(k) do { if (blockIdx.x == 0 && tid == 0) lh_be_trace_buf[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BE_STAMP(k) do { if (blockIdx.x == 3 && tid == 0 && tk == k0_ + 2) lh_be_trace_buf[k] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define BE_STAMP(k) do { } while (0)
#endif

// grid = persistent (<= 256), block 512
__global__ void __launch_bounds__(BE_NT) k_deconv_istft(const float* __restrict__ y, const float* __restrict__ dbuf_in,
                                                        float* __restrict__ dbuf_out, const float* __restrict__ ibuf_in,
                                                        float* __restrict__ ibuf_out, const _Float16* __restrict__ wd_pk,
                                                        const float* __restrict__ bd, const _Float16* __restrict__ wfb_pk,
                                                        float* __restrict__ wave_out, unsigned* __restrict__ range_flag,
                                                        int keep_nonfinite, int B, int T, int runs_per_b) {
    // Two input frames per loop iteration (half the barriers; at one frame the loop was ~80 % stall): two A images, and
    // a partial-product ring of 4 frames (2 being written while the gather still reads the 2 before them).
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * FR_A];                 // A images of two input frames
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * FR_A];
    // partial products of 4 frames; bin f sits in row f + 1, rows 0 and 98 stay zero (the f -/+ 1 taps of the edge bins)
    __shared__ __attribute__((aligned(16))) float pring[4][NF + 2][BE_PP];
    __shared__ __attribute__((aligned(16))) _Float16 sxh[NSRC * BE_SA];             // A image of the 16 spectra
    __shared__ __attribute__((aligned(16))) _Float16 sxl[NSRC * BE_SA];
    // synthesis frames: only live after the frame loop, in the space of the (then dead) hi A images
    static_assert(sizeof(float) * BE_NJ * NSRC * BE_FP <= sizeof(_Float16) * 2 * FR_A, "frs must fit in ahi");
    float (*frs)[NSRC][BE_FP] = reinterpret_cast<float (*)[NSRC][BE_FP]>(ahi);
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;

    // B fragments: the deconv taps [64 -> 48 columns] (3 column tiles x 2 k-steps, 6 KB) sit in LDS in fragment order;
    // the synthesis filterbank tiles 2w, 2w+1 (of 12; 112 registers) are fetched from L2 once per tile, right before
    // they are used — the registers hold the frame prefetch ring while the frames stream
    __shared__ __attribute__((aligned(16))) _Float16 wds[3 * 2 * 64 * 16];
    // Range-safe splits (pow2_scale, lh_common.h).  The frames are rows of the un-normalised residual stream: the rows a wave
    // stages share one power of two per frame (stage_frame), `rinv[q][row]` undoes it on the partial products.  The spectra are split a second
    // time (synthesis A image): frame td is scaled by a power of two taken from a BOUND of its magnitude,
    //     |D[td]| <= max|b| + 3 W1 (M[td] + M[td-1] + M[td-2]),   M[fr] = max |Y[fr]|,  W1 = max_col sum_c |Wd[col][c]|
    // (data-independent of the products, so no extra barrier); `sinv[jd]` undoes it on the synthesis accumulators.
    __shared__ __attribute__((aligned(16))) float rinv[2][FR_RP];
    __shared__ float wmax[2][BE_NT / 64];             // per-wave maxima of the two frames being staged
    __shared__ float fmaxr[4];                        // M[fr] ring, slot (fr + 4) & 3 like the partial products
    __shared__ __attribute__((aligned(16))) float sinv[BE_NJ];
    __shared__ float w1s[48];
#if defined(LH_PROBE_TRACE_T1)
    { const int tk = 0, k0_ = 0; (void)tk; (void)k0_; BE_STAMP(20); }
#endif
    for (int i = tid; i < 3 * 2 * 64 * 2; i += BE_NT)
        *reinterpret_cast<f16x8*>(&wds[i * 8]) = *reinterpret_cast<const f16x8*>(&wd_pk[i * 8]);
    const bool synth = wave < 6;
    float bias4[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) bias4[o] = bd[o];
    const float bmax = fmaxf(fmaxf(fabsf(bias4[0]), fabsf(bias4[1])), fmaxf(fabsf(bias4[2]), fabsf(bias4[3])));
    if (tid < 2 * FR_RP) rinv[0][tid] = 0.f;
    if (tid < 4) fmaxr[tid] = 0.f;
    if (tid < BE_NJ) sinv[tid] = 0.f;
    __syncthreads();                              // wds complete
    if (tid < 48) {                               // L1 norm of tap column tid (|hi| + |lo| >= |w|), fragment order
        float a = 0.f;
        for (int k = 0; k < C; ++k) {
            const int idx = (((tid >> 4) * 2 + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (tid & 15)) * 16 + (k & 7);
            a += fabsf((float)wds[idx]) + fabsf((float)wds[idx + 8]);
        }
        w1s[tid] = a;
    }
    __syncthreads();
    float W1 = 0.f;
    for (int i = 0; i < 48; ++i) W1 = fmaxf(W1, w1s[i]);
    W1 *= 3.0f;                                   // three frequency taps per frame

    // rows / k-padding of the A images that are never written only feed dropped outputs or multiply zero weights,
    // but must be finite
    static_assert((2 * FR_A) % 8 == 0 && (NSRC * BE_SA) % 8 == 0, "16-byte zero fill");
    const f16x8 z8 = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < 2 * FR_A / 8; i += BE_NT) {
        *reinterpret_cast<f16x8*>(&ahi[i * 8]) = z8;
        *reinterpret_cast<f16x8*>(&alo[i * 8]) = z8;
    }
    for (int i = tid; i < NSRC * BE_SA / 8; i += BE_NT) {
        *reinterpret_cast<f16x8*>(&sxh[i * 8]) = z8;
        *reinterpret_cast<f16x8*>(&sxl[i * 8]) = z8;
    }
    auto zero_guards = [&]() {                // rows 0 and 98 of the four partial-product slots
        for (int i = tid; i < 4 * 2 * BE_PP; i += BE_NT) pring[i / (2 * BE_PP)][((i / BE_PP) & 1) * (NF + 1)][i % BE_PP] = 0.f;
    };
    zero_guards();
    // The carried conv halo is stored channel-major ([B][64][2][97], the reference's state layout), the frames bin-major
    // ([97][64]): both directions of that transpose go through LDS so that both global sides stay coalesced (the 4-byte
    // strided loads of the two halo frames cost 11.6 k cycles of a 61 k-cycle streaming call, the element-wise state copy
    // ~18 k: profiles/r03m_backend_t1.txt).  The buffer lives in the partial-product ring while that is empty / dead.
    static_assert(2 * C * NF <= 4 * (NF + 2) * BE_PP, "transpose buffer must fit in pring");
    float* const tbuf = &pring[0][0][0];

    // A workgroup walks a RUN of consecutive tiles of one utterance: only the run's first tile pays the 3-frame halo
    // (re-multiplying frames t0-3 .. t0-1); later tiles find the partial products of frames t0-2, t0-1 still in the ring
    // and take Sx[0] (= the spectrum of frame t0-1) from the previous tile's row 15.  At 42 tiles per 5 s clip the halo
    // was 20 % extra frames (PMC: 1.42x the algorithmic bytes); 8 runs per utterance at B = 32 make it 4 %.
#if defined(LH_PROBE_TRACE_T1)
    { const int tk = 0, k0_ = 0; (void)tk; (void)k0_; BE_STAMP(21); }
#endif
    const int tiles_per_b = (T + BE_TT - 1) / BE_TT;
    const int n_runs = B * runs_per_b;
    const long L = (long)HOP * T;
    // flat loop over the tiles of this workgroup's runs: runs blockIdx.x, blockIdx.x + gridDim.x, ...; inside a run the
    // tiles follow each other
    int run = blockIdx.x, tk = 0, k1 = 0, k0_ = 0;    // tk == k1: fetch the next run (k0_: its first tile, probe only)
    for (;;) {
        bool first = false;
        if (tk == k1) {
            if (run >= n_runs) break;
            const int rb0 = run % runs_per_b;
            tk = (int)((long)tiles_per_b * rb0 / runs_per_b);
            k1 = (int)((long)tiles_per_b * (rb0 + 1) / runs_per_b);
            first = true;
            k0_ = tk;
        }
        const int b = run / runs_per_b;
        {
        const int t0 = tk * BE_TT;
        const int nt_out = min(BE_TT, T - t0);
        const bool cont = !first;                     // ring and Sx[15] of the previous tile are this tile's history
        // `tk >> 30` is always 0 but ties the thread index to the loop variable: without it LICM hoists every per-thread
        // address of the frame loop out of the persistent loop and the kernel spills ~35 registers (measured 0.28 ms per
        // call with the hoisted addresses and spills, 0.26 ms with the indices recomputed per tile)
        const int tv = tid + (tk >> 30);
        __syncthreads();
        BE_STAMP(0);

        // the BE_NLD float4 of a frame this thread stages (element e: row e >> 4, channels 4 (e & 15) ..) -> A image, scaled by
        // one power of two per WAVE (the rows a wave stages: 16 lanes each, so every row has one scale; lh_split.h
        // frame_store_scaled); returns the wave's maximum (wave-uniform)
        auto stage_frame = [&](_Float16* ih, _Float16* il, float* riv, const float4 (&v)[BE_NLD]) -> float {
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < BE_NLD; ++i) m = fmaxf(m, absmax4(v[i]));
            m = wave_max_uniform(m);
            float sc, iv;
            pow2_scale<FR_TE>(m, sc, iv);
#pragma unroll
            for (int i = 0; i < BE_NLD; ++i) {
                const int e = tv + BE_NT * i;
                if (e < NF * 16) {
                    store_split4<FR_RP>(ih, il, e >> 4, (e & 15) * 4, make_float4(v[i].x * sc, v[i].y * sc, v[i].z * sc, v[i].w * sc));
                    if ((e & 15) == 0) riv[e >> 4] = iv;
                }
            }
            return m;
        };
        // one spectrum value -> row jd of source s's A image
        auto put_sx = [&](int jd, int s, int k, float v) {
            _Float16 h, l;
            split_hl(v, h, l);
            const int idx = s * BE_SA + a_index<BE_NJ>(jd, k);
            sxh[idx] = h;
            sxl[idx] = l;
        };
        // Sx frame 0: the carried spectrum of the previous call (first tile of the clip), the previous tile's last
        // frame (inside a run), or recomputed from the halo frames (first tile of a later run)
        if (t0 == 0) {                                // (workgroup-uniform)
            static_assert(NSRC * NK <= BE_NT, "one carried spectrum value per thread");
            const float cv = tid < NSRC * NK ? ibuf_in[(long)b * NSRC * NK + tid] : 0.f;
            const float wm = wave_max(fabsf(cv));
            if (lane == 0) wmax[0][wave] = wm;
            __syncthreads();
            float m = wmax[0][0];
#pragma unroll
            for (int w = 1; w < BE_NT / 64; ++w) m = fmaxf(m, wmax[0][w]);
            float sc, iv;
            pow2_scale<12>(m, sc, iv);
            if (tid < NSRC * NK) put_sx(0, tid / NK, tid % NK, cv * sc);
            if (tid == 0) sinv[0] = iv;
            __syncthreads();                          // wmax is rewritten by the frame staging
        } else if (cont) {
            for (int i = tid; i < NSRC * NK; i += BE_NT) {
                const int s = i / NK, k = i % NK;
                const int src = s * BE_SA + a_index<BE_NJ>(BE_TT, k), dst = s * BE_SA + a_index<BE_NJ>(0, k);
                sxh[dst] = sxh[src];
                sxl[dst] = sxl[src];
            }
            if (tid == 0) sinv[0] = sinv[BE_TT];
            __syncthreads();                          // row 15 is rewritten by this tile's last frame
        }
        BE_STAMP(17);

        // input frames t0-3 .. t0+nt_out-1 ; after frame fr has been multiplied, output frame td = fr is complete
        float4 stg[BE_RING][BE_NLD];
        // fr >= 0: a frame of y.  Per-thread element offsets once per tile, the frame's base is wave-uniform: the load issue
        // of a frame pair took 1300 cycles when every float4 recomputed its 64-bit index
        unsigned eoff[BE_NLD];
#pragma unroll
        for (int i = 0; i < BE_NLD; ++i) eoff[i] = (unsigned)min(tv + BE_NT * i, NF * 16 - 1) * 16u;
        const char* yb = reinterpret_cast<const char*>(y) + (long)b * T * NF * C * 4;
        auto load_frame_y = [&](int fr, float4 (&dst)[BE_NLD]) {
            const char* base = yb + (long)fr * (NF * C * 4);
#pragma unroll
            for (int i = 0; i < BE_NLD; ++i) dst[i] = *reinterpret_cast<const float4*>(base + eoff[i]);
        };
        const int fr_first = cont ? t0 : max(t0 - 3, -2);   // frames below -2 do not exist (their taps see nothing)
        const int fr_end = t0 + nt_out;
        BE_STAMP(18);
        // BE_RING frames in flight per workgroup: with a single one the loop ran at one HBM round trip per frame
        if (fr_first >= 0) {                          // no carried halo frame in the first ring turn: four straight loads
#pragma unroll
            for (int u = 0; u < BE_RING; ++u) load_frame_y(min(fr_first + u, T - 1), stg[u]);
        } else {                                      // (the carried halo frames -2, -1 are staged from the state below)
#pragma unroll
            for (int u = 0; u < BE_RING; ++u)
                if (fr_first + u >= 0 && fr_first + u < fr_end) load_frame_y(fr_first + u, stg[u]);
        }
        BE_STAMP(1);
        for (int fbase = fr_first; fbase < fr_end; fbase += BE_RING) {
            BE_STAMP(2 + (fbase - fr_first) / BE_RING);
#pragma unroll
          for (int u = 0; u < BE_RING; u += 2) {
            const int fr = fbase + u;                 // this iteration: frames fr and fr + 1 (the second may not exist)
            if (fr >= fr_end) break;
            const bool two = fr + 1 < fr_end;
            const bool tr_it = fbase == fr_first + BE_RING && u == 0;      // probe only
            if (tr_it) BE_STAMP(11);
            // stage the frames into the two A images, refill their ring slots
            if (fr < 0) {
                // frames -2, -1 = rows 0, 1 of the carried halo [c][2][f] (first pair of a clip's first tile): a flat coalesced
                // copy into the transpose buffer, then rows of four channels like any other frame
                for (int i = tid; i < 2 * C * NF; i += BE_NT) tbuf[i] = dbuf_in[(long)b * 2 * C * NF + i];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float4 hv[BE_NLD];
#pragma unroll
                    for (int i = 0; i < BE_NLD; ++i) {
                        const int e = min(tv + BE_NT * i, NF * 16 - 1);
                        const int f = e >> 4, c0 = (e & 15) * 4;
                        hv[i] = make_float4(tbuf[((c0 + 0) * 2 + q) * NF + f], tbuf[((c0 + 1) * 2 + q) * NF + f],
                                            tbuf[((c0 + 2) * 2 + q) * NF + f], tbuf[((c0 + 3) * 2 + q) * NF + f]);
                    }
                    const float wm = stage_frame(ahi + q * FR_A, alo + q * FR_A, rinv[q], hv);
                    if (lane == 0) wmax[q][wave] = wm;
                }
                __syncthreads();
                zero_guards();                        // the buffer ran over guard rows; the products below write rows 1..97 only
            } else {
                const float wm0 = stage_frame(ahi, alo, rinv[0], stg[u]);
                const float wm1 = two ? stage_frame(ahi + FR_A, alo + FR_A, rinv[1], stg[u + 1]) : 0.f;
                if (lane == 0) { wmax[0][wave] = wm0; wmax[1][wave] = wm1; }
            }
            __syncthreads();
            if (wave == 7) {                           // the one wave without a product tile: M[fr], M[fr + 1] for the spectrum
#pragma unroll                                         // bounds of frames fr .. fr + 3, off the other waves' path
                for (int q = 0; q < 2; ++q) {
                    const float m = wave_max(lane < BE_NT / 64 ? wmax[q][lane] : 0.f);
                    if (lane == 0 && (q == 0 || two)) fmaxr[(fr + q + 4) & 3] = m;
                }
            }
            if (tr_it) BE_STAMP(12);
            // Refill the two ring slots UNCONDITIONALLY (frame index clamped to the tile: past its end the tile's last frame
            // is loaded again — a cache hit — and never used): with the loads under `if (fr + BE_RING < fr_end)` the compiler cannot count the outstanding
            // loads at the loop head and waits for vmcnt(0) there, i.e. for the loads it has just issued — every pair of
            // frames then paid a full HBM round trip (s_memtime trace: 8.5 k cycles per pair against ~3 k of work).
            // (frames fr + BE_RING >= fr_first + 4 >= 2: never the carried halo frames)
            // The second slot is refilled behind the products: all eight waves issuing both frames' loads at the same point
            // queued 2000 cycles on the CU's vector-memory path.
            load_frame_y(min(fr + BE_RING, fr_end - 1), stg[u]);
            if (tr_it) BE_STAMP(15);

            // P[fr + q] = Y[fr + q] (97 x 64) * Wd (64 x 48).  Wave w < 7 owns row tile w of both frames: its A fragments
            // are read once and feed the three column tiles, six independent accumulator chains per wave.  (Spreading
            // the 42 (row tile, column tile) products one by one over the 8 waves re-read A and W per product and ran
            // them strictly one after the other: 770 cycles per product in the s_memtime trace, 55 % of the frame loop.)
            if (wave < 7) {
                const int mt = wave;
                f16x8 ah[2][2], al[2][2];                 // [frame][k-step]
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int idx = q * FR_A + a_slot<FR_RP>(ks * 4 + g4, mt * 16 + l15);
                        ah[q][ks] = *reinterpret_cast<const f16x8*>(&ahi[idx]);
                        al[q][ks] = *reinterpret_cast<const f16x8*>(&alo[idx]);
                    }
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {          // each W fragment is read once and serves both frames
                    f32x4 am[2], ac[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) { am[q] = f32x4{0.f, 0.f, 0.f, 0.f}; ac[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const f16x8 wh = *reinterpret_cast<const f16x8*>(&wds[((nt * 2 + ks) * 64 + lane) * 16]);
                        const f16x8 wl = *reinterpret_cast<const f16x8*>(&wds[((nt * 2 + ks) * 64 + lane) * 16 + 8]);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {      // (the second frame's image is stale, never stored, when !two)
                            am[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q][ks], wh, am[q], 0, 0, 0);
                            ac[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q][ks], wl, ac[q], 0, 0, 0);
                            ac[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q][ks], wh, ac[q], 0, 0, 0);
                        }
                    }
                    const int col = nt * 16 + l15;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q == 1 && !two) break;
                        const int slot = (fr + q + 4) & 3;    // fr >= -2
                        const float4 iv4 = *reinterpret_cast<const float4*>(&rinv[q][mt * 16 + g4 * 4]);
                        const float iv[4] = {iv4.x, iv4.y, iv4.z, iv4.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = mt * 16 + g4 * 4 + r;
                            if (f < NF && col < BE_NP) pring[slot][f + 1][col] = (am[q][r] + ac[q][r]) * iv[r];
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_frame_y(min(fr + 1 + BE_RING, fr_end - 1), stg[u + 1]);
            if (tr_it) BE_STAMP(16);
            __syncthreads();
            if (tr_it) BE_STAMP(13);

            // output frames td = fr, fr + 1: D[o][f] = b[o] + sum_{kt,kf} P[td-kt][f+1-kf][(kt*3+kf)*4 + o]
            for (int i = tv; i < (two ? 2 : 1) * NF; i += BE_NT) {      // one (frame, bin) per thread: its 4 outputs together
                const int q = i >= NF, f = i - NF * q;
                const int td = fr + q;
                if (td < t0 - 1 || td < 0) continue;
                const int jd = td + 1 - t0;           // Sx frame index inside the tile
                float4 v = make_float4(bias4[0], bias4[1], bias4[2], bias4[3]);
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const int ps = (td - kt + 4) & 3;         // td >= 0, so frame td - kt >= -2 exists (halo or zero state)
#pragma unroll
                    for (int kf = 0; kf < 3; ++kf) {          // guard rows: no bounds
                        const float4 pv = *reinterpret_cast<const float4*>(&pring[ps][f + 2 - kf][(kt * 3 + kf) * 4]);
                        v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
                    }
                }
                const float vo[4] = {v.x, v.y, v.z, v.w};
                // frames td - 1, td - 2 below -2 do not exist; their ring slots then hold maxima of frames that do (>= 0: a
                // larger bound, still a bound)
                float ssx, isx;
                pow2_scale<14>(fmaf(W1, fmaxr[(td + 4) & 3] + fmaxr[(td + 3) & 3] + fmaxr[(td + 2) & 3], bmax), ssx, isx);
                if (f == 0) sinv[jd] = isx;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int s = o >> 1, k = (o & 1) * NF + f;
                    put_sx(jd, s, k, vo[o] * ssx);
                    if (td == T - 1) ibuf_out[((long)b * NSRC + s) * NK + k] = vo[o];      // new carried spectrum (exact fp32)
                }
            }
            if (tr_it) BE_STAMP(14);
            // the next iteration's staging barrier orders these pring reads before their slots are overwritten
          }
        }
        __syncthreads();
        BE_STAMP(8);

        // new carried conv halo (last tile only): frames T-2, T-1 as [c][2][f]; a frame of y is transposed through LDS
        if (t0 + nt_out == T) {
            for (int r = 0; r < 2; ++r) {
                const int fr = T - 2 + r;             // workgroup-uniform
                if (fr >= 0) {
                    const float* src = y + ((long)b * T + fr) * NF * C;
                    for (int i = tid; i < NF * C; i += BE_NT) tbuf[(i & (C - 1)) * NF + (i >> 6)] = src[i];
                    __syncthreads();
                    for (int i = tid; i < NF * C; i += BE_NT) {
                        const int c = i / NF, f = i - c * NF;
                        dbuf_out[(((long)b * C + c) * 2 + r) * NF + f] = tbuf[i];
                    }
                    __syncthreads();
                } else {                              // still a carried frame: same layout on both sides
                    for (int i = tid; i < NF * C; i += BE_NT) {
                        const int c = i / NF, f = i - c * NF;
                        dbuf_out[(((long)b * C + c) * 2 + r) * NF + f] = dbuf_in[(((long)b * C + c) * 2 + (fr + 2)) * NF + f];
                    }
                }
            }
            zero_guards();                            // (ordered before the next tile's gather by the barriers in between)
        }

        // synthesis: fr[jd][s][n] = sum_k Sx[jd][s][k] Wdec[k][n]; row tile = the 16 frames of a source, waves 0..5
        // own 2 of the 12 column tiles of 16 samples each
        if (synth) {
#pragma unroll 1
            for (int i = 0; i < 2; ++i) {             // one column tile at a time: 56 registers of B fragments
                f16x8 wfh[BE_SK], wfl[BE_SK];
                load_w<BE_SK>(wfb_pk, 2 * wave + i, lane, wfh, wfl);
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    const f32x4 acc = mma_tile<BE_NJ, BE_SK>(sxh + s * BE_SA, sxl + s * BE_SA, 0, g4, l15, wfh, wfl, 0.f);
                    const float4 iv4 = *reinterpret_cast<const float4*>(&sinv[g4 * 4]);      // 1 / scale of spectra g4*4 ..
                    const float iv[4] = {iv4.x, iv4.y, iv4.z, iv4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) frs[g4 * 4 + r][s][(2 * wave + i) * 16 + l15] = acc[r] * iv[r];
                }
            }
        }
        __syncthreads();
        BE_STAMP(9);

        // overlap-add: output frame t (samples 128t..128t+127) = fr[t+1][0:128] + fr[t][128:192]
        bool bad = false;
        for (int i = tid; i < nt_out * NSRC * HOP; i += BE_NT) {
            const int n = i % HOP, s = (i / HOP) % NSRC, jt = i / (HOP * NSRC);
            float v = frs[jt + 1][s][n];
            if (n < NFFT - HOP) v += frs[jt][s][n + HOP];
            // inf / NaN (the input or the state held inf / NaN, or a true fp32 overflow): the CALLER's flag word is raised
            // (range contract, lookonce_hip.h) and the sample is stored as the caller asked — as it is (the reference's
            // behaviour: NaN out, nothing hidden; the offline host) or as 0 (silence, not NaN, reaches a listener; streaming)
            const bool nf = (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u;
            wave_out[((long)b * NSRC + s) * L + (long)(t0 + jt) * HOP + n] = (nf && !keep_nonfinite) ? 0.f : v;
            bad |= nf;
        }
        // sticky until lh_range_status / lh_range_flag_copy fetch it.  A plain store (every writer writes 1), system scope: the
        // word may live in device memory or in pinned host memory the caller reads directly (Streamer: no polling launches)
        if (bad && range_flag) __hip_atomic_store(range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        // frs lived in the hi A images: their pad rows (97..111 feed dropped outputs) must hold finite numbers again —
        // only those: the 97 real rows are rewritten by the staging of the next frames
        for (int i = tid; i < 2 * (FR_RP - NF) * 16; i += BE_NT) {
            const int img = i / ((FR_RP - NF) * 16), e = i % ((FR_RP - NF) * 16);
            *reinterpret_cast<f16x4*>(&ahi[img * FR_A + a_index<FR_RP>(NF + (e >> 4), (e & 15) * 4)]) = f16x4{0, 0, 0, 0};
        }
        BE_STAMP(10);
        }
        if (++tk == k1) run += gridDim.x;
    }
}

static int g_runs_per_b = 0;            // lh_set_tuning key 6: runs per utterance (0 = automatic); tests force 1 on tiny batches
int backend_set_runs(int v) {
    if (v < 0) return LH_ERR_ARG;
    g_runs_per_b = v;
    return LH_OK;
}

}  // namespace lh

#if defined(LH_PROBE_TRACE)
extern "C" int lh_probe_be_trace_read(unsigned long long* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lh::lh_be_trace_buf), sizeof(lh::lh_be_trace_buf)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int lh_deconv_istft(const float* y, const float* deconv_buf_in, float* deconv_buf_out,
                               const float* istft_buf_in, float* istft_buf_out, const void* wdec_pk,
                               const float* bdec, const void* wfb_dec, float* wave_out, unsigned* range_flag,
                               int keep_nonfinite, int B, int T, lh_stream_t stream) {
    using namespace lh;
    if (!y || !deconv_buf_in || !deconv_buf_out || !istft_buf_in || !istft_buf_out || !wdec_pk || !bdec || !wfb_dec ||
        !wave_out || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    if (deconv_buf_in == deconv_buf_out || istft_buf_in == istft_buf_out) return LH_ERR_ARG;
    // runs of consecutive tiles: as many per utterance as keep every CU busy (one workgroup per CU), at most one per tile
    const int tiles_per_b = (T + BE_TT - 1) / BE_TT;
    int runs_per_b = g_runs_per_b ? g_runs_per_b : 256 / B;
    runs_per_b = runs_per_b < 1 ? 1 : (runs_per_b > tiles_per_b ? tiles_per_b : runs_per_b);
    const int n_runs = B * runs_per_b;
    hipLaunchKernelGGL(k_deconv_istft, dim3(n_runs < 256 ? n_runs : 256), dim3(BE_NT), 0, (hipStream_t)stream, y,
                       deconv_buf_in, deconv_buf_out, istft_buf_in, istft_buf_out, (const _Float16*)wdec_pk, bdec,
                       (const _Float16*)wfb_dec, wave_out, range_flag, keep_nonfinite, B, T, runs_per_b);
    return check_launch();
}

// ---- range contract of the split-precision arithmetic (include/lookonce_hip.h).  The flag is CALLER-OWNED device memory:
// two 32-bit words, [0] = the sticky word lh_deconv_istft raises, [1] = the value the last fetch took out of it.  One
// atomicExch moves [0] to [1] and clears it, so a flag raised between "read" and "clear" cannot be lost, and a caller
// can only ever consume its own forwards' flag (round 3 kept ONE word per device that every Net / Streamer shared).
namespace lh {
__global__ void k_range_fetch(unsigned* flag) { flag[1] = atomicExch(&flag[0], 0u); }
}  // namespace lh
static int range_fetch(unsigned* flag, lh_stream_t stream) {
    if (!flag) return LH_ERR_ARG;
    hipLaunchKernelGGL(lh::k_range_fetch, dim3(1), dim3(1), 0, (hipStream_t)stream, flag);
    return lh::check_launch();
}
extern "C" int lh_range_flag_copy(unsigned* flag, void* host_pinned, lh_stream_t stream) {
    if (!host_pinned) return LH_ERR_ARG;
    const int rc = range_fetch(flag, stream);
    if (rc != LH_OK) return rc;
    return hipMemcpyAsync(host_pinned, flag + 1, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess
               ? LH_OK : LH_ERR_LAUNCH;
}
extern "C" int lh_range_flag_clear(unsigned* flag, lh_stream_t stream) {
    if (!flag) return LH_ERR_ARG;
    return hipMemsetAsync(flag, 0, 2 * sizeof(unsigned), (hipStream_t)stream) == hipSuccess ? LH_OK : LH_ERR_LAUNCH;
}
extern "C" int lh_range_status(unsigned* flag, lh_stream_t stream) {
    unsigned v = 0;
    const int rc = range_fetch(flag, stream);
    if (rc != LH_OK) return rc;
    if (hipMemcpyAsync(&v, flag + 1, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return LH_ERR_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return LH_ERR_LAUNCH;
    return v ? LH_ERR_RANGE : LH_OK;
}

namespace lh {
__global__ void k_selftest_subnormal(float* out) {
    f16x8 a, b, b2;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)9.5367431640625e-07f;      // 2^-20: subnormal in fp16
        b[i] = (_Float16)1.0f;
        b2[i] = (_Float16)6.103515625e-05f;         // 2^-14: smallest normal
    }
    f32x4 c = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);        // 32 * 2^-20
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b2, c2, 0, 0, 0);     // 32 * 2^-34
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = c2[0]; }
}
}  // namespace lh
extern "C" int lh_selftest_fp16_subnormal(lh_stream_t stream) {
    float* d = nullptr;
    if (hipMalloc(&d, 2 * sizeof(float)) != hipSuccess) return LH_ERR_LAUNCH;
    hipLaunchKernelGGL(lh::k_selftest_subnormal, dim3(1), dim3(64), 0, (hipStream_t)stream, d);
    float h[2] = {0.f, 0.f};
    const bool ok = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess &&
                    hipStreamSynchronize((hipStream_t)stream) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return LH_ERR_LAUNCH;
    return (h[0] == 32.0f * 9.5367431640625e-07f && h[1] == 32.0f * 9.5367431640625e-07f * 6.103515625e-05f)
               ? LH_OK : LH_ERR_UNSUPPORTED;
}
