/* lookonce_hip.h — C ABI of the MI355X-native LookOnceToHear separator forward path.
 *
 * Drop-in boundary: the reference selects its model by a dotted string
 * (`utils.import_attr(model)(**model_params)`, reference src/ts_hear_embed_pl_module.py:25 with the string
 * from configs/tsh.json:4).  `lookoncetohear_amd.net.Net` is that class; it owns parameters and streaming
 * state as torch tensors and calls the stateless entry points below with raw device pointers on the current
 * HIP stream.  Conventions (same as the only C-ABI precedent in the reference,
 * src/datasets/motion_simulator.py:41-46): every function returns int, 0 = OK, and the caller asserts.
 *   - all pointers are DEVICE pointers to fp32 unless stated; nothing is allocated or freed here;
 *   - activations are channel-last  X[b][t][f][c]  (B,T,F=97,C=64);
 *   - `*_in` / `*_out` state pairs must not alias (several workgroups read the old state while one writes
 *     the new one);
 *   - shape constants are those of configs/tsh.json (nfft 192, hop 128, F 97, C 64, H 64, heads 4, E 6,
 *     Vd 16, window 50, 2 mics, 2 sources); `lh_check_config` returns LH_ERR_UNSUPPORTED for anything else.
 *
 * Each entry point cites the reference lines it replaces.
 */
#ifndef LOOKONCE_HIP_H
#define LOOKONCE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lh_stream_t; /* hipStream_t */

enum {
    LH_OK = 0,
    LH_ERR_ARG = 1,         /* null pointer / non-positive size */
    LH_ERR_UNSUPPORTED = 2, /* shape constants differ from the compiled configuration */
    LH_ERR_LAUNCH = 3,      /* hipGetLastError() != hipSuccess after the launch */
    LH_ERR_RANGE = 4        /* lh_range_status only: a non-finite output sample was produced since the caller's last check */
};

/* Range contract of the split-precision ("f16x3") arithmetic.  Operands are carried as fp16 hi + fp16 lo, so every value
 * a kernel SPLITS must satisfy |v| < 65504 AFTER the kernel's own scaling.  LayerNorm outputs, hidden states, attention
 * rows and weights always do.  The kernels that read un-normalised data (the waveform in lh_stft_conv_in, the residual
 * stream in lh_qkv_proj_ln and lh_deconv_istft) multiply each row / tile by an exact power of two chosen from its own
 * maximum before the split and undo it after the fp32 accumulation (since ABI 12), so any finite fp32 input whose exact
 * result is finite in fp32 is computed to the same ~22 bits relative to the row maximum — the reference's behaviour
 * (plain fp32, tfgridnet_causal.py:188-283).  What remains is a GUARD for non-finite data: lh_deconv_istft raises a
 * CALLER-OWNED flag for a non-finite output sample and stores it as the caller asks (`keep_nonfinite`, ABI 13: as it is —
 * the reference's behaviour, the default of the offline host — or as 0 for a listener):
 *   range_flag       two 32-bit words of DEVICE-ACCESSIBLE memory (device memory, or pinned host memory the caller reads
 *                    directly — the streaming host does that and needs no polling launch at all), zero-initialised by the
 *                    caller: [0] = sticky word lh_deconv_istft sets to 1 with a system-scope store (NULL = no reporting),
 *                    [1] = the value the last fetch took out of [0].  Each Net / Streamer / host thread owns its own, so
 *                    concurrent forwards on one device cannot consume or clear one another's flag (ABI <= 11 kept one
 *                    word per device).
 *   lh_range_status: [1] = atomicExch([0], 0) on `stream`, copies [1] to the host, WAITS for the stream; LH_OK, or
 *                    LH_ERR_RANGE when it was set.  The one entry point that synchronises: call it once per forward.
 *   lh_range_flag_copy: the same exchange followed by an asynchronous copy of [1] to `host_pinned` (4 bytes of pinned host
 *                    memory) on `stream`, no wait (the streaming host looks at the word when later chunks arrive).
 *   lh_range_flag_clear: asynchronous clear of both words on `stream`.
 *   lh_selftest_fp16_subnormal: runs one v_mfma_f32_16x16x32_f16 on fp16-subnormal operands (the un-rescaled lo halves
 *                    rely on the matrix core taking them at full value); LH_OK, or LH_ERR_UNSUPPORTED when they are
 *                    flushed.  Synchronises. */
int lh_range_status(unsigned* range_flag, lh_stream_t stream);
int lh_range_flag_copy(unsigned* range_flag, void* host_pinned, lh_stream_t stream);
int lh_range_flag_clear(unsigned* range_flag, lh_stream_t stream);
int lh_selftest_fp16_subnormal(lh_stream_t stream);

/* Contraction arithmetic of the recurrent kernels (argument `mode`):
 *   LH_GEMM_F32   exact fp32 MFMA (v_mfma_f32_16x16x4_f32); w_pk = fp32 image  [dirs][4][4][32][64]
 *   LH_GEMM_F16X3 split precision: each fp32 operand = fp16 hi + fp16 lo, three fp16 MFMAs
 *                 (hi*hi, hi*lo, lo*hi) accumulated in fp32 (~22 mantissa bits).  Every split-precision image and
 *                 row of every entry point stores lo = fp16(v - hi), NOT rescaled (fp16 subnormals go through the
 *                 matrix core at full value, lh_selftest_fp16_subnormal checks it): the recurrent kernels since ABI 6,
 *                 the other separator kernels since ABI 9, the embedder entry points (lh_emb_*) since ABI 10;
 *                 w_pk = fp16 image [dirs][4 waves][4 gates][4 ksteps][64 lanes][hi 8 | lo 8] (lo unscaled) of
 *                 [W_ih * ln_w | W_hh] and b_sum = b_ih + b_hh + W_ih ln_b, rows of both scaled by the exponent factor of their gate (-log2 e for i, f, o;
 *                 -2 log2 e for g: weights.py gate_prescale). The LayerNorm affine is folded into
 *                 the image (weights.py pack_block), the kernel only standardises x; ln_w/ln_b are ignored  */
enum { LH_GEMM_F32 = 0, LH_GEMM_F16X3 = 1 };

/* ABI version of this header; bumped on any signature change. */
int lh_abi_version(void);

/* Launch-shape tuning knobs (benchmark A/B only; 0 = automatic): key 0 = sequences-per-workgroup/16 of the
 * intra LSTM, key 1 = same for the inter LSTM, key 2 = fused intra kernel (0 = k_intra_xp, the hand-ordered
 * software-pipelined step; 2 = the previous k_ln_lstm_lin), key 3 = 0 switches off the issue-priority de-phasing of
 * the two workgroups that share a CU in k_ln_lstm_lin (default on), key 4 = query frames per attention workgroup
 * (1 / 2 = tiles of 16, 3 = 40 frames in three tiles; 0 = automatic), key 5 = fused inter kernel (0 = k_inter_xp,
 * the hand-ordered step with per-phase issue priority; 2 = the previous k_lstm_lin8p), key 6 = runs of consecutive tiles per utterance in
 * lh_deconv_istft (0 = automatic: 256 / B), key 7 = bytes of dynamic LDS added to k_intra_xp launches (timing probe:
 * one workgroup per CU), key 8 = issue priority in k_intra_xp (0 = none, 1 = static priority for the odd wave slot of a SIMD: default, 2 / 3 =
 * every wave raised during the on-chain / off-chain phase of the step), key 9 = issue priority in k_inter_xp (0 = none,
 * 1 / 2 = the LayerNorm / projection waves, 3 = every wave during the on-chain phase: default, 4 = off-chain phase; + 16 = the
 * two roles on the other four waves: A/B only, 6-8 % slower),
 * key 16 = issue priority for the on-chain MFMAs of the embedder's recurrent kernel k_emb_rec (0 = off: default, 1 = on). */
int lh_set_tuning(int key, int value);

/* Validates model_params (reference net.py:21-49 / configs/tsh.json:5-19) against the compiled constants. */
int lh_check_config(int nfft, int hop, int n_mics, int emb_dim, int n_blocks_unused, int lstm_hidden,
                    int n_heads, int attn_window, int n_srcs, int spk_emb_dim);

/* A.1  STFT analysis + re/im channel split + causal 3x3 Conv2d(4->64).
 * Replaces tfgridnet_causal.py:229-242 (asteroid Encoder conv1d, cat/transpose, conv_buf halo, self.conv).
 *   x            [B][2][n_samples]                 n_samples = 128*T + 64
 *   conv_buf_in  [B][4][2][97]   conv_buf_out same shape (last two frames of the halo-extended spectrum)
 *   wfb_pk       split-precision B image [13 tiles][6 ksteps][64 lanes][hi 8 | lo 8] of enc.filterbank._filters
 *                [194 rows -> 208][192 samples]
 *   wconv_pk     split-precision B image [4 tiles][2 ksteps][64 lanes][hi 8 | lo 8] of conv.0.weight as [64 channels]
 *                x [K = 64]:  k = 16 kf + 4 kt + ch  (kf, kt < 3, ch < 4), zero elsewhere  (weights.py pack_all);
 *                bconv [64]
 *   z            [B][T][97][64]  out
 */
int lh_stft_conv_in(const float* x, const float* conv_buf_in, float* conv_buf_out, const void* wfb_pk,
                    const void* wconv_pk, const float* bconv, float* z, int B, int T, int n_samples,
                    lh_stream_t stream);

/* A.2  speaker gain  g = LayerNorm_6208(W e + b)  stored f-major:  gain[b][f][c] = g[b][c*97+f].
 * Replaces tfgridnet_causal.py:247-248 (embed_to_feats_proj + reshape).
 *   emb [B][256]; w [6208][256]; bias, ln_w, ln_b [6208]; scratch [B][6208] (raw projection, workspace);
 *   gain [B][97][64] out
 */
int lh_embed_proj_ln(const float* emb, const float* w, const float* bias, const float* ln_w, const float* ln_b,
                     float* scratch, float* gain, int B, lh_stream_t stream);

/* A.3.1  intra-frame path: LayerNorm(C) -> BiLSTM over frequency (zero initial state) ; hidden states only.
 * Replaces tfgridnet_causal.py:505-512 (intra_norm, intra_rnn).
 *   x      [B*T][97][64]
 *   ln_w/b [64]
 *   w_pk   [2 dirs][4 waves][4 gates][32 ksteps][64 lanes]  MFMA B-operand image of [W_ih | W_hh]^T (see
 *          lookoncetohear_amd/weights.py: pack_lstm);  b_sum [2][256] = bias_ih + bias_hh
 *   h_out  [B*T*97][128]   (forward hidden in cols 0..63, reverse in 64..127)
 */
int lh_ln_lstm_intra(const float* x, const float* ln_w, const float* ln_b, const void* w_pk, const float* b_sum,
                     float* h_out, int n_frames /* B*T */, int mode, lh_stream_t stream);

/* A.3.2  inter-frame path: LayerNorm(C) -> causal LSTM over time with carried state ; hidden states only.
 * Replaces tfgridnet_causal.py:521-532 (inter_norm, transpose/reshape to [B*F,T,C], inter_rnn, h0/c0 in/out).
 *   x [B][T][97][64]; h0,c0,hN,cN [B*97][64] (sequence index b*97+f); w_pk [1][4][4][32][64]; b_sum [256]
 *   h_out [B*T*97][64] in the same (b,t,f) row order as x
 */
int lh_ln_lstm_inter(const float* x, const float* ln_w, const float* ln_b, const void* w_pk, const float* b_sum,
                     const float* h0, const float* c0, float* hN, float* cN, float* h_out, int B, int T, int mode,
                     lh_stream_t stream);

/* A.3.1 for very few frames (streaming, batch-1): same result as lh_ln_lstm_intra, one workgroup per (frame, direction):
 * the input half of all 97 steps as one MFMA GEMM, the recurrent half as a 256 x 64 fp32 mat-vec per step.
 *   x [n_frames][97][64]; h_out [n_frames*97][128]
 *   wih_pk fp16 hi/lo B image [2 dirs][16 ntiles][2 ksteps][64 lanes][16] of W_ih * ln_w with the gate columns in the
 *   kernel's order (column n = PyTorch row (n&3)*64 + (n>>5)*8 + ((n>>3)&3)*2 + ((n>>2)&1));  b_sum [2][256] in the
 *   same order;  whh [2][512][32] fp32 = the same rows of W_hh cut into their two k halves (weights.py pack_block:
 *   intra_s_*)
 */
int lh_intra_stream(const float* x, const void* wih_pk, const float* b_sum, const float* whh, float* h_out,
                    int n_frames, lh_stream_t stream);

/* A.3.2 + Linear fused for few sequences (the host uses it while they fit one round of CUs: batch <= 2): same result as lh_inter_block, one workgroup per sequence
 * (b, f), 64-step chunks: input half and output projection as MFMA GEMMs around a mat-vec recurrence.
 *   x, out [B][T][97][64] (must not alias); h0, c0, hN, cN [B*97][64]
 *   wih_pk [16 ntiles][2 ksteps][64 lanes][16], b_sum [256], whh [512][32]: the lh_intra_stream layouts for the inter
 *   LSTM (weights.py pack_block: inter_s_*);  wlin_pk [4][2][64][16], blin [64] as lh_inter_block
 */
int lh_inter_matvec(const float* x, const void* wih_pk, const float* b_sum, const float* whh, const void* wlin_pk,
                    const float* blin, const float* h0, const float* c0, float* hN, float* cN, float* out, int B, int T,
                    lh_stream_t stream);

/* A.3.1 + Linear fused (split-precision mode): LayerNorm -> BiLSTM over frequency -> Linear(128->64) -> + residual in
 * ONE kernel; replaces tfgridnet_causal.py:505-516.  A workgroup runs the forward then the reverse direction over its
 * sequences and accumulates both halves of the projection into the same output rows (no hidden-state round trip).
 *   x, out [B*T][97][64] (must not alias); w_pk / b_sum as lh_ln_lstm_intra in LH_GEMM_F16X3 mode;
 *   wlin_pk [2 passes][4 ntiles][2 ksteps][64 lanes][hi 8 | lo 8] = intra_linear.weight[:, 0:64] and [:, 64:128]
 *   (weights.py pack_linear_f16x3(unscaled=True): lo = fp16(w - hi)); blin [64]
 */
int lh_intra_block(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin,
                   float* out, int n_frames, lh_stream_t stream);

/* A.3.2 + Linear fused: LayerNorm -> causal LSTM over time (state in/out) -> Linear(64->64) -> + residual;
 * replaces tfgridnet_causal.py:521-538.   x, out [B][T][97][64]; wlin_pk [4][2][64][16] fp16 hi/lo (lo unscaled,
 * weights.py `inter_lin_wu`); blin [64]
 *   w_pk   [8 waves][2 tiles][4 ksteps][64 lanes][hi 8 | lo 8] fp16 (weights.py pack_lstm_f16x3_w8, `inter_w8`): the
 *          eight-wave kernel multiplies transposed (weights = MFMA A operand); lane l of (wave v, tile m, kstep ks)
 *          holds row gate*64 + unit of [W_ih * ln_w | W_hh], gate = (l & 15) & 3, unit = 8v + 2 ((l & 15) >> 2) + m,
 *          at k = 32 ks + 8 (l >> 4) + j; lo unscaled, rows scaled by the gate's exponent factor;
 *   b_sum  [256] in PyTorch gate order (i, f, g, o) x 64, same scaling (weights.py `inter_b16`)
 */
int lh_inter_block(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin,
                   const float* h0, const float* c0, float* hN, float* cN, float* out, int B, int T,
                   lh_stream_t stream);

/* Row-wise Linear(K->64) + bias + residual:  out[r][:] = res[r][:] + W h[r][:] + b.
 * Replaces intra_linear + residual (tfgridnet_causal.py:513-516, K=128) and inter_linear + view/transpose +
 * residual (:534-538, K=64).   h [rows][K]; bias [64]; res,out [rows][64];
 *   w_pk  split-precision fp16 image [4 ntiles][K/32 ksteps][64 lanes][hi 8 | lo 8] (weights.py pack_linear_f16x3)
 */
int lh_linear_res(const float* h, const void* w_pk, const float* bias, const float* res, float* out, int rows,
                  int K, lh_stream_t stream);

/* Split-precision activation rows shared by lh_qkv_proj_ln / lh_local_attn / lh_ring_pack / lh_ring_unpack.
 * Every value v is stored as two fp16 numbers  hi = fp16(v), lo = fp16(v - hi)  (4 bytes per element,
 * like fp32; v ~ hi + lo keeps ~22 mantissa bits for |v| >= 2^-3 and an absolute 2^-25 below) in the order the
 * attention MFMA operands consume:
 *   q   [B*4][T][1216 halves]          per row 76 blocks of 8 features (f*6+e), each [hi 8 | lo 8]; features
 *                                      582..607 are zero
 *   kx  [B*4][T+49+PAD][1216 halves]   same row format; rows 0..48 = history (K_buf), row 49+t = K[t]
 *   vx  [B*4][T+49+PAD][3104 halves]   per row 388 quads of 4 columns (f*16+v), each [hi 4 | lo 4]
 * PAD = LH_KV_PAD_ROWS rows behind row T+48 that the CALLER zero-fills once and the library never writes: attention
 * tiles read (without needing them) up to 47 rows past their last key. */
#define LH_KV_PAD_ROWS 48

/* A.3.3  Q/K/V: pointwise Linear + PReLU, head split, joint LayerNorm over (f,e) per head.
 * Replaces attn_conv_Q/K/V (tfgridnet_causal.py:354-387, used :547-551) and the K/V history concat (:553-562):
 * K and V rows are written at row (49 + t) of the history-extended buffers.
 *   y      [B][T][97][64]
 *   w_pk   fp16 hi/lo image [7 ntiles][2 ksteps][64 lanes][16] of the stacked weight, rows 0..23 Q(h*6+e),
 *          24..47 K, 48..111 V(h*16+v); bias [112]
 *   slopes [3] PReLU slopes (Q,K,V);  lnq_w/b, lnk_w/b [608] = the 582 affine values zero-padded;  lnv_w/b [1552]
 *   q, kx, vx  split-precision rows (above)
 *   ring_pos   NULL, or (T = 1 only) a device counter: the K / V row goes to slot (*ring_pos mod 50) of a persistent
 *              50-row ring instead of row 49 — rows 0..49 are then exactly the window of the one query frame, in
 *              rotated order (softmax and P.V are order-free), and no history row ever has to be moved
 */
int lh_qkv_proj_ln(const float* y, const void* w_pk, const float* bias, const float* slopes, const float* lnq_w,
                   const float* lnq_b, const float* lnk_w, const float* lnk_b, const float* lnv_w,
                   const float* lnv_b, void* q, void* kx, void* vx, const int* ring_pos, int B, int T,
                   lh_stream_t stream);

/* A.3.5  local windowed attention over exactly 50 slots (frames t-49..t incl. history rows, no mask) with
 * the head merge fused into the store.  Replaces tfgridnet_causal.py:564-581 without materialising the
 * 50x unfolded K/V (`_causal_unfold_chunk`, :429-454).
 *   q, kx, vx  split-precision rows (above)
 *   merged [B][T][4][97][16]   merged[b][t][h][f][v] = O[b*4+h][t][f*16+v]   (head-major frame slabs)
 */
int lh_local_attn(const void* q, const void* kx, const void* vx, float* merged, int B, int T, lh_stream_t stream);

/* A.3.4  streaming state <-> history rows (tfgridnet_causal.py:553-562).  The reference carries the last 49 K / V
 * rows as fp32 state:  lh_ring_pack writes k_buf [B*4][49][582] / v_buf [B*4][49][1552] into rows 0..48 of kx / vx
 * before lh_qkv_proj_ln;  lh_ring_unpack reads rows T..T+48 (the new history) back into fp32 state tensors
 * (hi + lo, i.e. exactly the values the attention kernel used). */
int lh_ring_pack(const float* k_buf, const float* v_buf, void* kx, void* vx, int B, int T, lh_stream_t stream);
int lh_ring_unpack(const void* kx, const void* vx, float* k_buf, float* v_buf, int B, int T, lh_stream_t stream);
/* Streaming ring slot counter (lh_qkv_proj_ln's ring_pos): *ring_pos = (*ring_pos + 1) mod modulo, on the device and in
 * stream order — one node of the captured per-chunk graph (the reference has no counterpart: it shifts K_buf / V_buf by
 * one row per chunk, tfgridnet_causal.py:553-562).  modulo in [1, 2^30]. */
int lh_ring_advance(int* ring_pos, int modulo, lh_stream_t stream);

/* A.3.6  attn_concat_proj: Linear(64->64)+PReLU, joint LayerNorm over (f,c), residual; optional speaker gain.
 * Replaces tfgridnet_causal.py:583-588 and, when gain != NULL, the `batch = batch * embed` applied to the
 * input of block 1 (:250-251):  out = (y2 + LN(PReLU(W m + b))) * gain[b][f][c].
 *   merged [B][T][4][97][16] (lh_local_attn's output order); y2, out [B][T][97][64]; w_pk fp16 hi/lo image [4][2][64][16]; bias [64]; slope [1]; ln_w/b [6208]; gain [B][97][64]|NULL
 */
int lh_proj_ln_res(const float* merged, const void* w_pk, const float* bias, const float* slope,
                   const float* ln_w, const float* ln_b, const float* y2, const float* gain, float* out, int B,
                   int T, lh_stream_t stream);

/* A.4  causal ConvTranspose2d(64->4,3x3) + spectrum re-pack + iSTFT synthesis/overlap-add.
 * Replaces tfgridnet_causal.py:256-273 and the look-ahead trim of net.py:61.
 *   y [B][T][97][64]; deconv_buf_in/out [B][64][2][97]; istft_buf_in/out [B][2][194][1]
 *   wdec_pk fp16 hi/lo B image [3 ntiles][2 ksteps][64 lanes][16] of deconv.weight as [(kt,kf,o) 36 -> 48] x [64 c];
 *   bdec [4]; wfb_dec fp16 hi/lo B image [12 ntiles][7 ksteps][64 lanes][16] of dec.filterbank._filters^T
 *   [192 samples] x [194 -> 224 rows]   (weights.py pack_linear_f16x3)
 *   wave_out [B][2][128*T];  range_flag: the caller's two-word flag (range contract above) or NULL;
 *   keep_nonfinite (ABI 13): what a non-finite output sample is stored as — 1: as it is (inf / NaN reach the caller exactly
 *   as from the reference's plain-fp32 forward, nothing is hidden; the offline host `Net` passes 1), 0: as 0 (silence, not
 *   NaN, reaches a listener; the streaming host passes 0).  The flag is raised either way.
 */
int lh_deconv_istft(const float* y, const float* deconv_buf_in, float* deconv_buf_out, const float* istft_buf_in,
                    float* istft_buf_out, const void* wdec_pk, const float* bdec, const void* wfb_dec,
                    float* wave_out, unsigned* range_flag, int keep_nonfinite, int B, int T, lh_stream_t stream);

/* ---- time windows (ABI 14) ------------------------------------------------------------------------------------------------
 * Every stage of a GridNetBlock is causal in time (tfgridnet_causal.py:489-590): the intra pass is per frame, the inter LSTM
 * carries (h, c) (:521-532), the attention sees the 49 previous K / V rows (:553-562) and the frame stages are per frame.  The
 * `_win` entry points run the SAME kernels on frames [t0, t0 + Tc) of every utterance of buffers laid out for T frames
 * ([B][T][97][64] activations, q [4B][T][..], kx / vx [4B][T + 49 + PAD][..]); `lh_X(..., B, T, s)` is `lh_X_win(..., B, T, 0,
 * T, s)`.  A host can then cut the time axis and run block i on window k + 1 beside block i + 1 on window k on separate
 * streams (lookoncetohear_amd/net.py, `Net.time_chunks`): the inter LSTM's 625-step dependent chain, which fills only 194 of
 * the 256 CUs, overlaps with the other stages.  State hand-over between windows of one block, all on the device:
 *   lh_inter_block_win   (hN, cN) of window k are (h0, c0) of window k + 1 (distinct buffers per window).  `carry`: bit 0 =
 *                        c0 holds the kernel's internal cell state (-2 log2(e) c, written by the previous window with bit 1),
 *                        bit 1 = cN is written in that form; 0 = both are the reference's c (tfgridnet_causal.py:526-532).  With
 *                        the inner boundaries carried in the internal form the windows reproduce the whole-clip launch bit
 *                        for bit (c / k followed by k * c is two roundings);
 *   lh_qkv_proj_ln_win   writes rows 49 + t of kx / vx, lh_local_attn_win of window k + 1 reads rows t0 .. of the same
 *                        buffers: the history IS the previous window's rows, nothing is packed or unpacked.  (Attention tiles
 *                        read — never need — up to 47 rows past the window: they must hold finite values, e.g. the zeros of the
 *                        allocation or an earlier forward's rows.)
 * Tc >= 2 for lh_inter_block_win; t0 + Tc <= T. */
int lh_intra_block_win(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin,
                       float* out, int B, int T, int t0, int Tc, lh_stream_t stream);
int lh_inter_block_win(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin,
                       const float* h0, const float* c0, float* hN, float* cN, float* out, int B, int T, int t0, int Tc,
                       int carry, lh_stream_t stream);
/* the unfused intra pair of small batches on a window (split-precision mode): lh_ln_lstm_intra + lh_linear_res on frames (b, t0 + j);
 * h_out / h [B*T*97][128] in the (b, t, f) row order of x */
int lh_ln_lstm_intra_win(const float* x, const float* ln_w, const float* ln_b, const void* w_pk, const float* b_sum, float* h_out,
                         int B, int T, int t0, int Tc, lh_stream_t stream);
int lh_linear_res_win(const float* h, const void* w_pk, const float* bias, const float* res, float* out, int B, int T, int t0,
                      int Tc, int K, lh_stream_t stream);
/* the few-sequences form of the inter stage (lh_inter_matvec) on a window; `carry` as above.  Windows that start on multiples of
 * its 64-step chunk reproduce the whole-clip launch bit for bit. */
int lh_inter_matvec_win(const float* x, const void* wih_pk, const float* b_sum, const float* whh, const void* wlin_pk,
                        const float* blin, const float* h0, const float* c0, float* hN, float* cN, float* out, int B, int T,
                        int t0, int Tc, int carry, lh_stream_t stream);
int lh_qkv_proj_ln_win(const float* y, const void* w_pk, const float* bias, const float* slopes, const float* lnq_w,
                       const float* lnq_b, const float* lnk_w, const float* lnk_b, const float* lnv_w,
                       const float* lnv_b, void* q, void* kx, void* vx, const int* ring_pos, int B, int T, int t0, int Tc,
                       lh_stream_t stream);
int lh_local_attn_win(const void* q, const void* kx, const void* vx, float* merged, int B, int T, int t0, int Tc,
                      lh_stream_t stream);
int lh_proj_ln_res_win(const float* merged, const void* w_pk, const float* bias, const float* slope, const float* ln_w,
                       const float* ln_b, const float* y2, const float* gain, float* out, int B, int T, int t0, int Tc,
                       lh_stream_t stream);

/* ---- plain-fp32 reference kernels of the frame stages (gemm_mode "f32all"; lh_ref32.hip) ----------------------------------
 * NON-PRODUCT: test / diagnosis kernels that ship in the library so that a host can bisect a disagreement with a real
 * checkpoint without a second build.  Never benchmark them, never route a product path through them.
 * The product frame kernels above are split-precision (fp16 hi + lo, ~22 bits) in EVERY arithmetic mode; with these and the
 * exact fp32-MFMA recurrences (lh_ln_lstm_intra / _inter in LH_GEMM_F32) a forward exists whose every contraction is an
 * fp32 fmaf chain like the reference's (tfgridnet_causal.py:188-283): the A/B that separates split-precision error from a
 * kernel bug on a real checkpoint.  Written for obviousness (one thread per output, natural summation order), ~50x slower
 * than the product path, test / diagnosis only.  Weights are the PyTorch tensors of the state dict as they are (no packed
 * images); activations channel-last [B][T][97][64] like everywhere else; Q / K / V plain fp32:
 *   q [4B][T][582], kx [4B][T+49][582], vx [4B][T+49][1552] (rows 0..48 = K_buf / V_buf, copied by the caller).
 * lh_ref32_stft_conv_in   :229-242   filters = enc.filterbank._filters [194][1][192]; conv_w [64][4][3][3]; spec_scratch
 *                                    [B][4][T+2][97]
 * lh_ref32_linear         out[r][n] = (res[r][n] +) act(bias[n] + sum_k in[r][k] w[n][k]); slope = PReLU weight or NULL
 * lh_ref32_head_ln        :360-376   head split + LayerNorm over (f, d) of columns col0 + h*D + d of `pre` [B][T][97][ncol]
 *                                    into dst[(b*4+h)][row0 + t][f*D + d]; D = 6 (Q, K) or 16 (V)
 * lh_ref32_local_attn     :564-581   50 slots, no mask; merged [B][T][4][97][16]
 * lh_ref32_proj_ln_res    :583-588, 250-251   w [64][64]; rows_scratch, pre_scratch [B*T*97][64]
 * lh_ref32_deconv_istft   :256-273, net.py:61   deconv_w [64][4][3][3]; filters = dec.filterbank._filters; sx_scratch
 *                                    [B][2][T+1][194]; non-finite samples are stored as they are (no flag)
 */
int lh_ref32_stft_conv_in(const float* x, const float* conv_buf_in, float* conv_buf_out, const float* filters,
                          const float* conv_w, const float* conv_b, float* spec_scratch, float* out, int B, int T,
                          int n_samples, lh_stream_t stream);
int lh_ref32_linear(const float* in, const float* w, const float* bias, const float* slope, const float* res, float* out,
                    int rows, int K, int N, lh_stream_t stream);
int lh_ref32_head_ln(const float* pre, int ncol, int col0, int D, const float* ln_w, const float* ln_b, float* dst,
                     int row0, int rows_per_bh, int B, int T, lh_stream_t stream);
int lh_ref32_local_attn(const float* q, const float* kx, const float* vx, float* merged, int B, int T, lh_stream_t stream);
int lh_ref32_proj_ln_res(const float* merged, const float* w, const float* bias, const float* slope, const float* ln_w,
                         const float* ln_b, const float* y2, const float* gain, float* rows_scratch, float* pre_scratch,
                         float* out, int B, int T, lh_stream_t stream);
int lh_ref32_deconv_istft(const float* y, const float* deconv_buf_in, float* deconv_buf_out, const float* istft_buf_in,
                          float* istft_buf_out, const float* deconv_w, const float* deconv_b, const float* filters,
                          float* sx_scratch, float* wave_out, int B, int T, lh_stream_t stream);

/* ---- enrollment embedder (reference src/models/tfgridnet_orig/tfgridnet.py:88-127 + espnet2 TF-GridNet trunk) ----
 * Front end, tfgridnet.py:109-117: x / std(x) (unbiased, over samples and mics), STFT(n_fft 128, hop 64, hann, centred
 * with reflect padding), re/im channel stacking, Conv2d(4->64, 3x3, padding 1), GroupNorm(1, 64).
 *   x [B][2][n_samples]; inv_std [B] (out); wfb_pk fp32 MFMA image [9][32][64] of the windowed DFT rows [128 x 130];
 *   wconv_pk [4][9][64] of conv.0.weight as [36 taps] x [64]; bconv, gn_w, gn_b [64];
 *   gn_part fp64 scratch [B * ceil(T/14)][2]; z [B][T][65][64] out, T = n_samples/64 + 1
 *   xsplit_next  NULL, or 2*B*T*65*64 fp16: z channel-normalised and split, for lh_emb_axis_fused(have_xsplit = 1)
 */
int lh_emb_frontend(const float* x, float* inv_std, const float* wfb_pk, const float* wconv_pk, const float* bconv,
                    const float* gn_w, const float* gn_b, double* gn_part, float* z, void* xsplit_next, int B, int T,
                    int n_samples, lh_stream_t stream);

#ifdef LH_LEGACY  /* A/B lab builds only (-DLH_LEGACY): not exported by the product library */
/* One axis path of an espnet2 GridNetBlock of the embedder (intra: inter = 0, sequences = frames, scan over the 65
 * bins; inter = 1: sequences = bins, scan over time): LayerNorm(C) -> unfold(4) -> BiLSTM(256 -> 64) ->
 * ConvTranspose1d(128 -> 64, 4) -> + residual, as three launches (input GEMM over all windows, recurrence, gather-GEMM).
 *   x, out [B][T][65][64] (no alias); wih_pk fp16 hi/lo image [32][8][64][16] (both directions, LN affine folded,
 *   features window-major, columns (dir, unit, gate)); bih [512]; whh_pk [2][4][4][2][64][16]; wct_pk [4][16][64][16]
 *   of the taps as [64] x [4*128]; bct [64]; xsplit scratch 2*B*T*65*64 fp16 (hi | lo images of the channel-
 *   normalised input); gx scratch [nseq*P][512]; hbuf scratch [nseq*P][128] (P = L - 3)
 */
int lh_emb_axis(const float* x, const void* wih_pk, const float* bih, const void* whh_pk, const void* wct_pk,
                const float* bct, void* xsplit, float* gx, float* hbuf, float* out, int B, int T, int inter,
                lh_stream_t stream);
#endif /* LH_LEGACY */

/* The same axis path without the gate pre-activation round trip (round 4): LayerNorm + split -> ONE recurrent kernel that
 * computes the 256-wide input half one step ahead of the dependent chain (weights resident in registers, one workgroup of
 * 16 sequences x one direction per CU) and writes the hidden states as fp16 hi | lo images -> transposed-conv GEMM + residual.
 *   wrec_pk  fp16 [2 dirs][8 waves][40 fragments][64 lanes][8]: MFMA A fragments of [W_ih (4 window slots) | W_hh], rows
 *            ordered (unit, gate), LayerNorm gamma and the gates' exponent factors folded in (embed_net.py pack_rec)
 *   brec     [2][256] in (unit, gate) order, same folding;  wct_pk, bct as lh_emb_axis
 *   xsplit   scratch 2*B*T*65*64 fp16;  hsplit scratch 2*nseq*P*128 fp16 (hi | lo images of the hidden states)
 *   have_xsplit  non-zero: xsplit already holds the channel-normalised, split x (left there by the previous axis call's
 *            emit_split or by lh_emb_attn_block's xsplit_next) and the normalisation launch is skipped
 *   emit_split   non-zero: xsplit is overwritten with the normalised, split OUT rows once the recurrence has consumed it */
int lh_emb_axis_fused(const float* x, const void* wrec_pk, const float* brec, const void* wct_pk, const float* bct,
                      void* xsplit, void* hsplit, float* out, int B, int T, int inter, int have_xsplit, int emit_split,
                      lh_stream_t stream);

/* The inter axis for SMALL batches (ABI 14): as lh_emb_axis_fused(inter = 1) with the recurrence on one workgroup per (sequence,
 * direction) — the quad-lane mat-vec step of the separator's batch-1 kernels, 0.39 us per step against 1.4 us for a 16-sequence
 * tile that a single enrollment fills with 9 workgroups.  Same result to fp32 rounding.
 *   wih_pk  fp16 hi/lo B image [2 dirs x 16 ntiles][8 ksteps][64 lanes][16] of W_ih' [256 x 256] (LayerNorm gamma folded, K = window
 *           slot*64 + channel in natural tap order, columns (direction, unit, gate));  bih [2][256] same column order
 *           (b_ih + b_hh + W_ih beta);  whh fp32 [2][256][64], row 4 unit + gate;  the rest as lh_emb_axis_fused */
int lh_emb_axis_mv(const float* x, const void* wih_pk, const float* bih, const float* whh, const void* wct_pk,
                   const float* bct, void* xsplit, void* hsplit, float* out, int B, int T, int have_xsplit, int emit_split,
                   lh_stream_t stream);

/* Enrollment embedder, attention branch of one GridNetBlock (espnet2 GridNetBlock.forward attention part, restated in
 * oracle/embedder_oracle.py:149-168): per-head Q/K/V 1x1 conv + PReLU + LayerNorm over (channel, bin), full T x T
 * softmax attention per (head, utterance), head merge, attn_concat_proj (1x1 conv + PReLU + LayerNorm) + residual.
 * Tp = T rounded up to a multiple of 64.
 *   y2, out  [B][T][65][64];  merged scratch [B][T][65][64]
 *   q, k     scratch fp16 [2][4B][T][544] (hi | lo images);  v scratch fp32 [4B][T][1040]
 *   vt       scratch fp16 [2][4B][1040][Tp];  sc scratch fp32 [4B][T][Tp];  p scratch fp16 [2][4B][T][Tp]
 *   wqkv_pk  fp16 hi/lo image [8][2][64][16] of the stacked conv weights [128 x 64] (Q h*8+e | K | V h*16+v)
 *   bqkv, slopes [128] (PReLU slope of each output column's head conv)
 *   lnq_*, lnk_* [4][520], lnv_* [4][1040]: LayerNorm affine re-ordered to (bin*d + channel)
 *   wproj_pk [4][2][64][16]; bproj [64]; slope_p [1]; lnp_* [4160] re-ordered to (bin*64 + channel)
 *   xsplit_next  NULL, or 2*B*T*65*64 fp16: the channel-normalised, split `out` rows for the next block's intra axis call
 *            (lh_emb_axis_fused have_xsplit)
 */
int lh_emb_attn_block(const float* y2, const void* wqkv_pk, const float* bqkv, const float* slopes,
                      const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                      const float* lnv_w, const float* lnv_b, const void* wproj_pk, const float* bproj,
                      const float* slope_p, const float* lnp_w, const float* lnp_b, void* q, void* k, float* v,
                      void* vt, float* sc, void* p, float* merged, float* out, void* xsplit_next, int B, int T,
                      lh_stream_t stream);

/* Enrollment embedder head (reference src/models/tfgridnet_orig/tfgridnet.py:120-127): Linear(65*64 -> 256) per
 * frame, LayerNorm(256), mean over frames.
 *   z [B][T][65][64];  w_pk fp16 hi/lo image [16][130][64][16] of the weight with inputs re-ordered to (bin*64 + c)
 *   bias, ln_w, ln_b [256];  part scratch [B][ceil(T/64)][256];  emb [B][256]
 */
int lh_emb_head(const float* z, const void* w_pk, const float* bias, const float* ln_w, const float* ln_b,
                float* part, float* emb, int B, int T, lh_stream_t stream);

/* Binaural rendering, the data-pipeline step before the separator (reference src/datasets/multi_ch_simulator.py:40-61
 * `_convolve` = `scipy.signal.convolve(src, rir[ear])[:len(src)]` per source and ear;
 * src/datasets/MixLibriSpeechNoisyEnrollNorm.py:176-202 noise scale, peak normalisation, mixture, target).
 *   src      [B][S1][N]       mono rows: the sources first, the noise bed LAST
 *   rir      [B][S1][2][Lh]   impulse response per row and ear (shorter ones zero-padded to Lh)
 *   gain     [B][S1]          applied after the convolution: 1 for sources, `noise_scale` for the noise row
 *   tgt_idx  [B] int32        row returned as `target`
 *   events   [B][S1][2][N]    out: rendered rows before peak normalisation
 *   peak     [B] uint32       out: bit pattern of the fp32 peak of |mixture| (the reference's `norm_factor`)
 *   mixture, target [B][2][N] out: divided by the peak when it exceeds 1 (IEEE division, reference order of sums)
 */
int lh_render_binaural(const float* src, const float* rir, const float* gain, const int* tgt_idx, float* events,
                       unsigned* peak, float* mixture, float* target, int B, int S1, int N, int Lh, lh_stream_t stream);

/* Eval metrics on the device (reference src/ts_hear_test.py:139-146, torchmetrics SI-SNR restated): per utterance
 * output_sisnr, si_snr_i (both averaged over the 2 channels) and cosine(embedding, embedding_gt); fp64 moments.
 *   outputs, target, mixture [B][2][n_samples]; emb, emb_gt [B][emb_dim]
 *   scratch  fp64 workspace, B*2*16*8 + B*3 doubles
 *   rows     [B][3] fp32 = (output_sisnr, si_snr_i, embedding_sim)   (the CSV columns of ts_hear_test.py:149-151)
 *   sums     [4] fp64 = (sum si_snr_i, sum output_sisnr, sum embedding_sim, B): the all-reduce payload
 */
int lh_metric_sums(const float* outputs, const float* target, const float* mixture, const float* emb,
                   const float* emb_gt, double* scratch, float* rows, double* sums, int B, int n_samples,
                   int emb_dim, lh_stream_t stream);

/* The path's ONE exchange step, for hosts that drive this ABI without Python: all-reduce (sum) of the
 * fp64 metric sums written by lh_metric_sums over one process per GPU — RCCL over xGMI, 32 bytes, latency-bound.
 * Replaces the reference's Lightning `sync_dist` all-reduce (src/ts_hear_embed_pl_module.py:82-107).  RCCL is bound
 * with dlopen at the first call (LOOKONCE_RCCL_LIB overrides the name; an RCCL already mapped into the process is
 * reused): LH_ERR_UNSUPPORTED when no RCCL can be found.  The Python host uses torch.distributed instead.
 *   lh_comm_unique_id  rank 0: 128 opaque bytes to hand to every rank through the host's own channel (ncclGetUniqueId)
 *   lh_comm_init       collective over all ranks, on each rank's current device (ncclCommInitRank)
 *   lh_allreduce_f64   in place, on `stream` (ncclAllReduce, ncclFloat64, ncclSum)
 */
int lh_comm_unique_id(void* id128);
int lh_comm_init(const void* id128, int n_ranks, int rank, void** comm);
int lh_allreduce_f64(void* comm, double* buf, int count, lh_stream_t stream);
int lh_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* LOOKONCE_HIP_H */
