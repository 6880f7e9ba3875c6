/* marl_hip.h - C ABI of the MI355X (gfx950) hot path for Skylarking/MARL.
 *
 * The reference has no FFI layer (it is pure Python/PyTorch); the functions below are what a
 * binding for its hot path would call, one per fused torch-op sequence.
 * Each entry cites the reference code it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless typed otherwise; row-major; sizes in elements
 *   - `stream` is a hipStream_t passed as void*; nothing synchronises, allocates or frees
 *   - return value: 0 on success, otherwise a hipError_t
 *   - workspaces are caller-provided; *_workspace() gives the byte count
 *   - (B,T,N,*) arrays are episode-major: row = (b*T + t)*N + n
 */
#ifndef MARL_HIP_H
#define MARL_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Virtual row-major matrix [M,K] = [dense0 | dense1 | nhot one-hot blocks | agent-id block].
 * Replaces the th.cat([...]) input builders: controller/share_params.py:84-112,
 * network/mixer.py:151-153 (QPLEX [s|actions]), :380,:386 (QTRAN [h|u], [s|enc]), :416.
 * Row remap (rpe != 0): source row = (row / rpe) * bs + (row % rpe) + off, and a row whose
 * (row % rpe) + off < 0 reads as zero / "no action" (used for (T+1)-slot episode storage and
 * for the one-step-shifted last action). */
typedef struct {
  const float* p0; long ld0; int k0;        /* dense segment 0, k0 columns */
  const float* p1; long ld1; int k1;        /* dense segment 1 */
  const int* idx; int nhot; int hot_w;      /* column j*hot_w + idx[row*nhot + j] = 1 (idx < 0: none) */
  int nid;                                  /* column (row % nid) = 1 */
  const float* m0; long ldm0;               /* optional relu gate on segment 0: v * (m0 > 0) */
  long rpe0, bs0, off0;                     /* row remap for p0 (and m0) */
  long rpei, bsi, offi;                     /* row remap for idx */
  const int* emap0;                         /* optional episode map of the p0 remap (needs rpe0 > 0): episode
                                               e = row / rpe0 reads storage episode emap0[e] - replay samples are
                                               read in place from the ring (common/replaybuffer.py:54-60) */
} marl_src_t;

/* Batched ("grouped") launches over equally shaped problems whose operands sit at a constant
 * element stride (QPLEX's 10 attention heads, network/mixer.py:117-145). 0 = shared operand. */
typedef struct {
  int groups;
  long gs_x0, gs_x1, gs_w, gs_b, gs_y, gs_m0;
} marl_group_t;

/* RNNQNet parameters (network/q_network.py:12-14), torch layouts. */
typedef struct {
  const float *fc1_w, *fc1_b;   /* (H,I), (H)   */
  const float *w_ih, *w_hh;     /* (3H,H) gate order r,z,n */
  const float *b_ih, *b_hh;     /* (3H) */
  const float *fc2_w, *fc2_b;   /* (A,H), (A) */
  int H;                        /* must be 64 */
} marl_agent_weights_t;

/* ---- dense layers on the fp32 matrix cores (gemm.hip) ---------------------------------------
 * Y = act(X W^T + b) [+ beta*Y].  Replaces every nn.Linear(+ReLU) on the path
 * (network/mixer.py:37-55,117-145,200-206,365-375,399-409).  w_kmajor=1 reads W as [K][N]
 * so the same kernel computes dX = dY W (autograd of nn.Linear wrt input). act: 0 none, 1 relu;
 * | 0x100 = round both operands to bf16 and use the bf16 matrix cores (fp32 accumulate) - opt-in for the
 * MIXER layers only (BASELINE config 5 "bf16 mixer with MFMA"; tolerance ~1e-2 instead of 1e-4). */
int marl_linear(const marl_src_t* x, const float* W, long ldw, int w_kmajor, const float* bias,
                float* Y, long ldy, int M, int N, int K, int act, float beta,
                const marl_group_t* grp, void* stream);
/* dW += G^T X, db += colsum(G) with G = dY * (Yact > 0 if Yact). Fixed-order slab reduction.
 * In grp: gs_y = dY stride, gs_m0 = Yact stride, gs_w / gs_b = dW / db strides.  flags: 1 = bf16 operands
 * (as act | 0x100 above). */
int marl_linear_wgrad(const float* dY, long lddy, const float* Yact, long ldya, const marl_src_t* x,
                      float* dW, long lddw, float* db, int M, int N, int K, int flags,
                      const marl_group_t* grp, float* ws, size_t ws_bytes, void* stream);
size_t marl_linear_wgrad_workspace(int M, int N, int K, int groups);
int marl_wgrad_slabs(int M);

/* ---- agent (agent.hip) ----------------------------------------------------------------------
 * T-step unroll of RNNQNet over B*N rows in ONE launch.  Replaces SharedMAC.get_current_q_values /
 * get_next_q_values (controller/share_params.py:125-168) incl. _build_inputs (:84-112), and with
 * T=1 the network call of SharedMAC.choose_action (:37-63).
 *   obs   : row (b,t,n) at obs + ((b*obs_bs) + (t+obs_t0)*N + n)*O        (obs_bs = rows/episode)
 *   ufed  : int32 action fed back at step t: ufed[b*u_bs + (t+u_t0)*N + n]; none if t+u_t0 < 0,
 *           value < 0 or ufed == NULL (one-hot of zeros, share_params.py:96-100)
 *   ep_len: per-episode int32 length or NULL; observations of steps t >= ep_len[b] read as zeros
 *           (the zero padding rollout.py:122-133 writes, needed when obs is (T+1)-slot storage)
 *   ep_map: per-episode int32 storage index or NULL: batch episode b reads obs of storage episode
 *           ep_map[b] (replay samples read in place, common/replaybuffer.py:54-60); ufed / ep_len / outputs
 *           stay indexed by b
 *   h0    : (B*N,64) or NULL = zeros (init_hidden, :74-76); h_last may alias h0
 *   q (B,T,N,A); hs (B,T,N,64) hidden AFTER each step or NULL; saved = (T+1) * R16 * 6 * 64 floats (R16 = B*N rounded up to
 *   16) or NULL: per row-step the 6 vectors hprev,x,r,z,n,hn for the backward pass, in a tile layout private to the two
 *   kernels ([T][16-row tile][plane][16-column tile][lane][4]: one 16-byte access per lane and plane on both sides)
 *   cu_budget: CUs (= workgroups) a T > 1 launch spreads its rows over, 1..256; 0 = 256 = the whole chip.  With 128
 *           the two independent unrolls of an update - eval current-Q and target next-Q (q_learner.py:97,104) - fit
 *           on the chip together and can be launched on two HIP streams; results do not depend on it (rows are
 *           independent).  A per-call argument: the library keeps no process-wide state.
 *   gi_out: NULL, or T * R16 * 3 * 64 floats (same tile layout): with `saved`, the input-side gate sums bias + x W_ih (r | z | n blocks) of every
 *           row-step are stored as well.
 *   gi_in : NULL, or the gi_out buffer an EARLIER unroll of the same weights wrote for the same rows whose step t+1 input
 *           equals this unroll's step t input for t < T-1 (the double-Q pass after the eval pass, q_learner.py:97-110:
 *           observations shifted by one step, same last actions): fc1 and the input-side gate products of those steps - 288
 *           of a row tile's 496 multiplies per step - are not recomputed; every GRU kernel here accumulates the input-side
 *           products before the hidden-side ones, so the result is bit-identical.  Ragged episodes stay exact: a step t at
 *           which a row of the workgroup has ep_len - 1 == t (the earlier unroll saw the zero padding there, this one sees
 *           the final observation) and the last step are computed in full.  Both pointers are ignored where the kernel
 *           chosen for the shape has no such variant (marl_agent_unroll_reuse_supported). */
int marl_agent_unroll_fwd(const marl_agent_weights_t* w, const float* obs, long obs_bs, int obs_t0,
                          const int* ufed, long u_bs, int u_t0, const int* ep_len, const int* ep_map,
                          const float* h0, float* q, float* hs, float* h_last, float* saved, int B, int T,
                          int N, int O, int A, int last_action, int reuse_network, int cu_budget,
                          float* gi_out, const float* gi_in, void* stream);
int marl_agent_unroll_reuse_supported(int B, int T, int N, int O, int A, int cu_budget);

/* Gradient destinations of the recurrent / output layers (accumulated into, torch layouts). */
typedef struct {
  float *w_ih, *w_hh;           /* (3H,H) */
  float *b_ih, *b_hh;           /* (3H)   */
  float *fc2_w, *fc2_b;         /* (A,H), (A) */
} marl_agent_grads_t;

/* BPTT (autograd of the unroll above; q_learner.py:171 loss.backward()), fused: per step the delta
 * pass AND the weight-gradient reductions of W_ih, W_hh, W_2 and their biases (in-register
 * accumulators, one partial slab per workgroup in `ws`, fixed-order reduce => reproducible).
 *   dq (B,T,N,A) gradient on q, OR (dq_idx != NULL) its sparse form: row (b,t,n) has the single non-zero
 *   dq_val[b,t,n] in column dq_idx[b,t,n] (the TD loss reaches q only through th.gather, q_learner.py:100;
 *   the dense tile is then never materialised); dq_idx2 / dq_val2: optional SECOND pair per row (QTRAN reaches q through the
 *   taken and the greedy action, qtran_learner.py:139,145; equal columns add); dq_gdiv > 1: the values are indexed by
 *   row / dq_gdiv (N: one value per (episode, step), shared by its agents - autograd of .sum(dim=-1));
 *   dhs (B,T,N,64) extra gradient on hs or NULL (QTRAN heads)
 *   saved: output of the forward pass (it holds h(t) as well; `hs` is not read any more and may be NULL);
 *   dxp (B,T,N,64) = gradient at the fc1 pre-activation
 * The fc1 gradient follows as ONE marl_linear_wgrad over dxp and the virtual input [obs|u|id]. */
size_t marl_agent_bwd_workspace(int B, int N, int A);
int marl_agent_unroll_bwd(const marl_agent_weights_t* w, const float* dq, const int* dq_idx,
                          const float* dq_val, const int* dq_idx2, const float* dq_val2, int dq_gdiv, const float* dhs,
                          const float* saved, const float* hs, float* dxp, float* dh0,
                          const marl_agent_grads_t* g, float* ws, size_t ws_bytes,
                          int B, int T, int N, int A, void* stream);

/* ---- per-row kernels (mixers.hip) -----------------------------------------------------------*/
/* out[row] = q[row,idx[row]] (th.gather, q_learner.py:100,114); idx < 0 -> 0.  With avail != NULL the
 * value read is the masked one: avail[row,idx]==0 ? mask_val : q (q_learner.py:105 then :114). */
int marl_q_gather(const float* q, const int* idx, const float* avail, float mask_val, float* out,
                  long rows, int A, void* stream);
/* q[avail==0] = mask_val; max / first-index argmax over actions (q_learner.py:105,112-117,125-127;
 * qtran_learner.py:104-113). avail may be NULL. out_max / out_arg may be NULL. */
int marl_q_masked_max(const float* q, const float* avail, float mask_val, float* out_max, int* out_arg,
                      long rows, int A, void* stream);
/* Double-Q selection in one pass (q_learner.py:104-117): arg[row] = first-index argmax of q_sel masked with
 * avail (mask_val where avail == 0), out_val[row] = q_val[row, arg] masked the same way; out_arg may be NULL. */
int marl_q_double_select(const float* q_sel, const float* q_val, const float* avail, float mask_val,
                         float* out_val, int* out_arg, long rows, int A, void* stream);
/* dq = 0; dq[row,idx1[row]] += g1[row/gdiv]; dq[row,idx2[row]] += g2[row/gdiv] (idx2/g2 may be
 * NULL): autograd of gather / max (+ of the sum over agents when gdiv = N). */
int marl_q_scatter(float* dq, const int* idx1, const float* g1, const int* idx2, const float* g2,
                   long rows, int A, int gdiv, void* stream);
/* out = a + b (q_tot = v_tot + a_tot, q_learner.py:135,154) */
int marl_vec_add(const float* a, const float* b, float* out, long n, void* stream);
/* out[r,d] = sum_n in[r,n,d]   (VDNMixer, mixer.py:15-16 with D=1; QTRAN .sum(dim=-2), :384,:414); ld_in / ld_out =
 * row strides (>= D) of the (rows*N, D) input and the (rows, D) output */
int marl_agent_sum(const float* in, long ld_in, float* out, long ld_out, long rows, int N, int D, void* stream);
/* out[r,n,d] = in[r,d] (+ out if accumulate): autograd of the sum above */
int marl_agent_bcast(const float* in, long ld_in, float* out, long ld_out, long rows, int N, int D, int accumulate,
                     void* stream);

/* QMixMixer.forward after the hypernet layers (mixer.py:64-80).  hy row = [w1raw (N*E, agent-major)
 * | b1 (E) | w2raw (E) | relu(hyper_b2.0) (E)].  b2 (rows) = hyper_b2.2 output, or NULL: then it is formed in the kernel from the
 * fourth block, b2 = w22 . hb + b22 (w22 (E), b22 (1): hyper_b2.2's weight and bias, mixer.py:46-47). */
int marl_qmix_mix_fwd(const float* hy, long ldh, const float* b2, const float* w22, const float* b22, const float* q,
                      float* q_tot, long rows, int N, int E, void* stream);
/* its autograd: fills dhy[w1raw|b1|w2raw], db2 (= dq_tot), dq; the 4th block of dhy is written by the caller, or here when w22 is
 * given: d hb = dq_tot w22 (hb > 0) - the backward of hyper_b2.2 and its relu */
int marl_qmix_mix_bwd(const float* hy, long ldh, const float* q, const float* dq_tot, const float* w22, float* dhy,
                      float* db2, float* dq, long rows, int N, int E, void* stream);
/* The state-conditioned bias layers of QMixMixer in one pass over s, written into hy (rows, ldhy):
 * hy[:, c_b1 : c_b1 + 32] = hyper_b1(s), hy[:, c_h : c_h + 32] = relu(hyper_b2.0(s)) (mixer.py:44-47; E = 32).  Wb1, Wh: (32, S)
 * weights with the same row stride; s as in marl_qtran_state_parts.  With marl_qmix_mix_fwd(b2 = NULL) the generic mixer path
 * (two_hyper_layers, mixer.py:36-43) launches no marl_linear. */
int marl_qmix_tail_fwd(const marl_src_t* s, long rows, int S, const float* Wb1, long ldb1, const float* bb1, const float* Wh,
                       long ldwh, const float* bh, float* hy, long ldhy, int c_b1, int c_h, void* stream);

/* Fused QMIX (qmix_fused.hip): hypernet GEMMs + mixing in one kernel; the backward recomputes the
 * hypernet tile and accumulates the hypernet weight gradients in registers.  Supported when
 * marl_qmix_fused_supported(N, S, E) (E == 32, N*E+3E <= 256, S <= 128); otherwise compose
 * marl_linear + marl_qmix_mix_*.  `s` must be a dense-segment-0 source (row remap allowed). */
typedef struct {
  const float *w1, *w1_b;       /* hyper_w1   (N*E,S), (N*E) */
  const float *b1, *b1_b;       /* hyper_b1   (E,S), (E)     */
  const float *w2, *w2_b;       /* hyper_w2   (E,S), (E)     */
  const float *h, *h_b;         /* hyper_b2.0 (E,S), (E)     */
  const float *b2_w, *b2_b;     /* hyper_b2.2 (1,E), (1)     */
} marl_qmix_weights_t;
int marl_qmix_fused_supported(int N, int S, int E);
size_t marl_qmix_fused_workspace(long rows, int N, int S);
int marl_qmix_fused_fwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, float* q_tot,
                        long rows, int N, int S, int E, void* stream);
/* grads: same struct, pointing at the gradient tensors (accumulated into) */
int marl_qmix_fused_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* dq_tot,
                        float* dq, const marl_qmix_weights_t* grads, float* ws, size_t ws_bytes, long rows,
                        int N, int S, int E, void* stream);
/* The same backward with the TD loss of q_learner.py:112-127 folded in (the backward pass recomputes q_tot anyway, so the eval
 * mixer's forward launch, marl_td_loss and its reduction are not needed): per row target = r + gamma q_tot_tgt (1 - term),
 * td = mask (target - q_tot), dL/dq_tot = -2 mask td with mask = 1 - padded.  loss2[0] += sum td^2, loss2[1] += sum mask
 * (un-normalised; fixed summation order); q_tot (rows) is written when not NULL. */
int marl_qmix_fused_loss_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* q_tot_tgt,
                             const float* r, const float* term, const float* padded, float gamma, float* q_tot,
                             float* dq, const marl_qmix_weights_t* grads, float* loss2, float* ws, size_t ws_bytes,
                             long rows, int N, int S, int E, void* stream);
/* The same three entry points with the two GEMMs of a tile - hypernet output (network/mixer.py:60-77) and the hypernet weight
 * gradient - as bf16x6 split products (args.gemm_mode = "bf16x6"; see the agent_x6 / mlp3_x6 blocks for the arithmetic: every fp32
 * operand split exactly into three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulate).  Same arguments, same
 * workspace, same supported shapes, same epilogue arithmetic; results differ from the fp32-MFMA entry points by summation order
 * and the dropped <= 2^-24 terms only. */
int marl_qmix_fused_fwd_x6(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, float* q_tot,
                           long rows, int N, int S, int E, void* stream);
int marl_qmix_fused_bwd_x6(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* dq_tot,
                           float* dq, const marl_qmix_weights_t* grads, float* ws, size_t ws_bytes, long rows,
                           int N, int S, int E, void* stream);
int marl_qmix_fused_loss_bwd_x6(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* q_tot_tgt,
                                const float* r, const float* term, const float* padded, float gamma, float* q_tot,
                                float* dq, const marl_qmix_weights_t* grads, float* loss2, float* ws, size_t ws_bytes,
                                long rows, int N, int S, int E, void* stream);

/* Fused QMIX for WIDE states (qmix_wide.hip; MMM2: S = 322, N = 10 -> a 416 x 322 concatenated hypernet that does not fit
 * the registers-resident design above): the weights are packed per call into MFMA-fragment order (L2 resident) and
 * streamed against 64-row state tiles in LDS; same arithmetic, same gradient destinations.  `s`: dense segment 0 whose
 * rows start on 16-byte boundaries and hold S rounded up to 4 readable floats (EpisodeRecord pads the state row stride).
 * flags & 1: bf16 operands for the hypernet GEMM (v_mfma_f32_16x16x32_bf16, fp32 accumulate; BASELINE config 5 "bf16
 * mixer with MFMA") - the forward kernel is then bound by reading the states from HBM; mixing arithmetic and gradients stay
 * fp32.  flags & 2 (backward entry points, only together with flags & 1): the weight-gradient GEMM dW += dhy^T s also takes
 * bf16 operands (dhy - a gradient - and the states rounded to bf16 where a 32-row chunk is staged; fp32 accumulate; the bias
 * gradient stays an exact fp32 column sum): the four hypernet weight gradients then carry ~2e-3 relative error; without the
 * bit that GEMM is fp32 (v_mfma_f32_16x16x4_f32).  Workspace: packed weights (+ for backward: d(hypernet output) rows x (N*E+3E)
 * and the slabs).  Supported when marl_qmix_wide_supported(N, S, E) (E == 32, N <= 10, S <= 384).
 * CONTRACT on the row padding: the kernels read the columns S .. 4 ceil(S / 4) - 1 of every state row (the 16-byte loads of the
 * last chunk) and multiply them by packed weights that are exactly zero, so these pad columns must hold FINITE values (0 * NaN
 * would poison q_tot); EpisodeRecord zero-initialises them, a caller with its own buffers must do the same.
 * marl_qmix_wide_fwd_kernel(): the name prefix (rocprofv3 kernel trace) of the forward kernel a call with this shape launches -
 * "qmix_wide_kernel<false" (streaming), "qmix_wide_res_fwd_kernel" (bf16, weights resident in LDS: preceded by the pack kernel and
 * a memset of q_tot) or "qmix_wide_res32_fwd_kernel". */
int marl_qmix_wide_supported(int N, int S, int E);
const char* marl_qmix_wide_fwd_kernel(long rows, int N, int S, int flags);
size_t marl_qmix_wide_workspace(long rows, int N, int S, int backward);
int marl_qmix_wide_fwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, float* q_tot, float* ws,
                       size_t ws_bytes, long rows, int N, int S, int E, int flags, void* stream);
int marl_qmix_wide_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* dq_tot, float* dq,
                       const marl_qmix_weights_t* grads, float* ws, size_t ws_bytes, long rows, int N, int S, int E,
                       int flags, void* stream);
/* marl_qmix_wide_bwd with the TD loss folded in (see marl_qmix_fused_loss_bwd): q_tot of a row is complete inside one wave of
 * the backward kernel, so dL/dq_tot is formed there; loss2[0] += sum (mask td)^2, loss2[1] += sum mask; q_tot optional. */
int marl_qmix_wide_loss_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* q_tot_tgt,
                            const float* r, const float* term, const float* padded, float gamma, float* q_tot, float* dq,
                            const marl_qmix_weights_t* grads, float* loss2, float* ws, size_t ws_bytes, long rows,
                            int N, int S, int E, int flags, void* stream);

/* ---- fused three-layer heads (mlp3_fused.hip) -----------------------------------------------
 * Y[:, g*gs_y + 0..N3) = W3_g relu(W2_g relu(W1_g x + b1_g) + b2_g) + b3_g for `groups` equally shaped heads
 * with hidden width 64 whose parameters sit at constant element strides: the key / agents / action extractors
 * of DMAQ_SI_Weight (network/mixer.py:117-145, evaluated at :155-169 - 10 heads x 3 families per mixer call).
 * One kernel per family: the 64-wide hidden activations never leave the CU (composed from marl_linear they
 * cross HBM four times per update).  x: [dense0 | dense1 | one-hot blocks], shared by all heads; no gate / id
 * block.  The backward recomputes the hidden activations, keeps the weight gradients of a stripe of rows in
 * registers and accumulates them (fixed-order slab reduction) into `grads` (same struct, gradient tensors);
 * inputs get no gradient (states / actions).  Use when marl_mlp3_supported(); otherwise compose marl_linear.
 * w2 == NULL (H2 = 0 in marl_mlp3_supported): two-layer heads y = W3 relu(W1 x + b1) + b3 - the transformation nets
 * hyper_w_final / V of DMAQer (network/mixer.py:200-206), evaluated as one launch with groups = 2. */
typedef struct {
  const float *w1, *b1;         /* (64,K1), (64) of head 0 */
  const float *w2, *b2;         /* (64,64), (64)           */
  const float *w3, *b3;         /* (N3,64), (N3)           */
  long gs_w1, gs_b1, gs_w2, gs_b2, gs_w3, gs_b3;   /* element strides between consecutive heads */
} marl_mlp3_weights_t;
int marl_mlp3_supported(const marl_src_t* x, int K1, int H1, int H2, int N3, int groups);
int marl_mlp3_fwd(const marl_mlp3_weights_t* w, const marl_src_t* x, float* Y, long ldy, long gs_y,
                  long M, int K1, int N3, int groups, void* stream);
size_t marl_mlp3_bwd_workspace(long M, int K1, int N3, int groups);
int marl_mlp3_bwd(const marl_mlp3_weights_t* w, const marl_src_t* x, const float* dY, long lddy, long gs_dy,
                  const marl_mlp3_weights_t* grads, float* ws, size_t ws_bytes, long M, int K1, int N3,
                  int groups, void* stream);
/* The same pair with the hidden activations KEPT between forward and backward instead of recomputed (what autograd
 * does for these heads in the reference, network/mixer.py:149-171): the forward writes relu(h1) [and relu(h2)] of
 * every 16-row tile as the MFMA fragments the backward wants (`hsave`: marl_mlp3_save_floats(M, three, groups)
 * floats, layout private to the pair), the backward reads them back with 16-byte coalesced loads and skips layers
 * 1-2.  Same values as the recomputing pair (outputs bit-identical; the gradients are sums over another number of row
 * stripes).  hsave == NULL: exactly marl_mlp3_fwd / marl_mlp3_bwd. */
size_t marl_mlp3_save_floats(long M, int three, int groups);
/* Wide outputs: two-layer heads (w2 == NULL) take 16 < N3 <= 160 outputs, N3 a multiple of 4 (hyper_w1 / hyper_w2 of QMixMixer
 * with two_hyper_layers, network/mixer.py:36-43: state -> 64 -> n_agents * embed); a wider head is evaluated as `groups` column
 * blocks that share layer 1 (gs_w1 = gs_b1 = 0: those gradients are then summed over the groups).
 * 1 when the backward of this shape exists only as marl_mlp3_bwd_saved with hsave != NULL (N3 > 16, or K1 > 192, e.g. the heads of
 * DMAQ_SI_Weight on MMM2: state 322 [+ 10 x 18 one-hot actions], network/mixer.py:117-145): W1 no longer fits in LDS beside the
 * operand exchange, so layer 1 is not recomputed. */
int marl_mlp3_needs_kept(const marl_src_t* x, int K1, int N3);
int marl_mlp3_fwd_save(const marl_mlp3_weights_t* w, const marl_src_t* x, float* Y, long ldy, long gs_y,
                       float* hsave, size_t hsave_floats, long M, int K1, int N3, int groups, void* stream);
int marl_mlp3_bwd_saved(const marl_mlp3_weights_t* w, const marl_src_t* x, const float* dY, long lddy, long gs_dy,
                        const marl_mlp3_weights_t* grads, float* ws, size_t ws_bytes, const float* hsave,
                        size_t hsave_floats, long M, int K1, int N3, int groups, void* stream);

/* ---- agent unroll, forward, with the multiplies as an fp32-accurate split on the bf16 matrix cores (agent_x6.hip) ----------------
 * Opt-in (args.gemm_mode = "bf16x6"): the same T-step unroll as marl_agent_unroll_fwd (controller/share_params.py:147-168;
 * network/q_network.py:16-21) with every fp32 product as six bf16 MFMA products (see the mlp3_x6 block below for the arithmetic).
 * Same arguments, same meaning: `saved` (the six activation planes per row-step the fp32 BPTT kernel marl_agent_unroll_bwd reads,
 * same tile layout), `gi_out` / `gi_in` (the input-side gate sums one unroll stores and the double-Q continuation reads; these
 * hold PLAIN sums here - a pair of launches that shares them must both be this entry), `cu_budget` (two row tiles per workgroup
 * once there are more tiles than that many CUs).  marl_agent_unroll_x6_supported(), exactly: H = 64; 8 <= O <= 192 and O a multiple
 * of 8 (a row tile's observations fit three 16-byte prefetch registers per thread of one team); input width
 * I = O + (last_action ? A : 0) + (reuse_network ? N : 0) <= 224; 1 <= A <= 16, or <= 32 when I > 160 (two action tiles only in the
 * widest instantiation); T >= 4; B T N 256 < 2^32 (32-bit offsets): 2s3z-, 3s5z- and MMM2-sized agents; beyond 96 input columns one row
 * tile per workgroup.  The entry point also wants obs on a 16-byte boundary.  The caller uses marl_agent_unroll_fwd otherwise. */
int marl_agent_unroll_x6_supported(int B, int T, int N, int O, int A, int last_action, int reuse_network);
/* 1 when a NON-SAVING launch of marl_agent_unroll_fwd_x6 on this batch (saved = gi_in = hs = NULL: the target network's unroll and the
 * double-Q continuation of reference q_learner.py:104-110) runs on the round-6 decomposition (csrc/agent_x6p.hip: the recurrent team
 * multiplies x W_ih and h W_hh in one chain, five row tiles per workgroup, two barriers per step): whole-chip launches (cu_budget 0 /
 * 256) of more than 512 row tiles of 2s3z-sized agents (<= 96 input columns, <= 16 actions).  The learner then keeps no input-side
 * gate sums (gi_out / gi_in) for that batch. */
int marl_agent_unroll_x6_plain_r6(int B, int T, int N, int O, int A, int last_action, int reuse_network, int cu_budget);
int marl_agent_unroll_fwd_x6(const marl_agent_weights_t* w, const float* obs, long obs_bs, int obs_t0,
                             const int* ufed, long u_bs, int u_t0, const int* ep_len, const int* ep_map,
                             const float* h0, float* q, float* hs, float* h_last, float* saved, int B, int T, int N,
                             int O, int A, int last_action, int reuse_network, int cu_budget, float* gi_out,
                             const float* gi_in, void* stream);

/* ---- BPTT of the unroll on the same split arithmetic (agent_bwd_x6.hip) -------------------------------------------------------
 * Opt-in (args.gemm_mode = "bf16x6"): marl_agent_unroll_bwd with every fp32 product - the delta pass and the weight-gradient
 * reductions over rows - as six bf16 MFMA products; bias gradients are exact fp32 sums.  Same argument meaning (no dense dq and no
 * `hs`: the Q-learning losses reach q through one or two (column, value) pairs per row); `saved` is what either forward entry
 * stored.  marl_agent_unroll_bwd_x6_supported(): H = 64, A <= 32, sparse dq, T >= 3; the caller uses marl_agent_unroll_bwd
 * otherwise.  Workspace: marl_agent_bwd_x6_workspace() bytes (one slab per 32 rows). */
int marl_agent_unroll_bwd_x6_supported(int B, int T, int N, int A, int sparse_dq);
size_t marl_agent_bwd_x6_workspace(int B, int N, int A);
int marl_agent_unroll_bwd_x6(const marl_agent_weights_t* w, const int* dq_idx, const float* dq_val, const int* dq_idx2,
                             const float* dq_val2, int dq_gdiv, const float* dhs, const float* saved, float* dxp, float* dh0,
                             const marl_agent_grads_t* g, float* ws, size_t ws_bytes, int B, int T, int N, int A, void* stream);

/* ---- the same heads with the multiplies as an fp32-accurate SPLIT on the bf16 matrix cores (mlp3_x6.hip) -------------
 * Opt-in (args.gemm_mode = "bf16x6"; the default is the pair above on v_mfma_f32_16x16x4_f32).  Every fp32 operand is split
 * exactly into three bf16 terms (hi + mid + lo) and a product is the six bf16 products hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid
 * accumulated in fp32 by v_mfma_f32_16x16x32_bf16: the error against fp64 is that of the fp32 MFMA path (dropped terms <= 2^-24 of a
 * product), at 0.4 of its matrix-pipe time.  Same math as marl_mlp3_fwd_save / marl_mlp3_bwd_saved (network/mixer.py:117-145,
 * :149-171) for THREE-layer heads (w2 != NULL, H1 = H2 = 64) with N3 <= 16 outputs that marl_mlp3_supported() covers, whose virtual input
 * width KV = K1 + (the kernels' own padding of a dense segment 0 that is not a multiple of 4) satisfies KV < 16 KC with
 * KC = 4 ceil(KV / 64) <= 12: at most 191 columns, and the last 64-column block must leave a free column (KV = 112 is accepted, 64 /
 * 128 / 192 are not) - the first free column carries the ones that produce the layer-1 bias gradient.  `hsave`: marl_mlp3_save_floats(M, 1, groups) floats in the
 * layout of the fp32 pair (the kept fp32 activations; the backward splits them).  hsave == NULL in the forward: nothing is kept
 * (target mixer).  Workspace: marl_mlp3_bwd_workspace(). */
int marl_mlp3_x6_supported(const marl_src_t* x, int K1, int H1, int H2, int N3, int groups);
int marl_mlp3_x6_fwd_save(const marl_mlp3_weights_t* w, const marl_src_t* x, float* Y, long ldy, long gs_y,
                          float* hsave, size_t hsave_floats, long M, int K1, int N3, int groups, void* stream);
int marl_mlp3_x6_bwd_saved(const marl_mlp3_weights_t* w, const marl_src_t* x, const float* dY, long lddy, long gs_dy,
                           const marl_mlp3_weights_t* grads, float* ws, size_t ws_bytes, const float* hsave,
                           size_t hsave_floats, long M, int K1, int N3, int groups, void* stream);

/* ---- fused QTRAN-base heads (qtran_fused.hip) ------------------------------------------------
 * QtranQBase.forward (network/mixer.py:378-388, A = n_actions > 0, AE = 64 + A) and QtranV.forward (:411-418, A = 0,
 * AE = 64):  out[bt] = q( [s | sum_n enc([h_n | onehot(u_n)])] ).  The per-agent encoder activations never leave the
 * CU; the second encoder layer is applied AFTER the agent sum (it is linear): esum = W_enc2 (sum_n e1_n) + N b_enc2.
 *   hidden (BT*N, 64) agent rows (bt, n) at bt*N + n; u (BT*N) int32 action index (< 0: all-zero one-hot) or NULL
 *   when A = 0; sp (BT, 64) = W_q0[:, :S] s + b_q0 - the state part of the head's first layer, one marl_linear call,
 *   shared by evaluations of the same network on the same states (taken / greedy actions, qtran_learner.py:116,133).
 *   s1, e2 (BT, AEP), y1, y2 (BT, 64), AEP = AE rounded up to 16 (pad columns are written as zeros): saved
 *   activations for the backward pass (sum_n relu(e1_n), esum, the two hidden layers), all four or none (NULL).
 * Backward: given d_out (BT) writes the row-level gradients dy2, dy1 (BT, 64), de2 (BT, AEP) - the callers feed them
 * to marl_linear_wgrad for W_q4, W_q2, W_q0 and W_enc2 (reductions over BT rows) - and the gradient on hidden
 * (dhidden = or +=), and ACCUMULATES the gradients the agent-level pass owns: d_enc0_w (AE, AE), d_enc0_b (AE) and
 * d_enc2_b (AE) (= N * colsum(de2)).  Slabs + fixed-order reduce: bitwise reproducible.
 * Use when marl_qtran_supported(N, A, AE) (A <= 16, hidden widths 64); otherwise compose marl_linear. */
typedef struct {
  const float *enc0_w, *enc0_b;   /* hidden(_action)_encoding.0  (AE, AE), (AE) */
  const float *enc2_w, *enc2_b;   /* hidden(_action)_encoding.2  (AE, AE), (AE) */
  const float* q0_w; long q0_ld; int q0_s;   /* q.0 / v.0 weight (64, S + AE), its row stride, S */
  const float *q2_w, *q2_b;       /* q.2 / v.2  (64, 64), (64) */
  const float *q4_w, *q4_b;       /* q.4 / v.4  (1, 64), (1)   */
} marl_qtran_weights_t;
int marl_qtran_supported(int N, int A, int AE);
int marl_qtran_head_fwd(const marl_qtran_weights_t* w, const float* hidden, const int* u, const float* sp, float* out,
                        float* s1, float* e2, float* y1, float* y2, long BT, int N, int A, int AE, void* stream);
/* The joint-Q head on the same states and hidden states for two action sets in one launch: out <- u (activations saved as in
 * marl_qtran_head_fwd), out2 <- u2 (nothing saved): the taken and the greedy actions of the eval mixer, qtran_learner.py:116
 * and :133.  The encoder's first-layer product is computed once.  A > 0 only. */
int marl_qtran_head_fwd2(const marl_qtran_weights_t* w, const float* hidden, const int* u, const int* u2, const float* sp,
                         float* out, float* out2, float* s1, float* e2, float* y1, float* y2, long BT, int N, int A, int AE,
                         void* stream);
size_t marl_qtran_bwd_workspace(long BT, int AE);
int marl_qtran_head_bwd(const marl_qtran_weights_t* w, const float* hidden, const int* u, const float* d_out,
                        const float* y1, const float* y2, float* dy1, float* dy2, float* de2, float* dhidden,
                        int accumulate, float* d_enc0_w, float* d_enc0_b, float* d_enc2_b, float* ws, size_t ws_bytes,
                        long BT, int N, int A, int AE, void* stream);

/* State parts of the heads' first layers: sp_k (BT, 64) = W_k[:, :S] s + b_k for k < nsets (1 or 2) - the `sp` argument of
 * marl_qtran_head_fwd.  The joint-Q head and the V head of one update read the same states (qtran_learner.py:116-118):
 * nsets = 2 serves both from one pass over s.  W_k: q.0 / v.0 weight (64, ldw_k), ldw_k >= S; s: BT rows of S columns as
 * one dense segment (p0, ld0 % 4 == 0, 16-byte aligned; the row remap and the episode map of marl_src_t are honoured -
 * the learner reads s and s_next in place from the (T+1)-slot storage; S % 4 != 0 reads the row's last 16 bytes, whose pad
 * must be finite).  Use when marl_qtran_state_parts_supported(S) (S <= 384); otherwise marl_linear. */
int marl_qtran_state_parts_supported(int S);
int marl_qtran_state_parts(const marl_src_t* s, long BT, int S, int nsets, const float* W0, long ldw0, const float* b0, float* sp0,
                           const float* W1, long ldw1, const float* b1, float* sp1, void* stream);
/* Every weight gradient of one head that is a reduction over the BT rows, in one pass (network/mixer.py:381,386-387 /
 * :412,416-417; replaces five marl_linear_wgrad calls on tensors marl_qtran_head_fwd / _bwd have written):
 *   d_q0_w (64, ld_q0) += dy1^T [s | e2[:, :AE]]   d_q0_b += colsum(dy1)   d_q2_w (64,64) += dy2^T y1   d_q2_b += colsum(dy2)
 *   d_q4_w (64) += d_out^T y2   d_q4_b (1) += sum(d_out)   d_enc2_w (AE, AE) += de2[:, :AE]^T s1[:, :AE]
 * s as in marl_qtran_state_parts; s1, e2, de2 (BT, AEP); y1, y2, dy1, dy2 (BT, 64); d_out (BT).  ws: marl_qtran_wgrad_rows_workspace(S, AE)
 * bytes.  Slabs + fixed-order reduce: bitwise reproducible.  Use when marl_qtran_wgrad_rows_supported(S, AE)
 * (S % 4 == 0, S <= 384, AE = 64 + A with A <= 16). */
int marl_qtran_wgrad_rows_supported(int S, int AE);
size_t marl_qtran_wgrad_rows_workspace(int S, int AE);
int marl_qtran_wgrad_rows(const marl_src_t* s, const float* s1, const float* e2, const float* y1, const float* y2,
                          const float* d_out, const float* dy1, const float* dy2, const float* de2, float* d_q0_w, long ld_q0,
                          float* d_q0_b, float* d_q2_w, float* d_q2_b, float* d_q4_w, float* d_q4_b, float* d_enc2_w, float* ws,
                          size_t ws_bytes, long BT, int S, int AE, void* stream);

/* QPLEX (DMAQer.forward + calc_v/calc_adv, mixer.py:211-288; DMAQ_SI_Weight tail :158-169).
 *  wv row = [w_raw (N) | v (N)] (outputs of hyper_w_final.2 / V.2);
 *  heads = key (rows,K,1) | agents (rows,K,N) | action (rows,K,N) raw extractor outputs.
 *  out: v_tot, a_tot (either may be NULL).  max_q NULL => is_v only. */
int marl_qplex_mix_fwd(const float* w_raw, const float* v, const float* q, const float* max_q,
                       const float* key, const float* ag, const float* ac, float* v_tot, float* a_tot,
                       float* lam_out, long rows, int N, int K, int weighted_head, int minus_one,
                       void* stream);
/* autograd for q_tot = v_tot + a_tot given g = dL/dq_tot: dq (N), dw_raw, dv, dkey, dag, dac */
int marl_qplex_mix_bwd(const float* w_raw, const float* q, const float* max_q, const float* key,
                       const float* ag, const float* ac, const float* g, float* dq, float* dw_raw,
                       float* dv, float* dkey, float* dag, float* dac, long rows, int N, int K,
                       int weighted_head, int minus_one, void* stream);

/* QLearner.get_max_episode_len (q_learner.py:49-66; qtran_learner.py:52-69): out[0] = max over episodes of
 * (first step t with terminated[e,t] == 1) + 1, 0 when no episode terminates (the caller then uses
 * episode_limit); episodes that never terminate are ignored (quirk Q2).  term: (E, >=T) fp32, row stride ld. */
int marl_first_terminated_len(const float* term, long ld, int E, int T, int* out, void* stream);

/* ReplayBuffer.sample (common/replaybuffer.py:54-60) for a ring that lives in HBM: the per-step arrays of the B sampled
 * episodes idx[b] (int64) in one launch.  Sources are the ring's arrays: u (E,T,N) int32, r / terminated / padded (E,T)
 * fp32, length / won (E) int32, avail (E,T+1,N,A) fp32 ((T+1)-slot storage).  Outputs: o_map (B) = idx as int32 (the
 * episode map the unroll / mixer kernels read obs and state through), u and u_act = max(u, 0) (B,T,N), r / term / padded
 * (B,T), length / won (B), avail_next (B,T,N,A) = avail slots 1..T; avail_cur (B,T,N,A) or NULL = avail slots 0..T-1 with
 * zeros from each episode's end on - the availability QPLEX / QTRAN mask the current-step greedy action with
 * (q_learner.py:135-140, qtran_learner.py:103-108).  obs and state are NOT copied. */
int marl_replay_gather(const long long* idx, int B, int T, int N, int A, const int* u_src, const float* r_src,
                       const float* term_src, const float* padded_src, const int* length_src, const int* won_src,
                       const float* avail_src, int* o_map, int* u, int* u_act, float* r, float* term, float* padded,
                       int* length, int* won, float* avail_next, float* avail_cur, void* stream);

/* TD target + masked squared error (q_learner.py:165-168).  Writes the UN-normalised gradient
 * dq_tot = -2 mask^2 td and out2 = {sum (mask td)^2, sum mask}; the 1/sum(mask) factor is applied
 * in the optimizer so that data-parallel ranks can all-reduce numerators. */
int marl_td_loss(const float* q_tot, const float* q_tot_tgt, const float* r, const float* term,
                 const float* padded, float gamma, float* dq_tot, float* out2, float* ws, long rows,
                 void* stream);
/* QTRAN-base losses (qtran_learner.py:121-152). out4 = {l_td, l_opt, l_nopt numerators, sum mask};
 * gradients un-normalised: d_jq (joint_q_evals), d_v, d_qsum_opt, d_qsum_nopt. */
int marl_qtran_loss(const float* jq, const float* jq_tgt, const float* v, const float* jq_hat,
                    const float* qsum_opt, const float* qsum_nopt, const float* r, const float* term,
                    const float* padded, float gamma, float lam_opt, float lam_nopt, float* d_jq,
                    float* d_v, float* d_qsum_opt, float* d_qsum_nopt, float* out4, float* ws, long rows,
                    void* stream);
size_t marl_loss_workspace(long rows);

/* ---- optimizer (optim.hip): clip_grad_norm_ + RMSprop / Adam on ONE flat buffer -------------
 * (q_learner.py:42-47,170-173; torch defaults).  g is the un-normalised gradient; den points to
 * sum(mask) on the device (NULL = 1).  sumsq[0] receives sum g^2 (before scaling). */
int marl_grad_sumsq(const float* g, long n, float* sumsq, float* ws, void* stream);
size_t marl_sumsq_workspace(long n);
int marl_rmsprop_step(float* p, const float* g, float* sq, long n, float lr, float alpha, float eps,
                      float clip, const float* sumsq, const float* den, void* stream);
int marl_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                   float beta2, float eps, float bc1, float bc2_sqrt, float clip, const float* sumsq,
                   const float* den, void* stream);

/* ---- rollout (rollout.hip) ------------------------------------------------------------------
 * epsilon-greedy action choice of SharedMAC.choose_action (share_params.py:66-70), batched:
 * explore iff u01(hash(rseed,EXPLORE,env,tg,n)) < eps, then the floor(u01(hash(..PICK..))*n_avail)-th
 * available action, else first-index argmax of avail-masked q.  act_out[e*act_es + n]; envs with
 * alive[e]==0 get -1. tg[e] = per-env global step id (NULL: tg0). */
int marl_select_actions(const float* q, const float* avail, long avail_es, const int* alive, float eps,
                        unsigned rseed, int env0, const int* tg, int tg0, int* act_out, long act_es,
                        int E, int N, int A, void* stream);
/* Synthetic SMAC-shaped environment (stands in for StarCraft II, main.py:16-20).
 * Episode storage is (T+1)-slot: obs (E,T+1,N,O), state (E,T+1,state_ld >= S), avail (E,T+1,N,A).  state_ld is the
 * row stride of the state storage in floats: a multiple of 4 keeps every state row 16-byte aligned for any S (MMM2:
 * S = 322 -> 324), which the GEMM kernels downstream need for their vector loads; pad columns are written as zeros
 * or left untouched (allocate them zeroed). */
int marl_synth_lengths(unsigned seed, int env0, int episode, int* len, int* won, int E, int T, void* stream);
int marl_synth_observe(unsigned seed, int env0, int episode, int t, const int* len, float* obs,
                       float* state, long state_ld, float* avail, int E, int T, int N, int O, int S, int A, void* stream);
/* reward / terminated / padded for step t given act (E,N) (rollout.py:86-96,122-133 for padding);
 * u (E,T,N) int32 gets the action or -1 on padding. alive_out[e] = (t+1 < len[e]) */
int marl_synth_step(unsigned seed, int env0, int episode, int t, const int* len, const int* act,
                    int* u, float* r, float* term, float* padded, int* alive_next, int E, int T, int N, int A,
                    void* stream);

/* select + step + observe(t+1) of the synthetic env in ONE launch per lock-step (same arithmetic as the
 * three calls above; q is the (E,N,A) output of the T=1 agent unroll). */
int marl_synth_fused_step(unsigned seed, unsigned rseed, int env0, int episode, int t, float eps, const int* len,
                          const float* q, float* obs, float* state, long state_ld, float* avail, int* u, float* r,
                          float* term, float* padded, int E, int T, int N, int O, int S, int A, void* stream);

/* The WHOLE rollout of the synthetic env in one persistent launch (rollout_fused.hip): agent step,
 * epsilon-greedy choice, env step and next observation for all T lock-steps; weights and hidden state
 * stay on chip.  eps[T] = epsilon per lock-step; eps == NULL: eps(0) = eps0 and the reference's per-step anneal
 * eps(t+1) = eps(t) > eps_min ? eps(t) - eps_anneal : eps(t) (rollout.py:100-101) evaluated in the kernel, in fp64.  Produces the same (T+1)-slot record as the
 * launch-per-step path.  stats (optional, [3][E] floats): per episode  sum_t r | won | length - what
 * RolloutWorker.generate_episodes returns besides the batch (rollout.py:135-140), without extra launches.  Needs whole environments per workgroup: marl_synth_rollout_supported(). */
int marl_synth_rollout_supported(int N, int O, int A);
int marl_synth_rollout(const marl_agent_weights_t* w, unsigned seed, unsigned rseed, int env0, int episode,
                       int fixed_len, const float* eps, float* obs, float* state, long state_ld, float* avail, int* u,
                       float* r, float* term, float* padded, int* length, int* won, float* h_out,
                       float* stats, double eps0, double eps_anneal, double eps_min, int E, int T, int N, int O, int S,
                       int A, int last_action, int reuse_network, void* stream);

/* The same whole rollout with the agent step (network/q_network.py:16-21) as bf16x6 split products (rollout_x6.hip; args.gemm_mode =
 * "bf16x6"): same arguments, same environment, same epsilon-greedy choice, same record - the integer fields agree with
 * marl_synth_rollout wherever no two available actions' Q values lie within fp32 rounding of each other (the choice is an argmax).
 * fc1 is evaluated as (bias + W1[:, obs | id] in) + W1[:, O + last action]: the observation part on the matrix cores a step ahead,
 * the chosen action's column added in fp32.  Supported: H = 64, 1 <= A <= 16, O a multiple of 4, O + A + N <= 160, N <= 64. */
int marl_synth_rollout_x6_supported(int N, int O, int A);
/* How marl_synth_rollout_x6 runs a batch of E environments (two decompositions of the same arithmetic, picked by batch size:
 * csrc/rollout_x6_v1.hip holds at most three row tiles of 16 (episode, agent) rows per workgroup, csrc/rollout_x6.hip up to five):
 * plan[0] = decomposition (1 / 2), plan[1] = workgroups, plan[2] = row tiles per workgroup, plan[3] = environments per workgroup,
 * plan[4] = fc1 chunks of 32 input columns.  What bench.py's roofline model counts the executed products by. */
int marl_synth_rollout_x6_plan(int E, int N, int O, int A, int last_action, int reuse_network, int* plan);
int marl_synth_rollout_x6(const marl_agent_weights_t* w, unsigned seed, unsigned rseed, int env0, int episode,
                          int fixed_len, const float* eps, float* obs, float* state, long state_ld, float* avail, int* u,
                          float* r, float* term, float* padded, int* length, int* won, float* h_out,
                          float* stats, double eps0, double eps_anneal, double eps_min, int E, int T, int N, int O, int S,
                          int A, int last_action, int reuse_network, void* stream);

const char* marl_hip_version(void);

/* Experiment switches (A/B measurements, variant tests): one table per process; NO entry point reads the environment.
 * Names and defaults: "fwd_xs" 1 (the double-Q unroll reads the eval unroll's input-side gate sums - replaces the work
 * q_learner.py:110 repeats), "fwd_dma" 0 (LDS-DMA observation tile of the saving unroll), "fwd_w2l" 1 (six prefetch registers for
 * wide observations), "bwd_pipe_max_rt" 4 (row tiles per workgroup up to which the one-barrier BPTT runs), "wgrad_tall" 1 (LDS-staged
 * tall weight-gradient kernel), "wide_res" 1 / "wide_res32" 0 (resident-weights forward of the wide-state QMIX mixer, mixer.py:57-80).
 * Results do not depend on them beyond fp32 summation order.  set: 0, or -1 for an unknown name; get: the value, or INT_MIN.
 * marl_amd/experiments.py forwards the MARL_* environment variables of the same names once, at import. */
int marl_experiment_set(const char* name, int value);
int marl_experiment_get(const char* name);

#ifdef __cplusplus
}
#endif
#endif
