/* immtsf.h -- C ABI of libimmtsf_hip.so: the MI355X (gfx950) implementation of IMM-TSF's multimodal-fusion
 * forward/backward hot path.
 *
 * The reference (blacksnail789521/IMM-TSF) is pure Python/PyTorch and has NO native interface; the seams this
 * library sits behind are the nn.Module methods listed next to each entry point (file:line into the reference).
 * The Python host side in imm-tsf_amd/ (fusions/, layers/, models/, lib/) keeps those classes' constructor and
 * forward signatures and state_dict keys and calls the functions below through ctypes (imm-tsf_amd/immtsf/_lib.py);
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory unless said otherwise; fp32 row-major tensors.
 *  - the caller owns every buffer, including workspaces (size them with the *_workspace_bytes functions) and the
 *    saved-for-backward state, which lives inside the forward workspace: keep it alive until backward has run.
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*, NULL = default stream); nothing
 *    synchronises, allocates or frees, so the calls can be captured into a hipGraph.
 *  - return value: 0 = ok; <0 = IMMTSF_E*; >0 = hipError_t of a failed launch.  Nothing throws.
 *  - every compute entry point is re-entrant: concurrent calls from different host threads (each on its own stream) are
 *    safe as long as they do not share output / workspace buffers (tests/test_gpu_abi_threads.py).  The library keeps NO
 *    per-call state between calls; what it does keep process-wide is optional, empty / off by default, and listed here:
 *      (1) the bf16 twin registry (immtsf_bf16_twin_register ...): <= 16 address ranges behind a reader/writer lock;
 *      (2) the GEMM timing tap (immtsf_timing_enable ...): a mutex-guarded record buffer, used by bench.py only;
 *      (3) the A/B switches immtsf_debug_gemm_config / immtsf_debug_gemm2_config / immtsf_set_side_stream: plain
 *          process-wide integers for measurements -- change them only while no call is in flight;
 *      (4) when immtsf_set_side_stream(1): one side stream + two events per (host thread, device), created at first use.
 *  - precision: 0 = exact fp32 (v_mfma_f32_16x16x4_f32; parity mode), 1 = bf16 MFMA operands with fp32
 *    accumulation (v_mfma_f32_16x16x32_bf16; tensors in memory stay fp32).
 */
#ifndef IMMTSF_H
#define IMMTSF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMMTSF_ABI_VERSION 6
#define IMMTSF_T2V_FOLD_MIN_ROWS 8192 /* see immtsf_fusion_cfg.form */
#define IMMTSF_FORM_NO_PROJ 16        /* immtsf_fusion_cfg.form bit, TTF_T2V_XAttn: leave proj_out to the consumer (E_txt := Z, dE_txt := dZ,
                                         which the backward overwrites in place; proj_out's gradients are not written) */
#define IMMTSF_FORM_HALF_OUT 32       /* immtsf_fusion_cfg.form bit, TTF_T2V_XAttn forward with out_h in the bf16 dataflow: the caller promises
                                         that the output's only reader takes the bf16 image -- the fp32 E_txt (0.4 GB per step at 4096
                                         windows, with no reader) is NOT written.  (ABI 6) */
#define IMMTSF_FORM_LOWRANK_OUT 64    /* immtsf_fusion_cfg.form bit, immtsf_mmf_xrank_p_backward_data_z: dZ = dP Wc is NOT written -- the producer's
                                         backward takes the pair (dP, Wc) itself (immtsf_fusion_cfg.lr_grad, immtsf_mmf_xrank_lowrank_basis).
                                         (ABI 6) */

#define IMMTSF_OK 0
#define IMMTSF_EINVAL (-1)       /* bad dimension / null pointer */
#define IMMTSF_EWORKSPACE (-2)   /* workspace too small */
#define IMMTSF_EUNSUPPORTED (-3) /* shape outside what the kernels implement (documented per call) */

typedef void* immtsf_stream_t;

int immtsf_abi_version(void);
/* sizeof of every struct of this ABI, in the order immtsf_fusion_cfg, immtsf_t2v_params, immtsf_recavg_params, immtsf_xadd_params,
 * immtsf_gr_params, immtsf_ttcn_params, immtsf_gcn_params, immtsf_decoder_params, immtsf_time2vec_params, immtsf_encoder_layer_cfg,
 * immtsf_encoder_layer_params, immtsf_ffn_block_cfg, immtsf_ffn_block_params, immtsf_store, immtsf_note_index, immtsf_lowrank_grad: lets a
 * binding check its own struct definitions against the library it loaded (tests/test_abi.py compares with ctypes.sizeof).  Writes
 * min(max, 16) entries to the HOST array `out`, returns the number of structs (16).  (ABI 5; 16 structs from ABI 6) */
/* the history's normalisation of PatchTST / TimesNet (reference models/PatchTST.py:104-109, models/TimesNet.py:113-117: x - mean over time,
 * / sqrt(biased variance + 1e-5)) as one launch: x (B, L, C) -> xn (B, L, C), means (B, C), stdev (B, C).  Data only: no gradient.  (ABI 6) */
int immtsf_instance_norm(const float* x, int32_t B, int32_t L, int32_t C, float* xn, float* means, float* stdev, immtsf_stream_t stream);
/* measurement aid: the gather that IS in the timed step of a packed batch in bf16 mode -- X_h[r, :d_m] = bf16(emb[src_rows[r], :]), X_h[r, d_m:
 * d_m + dt] = bf16(Time2Vec(tau[rowmap[r]])) for r < *total (one wave per packed row; csrc/rowops.hip notes_stage_kernel, the first launch of
 * immtsf_ttf_t2v_xattn_forward_packed's folded and staged forms).  bench.py times it for `roofline_hbm`.  reference: the note gather of
 * fusions/TTF_T2V_XAttn.py:120-131 and time2vec :27-39.  (ABI 6) */
int immtsf_notes_stage(const float* emb, int32_t d_m, const int32_t* src_rows, const int32_t* total, int32_t max_rows, void* X_h, int32_t ldx,
                       const float* tau, const int32_t* rowmap, int32_t dt, const float* lin_w, const float* lin_b, const float* per_w,
                       const float* per_b, immtsf_stream_t stream);
#define IMMTSF_ABI_NSTRUCTS 16
int immtsf_abi_sizes(int32_t* out, int32_t max);

/* ------------------------------------------------------------------------------------------------------------
 * Shape/config of one fusion call.  B windows, N padded notes per window, T padded forecast steps, C channels,
 * d_m = LLM embedding width, d = d_txt, H = n_heads_fusion.  Dropout is active iff training != 0 && p_drop > 0;
 * masks come from Philox4x32-10 keyed by (seed, site, element index) so backward regenerates them.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct immtsf_fusion_cfg {
    int32_t B, N, T, C, d_m, d, H;
    int32_t precision;
    int32_t training;
    float p_drop;
    float kappa;
    uint64_t seed;
    const uint64_t* seed_step_dev; /* optional device counter added to `seed` when the kernels run (NULL = 0): a captured
                                      hipGraph bumps it once per replay so every step draws fresh dropout masks */
    int32_t grads_prezeroed;       /* backward only: every gradient buffer passed in `grads` is already zero (e.g. one
                                      memset of a flat gradient buffer per step), so split-K weight-gradient GEMMs skip
                                      their own zero-fill */
    int32_t form;                  /* TTF_T2V_XAttn: 0 = the library chooses by batch size (the FOLDED form -- csrc/t2v_fold.hip -- from
                                      IMMTSF_T2V_FOLD_MIN_ROWS padded note rows B*N on, where its limits hold: N <= 64 padded notes,
                                      T <= 32, d <= 1024, H <= 4; below that the fixed cost of its parameter-only chains outweighs the
                                      rows it saves), 1 = the reference's GEMM chain as written, 2 = the folded form wherever its limits
                                      hold, 3 = the MIX-FIRST variant of the folded form (csrc/t2v_premix.hip: the window's raw rows are
                                      mixed per forecast step, then ONE B*T-row product; bf16 mode, one head, T <= 32) wherever ITS limits
                                      hold -- what 0 chooses for windows of more than 64 padded notes.  Same function, same parameters,
                                      same dropout masks either way; forward and backward of one call pair must be given the same
                                      value.  (ABI 3; occupies former padding) */
    /* bf16 mode, optional (NULL = off): bf16 images of the activations that cross a block boundary, so that the consumer's
     * GEMMs read them by LDS-DMA without a cast kernel of their own.  out_h: the call also writes its main activation
     * output there (ttf forward: E_txt (B*T*d); mmf q backward: dKV (B*T*2d); mmf kv backward: dE_txt).  in_h: image of
     * the call's main activation input, produced that way (mmf kv forward: E_txt; mmf kv backward: dKV; ttf backward:
     * dE_txt).  aux_h: mmf kv backward only, the image of E_txt its forward was given as in_h.  Only honoured when the
     * call runs the bf16 dataflow (precision 1, d a multiple of 16). */
    const void* in_h;
    const void* aux_h;
    void* out_h;
    int32_t* sched_flag;           /* TTF_T2V_XAttn backward, optional (NULL = off): a device flag (immtsf_flag_*) the call sets to 1 behind the
                                      last launch of its row-bound part -- what follows are parameter chains of small launches.  A SCHEDULING
                                      HINT, not a dependency: a caller that runs chip-filling work nothing waits for on another stream (the
                                      patch encoder's backward) lets it spin on this flag, so that it shares the chip with the small launches
                                      instead of the row-bound ones.  (ABI 4) */
    int32_t bwd_phase;             /* TTF_T2V_XAttn backward in its chain form, optional (0 = the whole backward in one call, every weight
                                      gradient in ONE grouped launch at its end): a mask -- bits 0..2 = the DATA path of phases A, B, C
                                      (IMMTSF_BWD_PHASE_*), bits 4..6 = the PARAMETER gradients of phases A, B, C (IMMTSF_BWD_WGRAD_*: the
                                      weight-gradient products, which nothing in the backward waits for, as one grouped launch per call,
                                      and phase B's query path).  The caller runs the data phases in order A, B, C and a phase's parameter
                                      gradients in or behind the call that ran its data path -- on any stream ordered behind it (same cfg,
                                      workspace, scratch and grads in every call).  Uses: a data-parallel step hands finished gradient
                                      buckets to the all-reduce while the later phases still run; a step with a parameter-only branch
                                      moves the early weight gradients off the text side's dependent chain (immtsf.train.FlagStep).  The
                                      folded form does all its work in the call that holds IMMTSF_BWD_PHASE_C.  (ABI 5) */
    int32_t reserved0;             /* must be 0 */
    const struct immtsf_note_index* note_index; /* TTF_T2V_XAttn *_packed calls, optional (NULL = the call derives the index itself: two
                                      launches in front of everything else): the ragged index of the batch's notes, built ONCE per batch by
                                      immtsf_note_index_build -- by the batch builder, where the per-window note counts are known
                                      (immtsf.data / SURVEY 8f row 1: "offsets authoritative") -- instead of inside every forward.  HOST
                                      struct of device pointers; forward and backward of a call pair get the same one; M_txt may then be NULL
                                      (it is note_index->mtxt).  (ABI 5) */
    const struct immtsf_lowrank_grad* lr_grad; /* TTF_T2V_XAttn backward with IMMTSF_FORM_NO_PROJ, optional (NULL = dE_txt holds dZ): the upstream
                                      gradient in LOW-RANK form, dZ = coef basis -- what a rank-r projection behind the block sends back
                                      (MMF_XAttn_Add's composed projection).  The LayerNorm backward forms the product in registers; dE_txt
                                      is then not read (it must still be a valid pointer).  Only where immtsf_ttf_t2v_xattn_accepts_lowrank
                                      says so.  HOST struct of device pointers.  (ABI 6) */
} immtsf_fusion_cfg;
typedef struct immtsf_lowrank_grad {
    const float* coef;             /* (B*T, rank) row-major, pitch ld */
    const float* basis;            /* (rank, d) row-major, 16-byte aligned */
    int32_t rank;
    int32_t ld;
} immtsf_lowrank_grad;
/* the ragged index of a packed batch (what immtsf_ragged_index derives from a zero-padded tensor): mask u8 (B*N: n < lengths[b]), mtxt u8
 * (B: lengths[b] > 0), lengths i32 (B), offsets i32 (B+1; offsets[B] = total notes), rowmap i32 (B*N; packed row -> b*N+n), seg i32 (B*N;
 * packed row -> window).  Reference: the quantities fusions/TTF_T2V_XAttn.py:107,124,146 re-derive from the padded tensor every call. */
typedef struct immtsf_note_index {
    uint8_t* mask;
    uint8_t* mtxt;
    int32_t* lengths;
    int32_t* offsets;
    int32_t* rowmap;
    int32_t* seg;
} immtsf_note_index;
int immtsf_note_index_build(const int32_t* lengths_in, int32_t B, int32_t N, const immtsf_note_index* out, immtsf_stream_t stream);
/* after phase A: d proj_out (unless IMMTSF_FORM_NO_PROJ), d layer_norm, d attn.out_proj are final; after B: d attn.in_proj_{weight,bias}
 * and d Q_param; after C: d input_proj, d time2vec, d KV_proj */
#define IMMTSF_BWD_PHASE_A 1
#define IMMTSF_BWD_PHASE_B 2
#define IMMTSF_BWD_PHASE_C 4
#define IMMTSF_BWD_WGRAD_A 16
#define IMMTSF_BWD_WGRAD_B 32
#define IMMTSF_BWD_WGRAD_C 64

/* a2: ragged index of a zero-padded note tensor.  reference: note_mask = (V.abs().sum(2) > 0)
 * fusions/TTF_T2V_XAttn.py:107,124,146 ; fusions/TTF_RecAvg.py:69,110.
 * notes (B,N,d_m) -> note_mask u8 (B,N), lengths i32 (B), offsets i32 (B+1), rowmap i32 (B*N; first offsets[B]
 * valid, value b*N+n), seg i32 (B*N; window of each packed row), m_txt u8 (B).  nan_flag (device int32, may be
 * NULL) is OR-ed with 1 if any input element is NaN (the reference raises ValueError, TTF_*.py:116 / :75). */
int immtsf_ragged_index(const float* notes, int32_t B, int32_t N, int32_t d_m, uint8_t* note_mask, int32_t* lengths,
                        int32_t* offsets, int32_t* rowmap, int32_t* seg, uint8_t* m_txt, int32_t* nan_flag,
                        immtsf_stream_t stream);

/* ---- a3/a4/a8: TTF_T2V_XAttn.forward  (fusions/TTF_T2V_XAttn.py:93-184; Time2Vec :7-24; MHA semantics of
 * torch.nn.MultiheadAttention at :79-84,161-166).  Parameter pointers use the module's state_dict order. */
typedef struct immtsf_t2v_params {
    float* Q_param;                     /* (d)            Q_param (1,1,d) */
    float *input_proj_w, *input_proj_b; /* (d,d_m),(d)    NULL,NULL when d_txt=None (then d == d_m) */
    float *t2v_lin_w, *t2v_lin_b;       /* (1),(1)        time2vec.linear */
    float *t2v_per_w, *t2v_per_b;       /* (d/2-1),(d/2-1) time2vec.periodic */
    float *kv_w, *kv_b;                 /* (d, d+d/2),(d) KV_proj */
    float *attn_in_w, *attn_in_b;       /* (3d,d),(3d)    attn.in_proj_* */
    float *attn_out_w, *attn_out_b;     /* (d,d),(d)      attn.out_proj.* */
    float *ln_w, *ln_b;                 /* (d),(d)        layer_norm */
    float *proj_out_w, *proj_out_b;     /* (d,d),(d)      proj_out */
} immtsf_t2v_params;

size_t immtsf_ttf_t2v_xattn_workspace_bytes(const immtsf_fusion_cfg* cfg);
size_t immtsf_ttf_t2v_xattn_scratch_bytes(const immtsf_fusion_cfg* cfg);
/* 1 when calls with this cfg run the block in its FOLDED form (csrc/t2v_fold.hip: scores as a mat-vec of the raw notes with a folded
 * vector per head, ONE sum-of-notes x (H d) x (d_m + d/2) product in place of input_proj / KV_proj / in-projection / out_proj,
 * parameter gradients by the chain rule through the folded factors), 0 when they run the reference's GEMM chain as written
 * (cfg->form == 1, or a shape outside the folded form's limits).  reference: fusions/TTF_T2V_XAttn.py:120-182 either way. */
int immtsf_ttf_t2v_xattn_folded(const immtsf_fusion_cfg* cfg);
/* 1 when the backward of calls with this cfg (IMMTSF_FORM_NO_PROJ set) takes its upstream gradient in low-rank form of this rank
 * (immtsf_fusion_cfg.lr_grad), 0 otherwise: the one-pass LayerNorm backward with sums must apply (d % 4 == 0, 256 <= d <= 1024, B*T >= 512,
 * (rank + 3) d floats of LDS <= 150 KB, rank <= 64) and, in the folded form, x_hat must be the bf16 image.  (ABI 6) */
int immtsf_ttf_t2v_xattn_accepts_lowrank(const immtsf_fusion_cfg* cfg, int32_t rank);
/* notes (B,N,d_m), tau (B,N)  ->  E_txt (B,T,d), M_txt u8 (B).  t_hat's values do not enter this block (only T). */
int immtsf_ttf_t2v_xattn_forward(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* notes,
                                 const float* tau, float* E_txt, uint8_t* M_txt, void* workspace, size_t workspace_bytes,
                                 int32_t* nan_flag, immtsf_stream_t stream);
/* dE_txt (B,T,d) -> every parameter gradient (overwritten, not accumulated).  `workspace` is the forward's. */
int immtsf_ttf_t2v_xattn_backward(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* notes,
                                  const float* tau, const float* dE_txt, void* workspace, size_t workspace_bytes,
                                  void* scratch, size_t scratch_bytes, const immtsf_t2v_params* grads,
                                  immtsf_stream_t stream);
/* packed-input variants: the notes stay in the resident embedding matrix `emb` [*, d_m];
 * src_rows[offsets[b] + i] is the row of window b's i-th note and lengths[b] its note count, both as emitted by
 * immtsf_collate_notes.  tau stays (B,N) padded (N = cfg->N >= max lengths).  No padded embedding tensor is read
 * and the note mask is not re-derived from |V|.  Same workspace / scratch sizes as the padded entry points. */
int immtsf_ttf_t2v_xattn_forward_packed(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* emb,
                                        const int32_t* src_rows, const int32_t* lengths, const float* tau, float* E_txt,
                                        uint8_t* M_txt, void* workspace, size_t workspace_bytes, immtsf_stream_t stream);
int immtsf_ttf_t2v_xattn_backward_packed(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* emb,
                                         const int32_t* src_rows, const float* tau, const float* dE_txt, void* workspace,
                                         size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                         const immtsf_t2v_params* grads, immtsf_stream_t stream);

/* ---- a5: TTF_RecAvg.forward (fusions/TTF_RecAvg.py:54-112) */
typedef struct immtsf_recavg_params {
    float* log_recency_sigma;           /* ()  */
    float *input_proj_w, *input_proj_b; /* (d,d_m),(d) or NULL */
    float *ln_w, *ln_b;                 /* (d) */
    float *proj_w, *proj_b;             /* (d,d),(d) */
} immtsf_recavg_params;

size_t immtsf_ttf_recavg_workspace_bytes(const immtsf_fusion_cfg* cfg);
size_t immtsf_ttf_recavg_scratch_bytes(const immtsf_fusion_cfg* cfg);
/* t_hat (B,T) (the caller broadcasts a (T,) vector) */
int immtsf_ttf_recavg_forward(const immtsf_fusion_cfg* cfg, const immtsf_recavg_params* p, const float* notes,
                              const float* tau, const float* t_hat, float* E_txt, uint8_t* M_txt, void* workspace,
                              size_t workspace_bytes, int32_t* nan_flag, immtsf_stream_t stream);
int immtsf_ttf_recavg_backward(const immtsf_fusion_cfg* cfg, const immtsf_recavg_params* p, const float* notes,
                               const float* tau, const float* t_hat, const float* dE_txt, void* workspace,
                               size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                               const immtsf_recavg_params* grads, immtsf_stream_t stream);

/* ---- a6/a8: MMF_XAttn_Add.forward (fusions/MMF_XAttn_Add.py:56-103) */
typedef struct immtsf_xadd_params {
    float *proj_q_w, *proj_k_w, *proj_v_w; /* (d,C),(d,d),(d,d) no bias */
    float *attn_in_w, *attn_in_b;          /* (3d,d),(3d) */
    float *attn_out_w, *attn_out_b;        /* (d,d),(d) */
    float *res_w, *res_b;                  /* (C,d),(C)  residual_head */
    float *ln_w, *ln_b;                    /* (C) */
} immtsf_xadd_params;

size_t immtsf_mmf_xattn_add_workspace_bytes(const immtsf_fusion_cfg* cfg);
size_t immtsf_mmf_xattn_add_scratch_bytes(const immtsf_fusion_cfg* cfg);
/* Y_ts (B,T,C), E_txt (B,T,d), M_txt u8 (B) -> Y_out (B,T,C) */
int immtsf_mmf_xattn_add_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts,
                                 const float* E_txt, const uint8_t* M_txt, float* Y_out, void* workspace,
                                 size_t workspace_bytes, immtsf_stream_t stream);
/* dY_out -> dY_ts (B,T,C), dE_txt (B,T,d), parameter grads (all overwritten) */
int immtsf_mmf_xattn_add_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts,
                                  const float* E_txt, const uint8_t* M_txt, const float* dY_out, float* dY_ts,
                                  float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch,
                                  size_t scratch_bytes, const immtsf_xadd_params* grads, immtsf_stream_t stream);
/* The same block as two halves, for two-stream scheduling: the key/value half depends only on E_txt -- it yields
 * KV (B*T, 2d) = (k | v) side by side, i.e. the MHA in-projections of proj_k(E_txt) and proj_v(E_txt), computed with the
 * per-step product weights W_in,k W_k and W_in,v W_v (fusion_blocks_mmf.hip) -- the query half is everything else.  Run
 * kv_forward on the text stream beside the backbone; in backward the query half yields dY_ts (-> backbone) and dKV
 * (-> kv_backward -> dE_txt).  Each half writes only the parameter gradients it owns (kv: proj_k_w, proj_v_w, rows
 * d..3d of attn_in_w/_b; q: the rest).  The monolithic entry points above are exactly kv + q. */
size_t immtsf_mmf_xattn_kv_workspace_bytes(const immtsf_fusion_cfg* cfg);
size_t immtsf_mmf_xattn_kv_scratch_bytes(const immtsf_fusion_cfg* cfg);
size_t immtsf_mmf_xattn_q_workspace_bytes(const immtsf_fusion_cfg* cfg);
size_t immtsf_mmf_xattn_q_scratch_bytes(const immtsf_fusion_cfg* cfg);
int immtsf_mmf_xattn_kv_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, float* KV,
                                void* workspace, size_t workspace_bytes, immtsf_stream_t stream);
int immtsf_mmf_xattn_kv_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, const float* dKV,
                                 float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                 const immtsf_xadd_params* grads, immtsf_stream_t stream);
/* The query half's per-step product weights (W_in,q W_q | W_res W_out | W_res b_out + b_res | W_res b_out) depend on
 * parameters only: q_fold forms them into `fold` (q_fold_floats(cfg) floats) ahead of time -- e.g. beside the key/value
 * half while the backbone still runs -- and q_forward / q_backward take them (fold == NULL: formed inside q_forward,
 * kept in its workspace). */
size_t immtsf_mmf_xattn_q_fold_floats(const immtsf_fusion_cfg* cfg);
int immtsf_mmf_xattn_q_fold(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, float* fold, immtsf_stream_t stream);
int immtsf_mmf_xattn_q_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts, const float* KV,
                               const uint8_t* M_txt, const float* fold, float* Y_out, void* workspace, size_t workspace_bytes,
                               immtsf_stream_t stream);
/* defer_params != 0: only the data path (dY_ts, dKV) is enqueued; keep `workspace` and `scratch` alive and call
 * q_backward_params with them later, on any stream ordered after this call, for the parameter gradients. */
int immtsf_mmf_xattn_q_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts, const float* KV,
                                const uint8_t* M_txt, const float* fold, const float* dY_out, float* dY_ts, float* dKV,
                                void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                const immtsf_xadd_params* grads, int32_t defer_params, immtsf_stream_t stream);
int immtsf_mmf_xattn_q_backward_params(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts,
                                       const uint8_t* M_txt, const float* fold, void* workspace, size_t workspace_bytes,
                                       void* scratch, size_t scratch_bytes, const immtsf_xadd_params* grads,
                                       immtsf_stream_t stream);

/* ---- a6, low-rank form (csrc/xrank.hip).  Y_ts has C columns and residual_head has C rows, and nothing between proj_{q,k,v} and
 * the MHA in-projections, or between out_proj and residual_head, is nonlinear: per head the scores are [Y|1] G_h [E_txt|1]^T and
 * the head output is A_h (U_h [E_txt|1]^T) with parameter-only (C+1) x (d+1) / C x (d+1) matrices G_h, U_h.  The P half forms
 * them (fold), projects P = [E_txt|1] W_fold^T ((2C+1) H columns, row pitch xrank_pw) and, backward, turns dP into dE_txt and the
 * gradients of every parameter except LayerNorm's by the chain rule; the Q half is the whole attention + head + LayerNorm(C) +
 * blend on those columns in one kernel per direction.  Same function and gradients as the entry points above (fp32
 * reassociation); xrank_pw == 0: shape outside this form's limits, use the full-rank entry points.
 * cfg->in_h / aux_h / out_h as for the kv half: p_forward in_h = bf16 E_txt; q_backward out_h = bf16 dP; p_backward in_h = bf16
 * dP, aux_h = bf16 E_txt, out_h = bf16 dE_txt. */
int32_t immtsf_mmf_xrank_pw(const immtsf_fusion_cfg* cfg);
size_t immtsf_mmf_xrank_p_workspace_bytes(const immtsf_fusion_cfg* cfg);
size_t immtsf_mmf_xrank_p_scratch_bytes(const immtsf_fusion_cfg* cfg);
size_t immtsf_mmf_xrank_q_workspace_bytes(const immtsf_fusion_cfg* cfg);
/* E_txt (B*T, d) -> P (B*T, pw), b_HO (C) = W_res b_out + b_res.  folded != 0: immtsf_mmf_xrank_fold has already run on this
 * workspace / bHO (the fold depends on parameters only, so it may run ahead of time on any stream this call is ordered behind) */
int immtsf_mmf_xrank_fold(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, float* bHO, void* workspace, size_t workspace_bytes,
                          immtsf_stream_t stream);
int immtsf_mmf_xrank_p_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, float* P, float* bHO,
                               void* workspace, size_t workspace_bytes, int32_t folded, immtsf_stream_t stream);
/* dP, d b_HO -> dE_txt and the gradients of all parameters but ln_w / ln_b (overwritten; grads->ln_* are not touched) */
int immtsf_mmf_xrank_p_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, const float* dP,
                                const float* dbHO, float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch,
                                size_t scratch_bytes, const immtsf_xadd_params* grads, immtsf_stream_t stream);
/* The two halves of immtsf_mmf_xrank_p_backward, for callers that order them themselves.  ..._data: dE_txt = dP W_fold and dW_fold =
 * dP^T E_txt (+ column sums) into `scratch`.  ..._params: the chain rule from dW_fold to the block's parameter gradients, three
 * dependent multi-job launches of which the call runs [first, last) (0 <= first <= last <= 3) -- parameter-only work that only the
 * optimizer waits for, so its tail may run on another stream ordered behind ..._data (immtsf.train.FlagStep gives it to the
 * backbone's branch).  Same workspace / scratch as the combined call. */
int immtsf_mmf_xrank_p_backward_data(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, const float* dP,
                                     float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                     immtsf_stream_t stream);
int immtsf_mmf_xrank_p_backward_params(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* dbHO, void* workspace,
                                       size_t workspace_bytes, void* scratch, size_t scratch_bytes, const immtsf_xadd_params* grads,
                                       int32_t first, int32_t last, immtsf_stream_t stream);
/* The P half when the text side's producer ends in a linear map whose only consumer is this projection (TTF_T2V_XAttn's proj_out,
 * fusions/TTF_T2V_XAttn.py:182, called with immtsf_fusion_cfg.form | IMMTSF_FORM_NO_PROJ so that it hands over Z = its LayerNorm +
 * dropout output): P = [Z | 1] [W_fold W_po | W_fold b_po + b_fold]^T -- the (B T) x d x d product, its data and weight gradients
 * become PW-row products, E_txt / dE_txt are never formed.  backward_data_z writes dZ (and dWc into `scratch`; cfg->bwd_phase
 * IMMTSF_BWD_PHASE_A: only dZ, IMMTSF_BWD_WGRAD_A: only dWc / dbc -- parameter-gradient work for a stream of its own); backward_pre_z --
 * parameters only -- the producer's gradients (g_proj_w, g_proj_b) and dW_fold / db_fold for immtsf_mmf_xrank_p_backward_params. */
int immtsf_mmf_xrank_fold_z(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* proj_w, const float* proj_b, float* bHO,
                            void* workspace, size_t workspace_bytes, immtsf_stream_t stream);      /* the parameter-only part, ahead of time */
int immtsf_mmf_xrank_p_forward_z(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* proj_w, const float* proj_b,
                                 const float* Z, float* P, float* bHO, void* workspace, size_t workspace_bytes, int32_t folded,
                                 immtsf_stream_t stream);
int immtsf_mmf_xrank_p_backward_data_z(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* proj_w, const float* proj_b,
                                       const float* Z, const float* dP, float* dZ, void* workspace, size_t workspace_bytes, void* scratch,
                                       size_t scratch_bytes, immtsf_stream_t stream);
/* the basis of the low-rank gradient ..._backward_data_z sends back (dZ = dP Wc): *basis = Wc (rank x d, fp32, inside `workspace`: valid
 * while the forward's workspace lives), *rank = immtsf_mmf_xrank_pw(cfg).  With IMMTSF_FORM_LOWRANK_OUT in cfg->form the data call leaves
 * dZ unwritten and the producer's backward is given {dP, Wc, rank, ld = rank} as immtsf_fusion_cfg.lr_grad.  (ABI 6) */
int immtsf_mmf_xrank_lowrank_basis(const immtsf_fusion_cfg* cfg, void* workspace, size_t workspace_bytes, const float** basis, int32_t* rank);
int immtsf_mmf_xrank_p_backward_pre_z(const immtsf_fusion_cfg* cfg, const float* proj_w, const float* proj_b, void* workspace,
                                      size_t workspace_bytes, void* scratch, size_t scratch_bytes, float* g_proj_w, float* g_proj_b,
                                      immtsf_stream_t stream);
int immtsf_mmf_xrank_q_forward(const immtsf_fusion_cfg* cfg, const float* ln_w, const float* ln_b, const float* Y_ts, const float* P,
                               const float* bHO, const uint8_t* M_txt, float* Y_out, void* workspace, size_t workspace_bytes,
                               immtsf_stream_t stream);
int immtsf_mmf_xrank_q_backward(const immtsf_fusion_cfg* cfg, const float* ln_w, const float* Y_ts, const float* P, const uint8_t* M_txt,
                                const float* dY_out, float* dY_ts, float* dP, float* dbHO, float* d_ln_w, float* d_ln_b, void* workspace,
                                size_t workspace_bytes, immtsf_stream_t stream);

/* The Q half's training step in ONE launch: forward, the masked-MSE loss of immtsf_masked_mse_counted against truth / mask (B*T, C)
 * with the per-variable observation counts cnt (C), and the backward seeded with d loss = grad_scale -- everything the backward
 * needs from the forward and the loss is local to a row once the counts are known.  Outputs: loss (1 float), dY_ts, dP (+ bf16
 * image in cfg->out_h), d b_HO, d ln_w, d ln_b; Y_out optional (NULL: not written).  ticket: two zero-initialised device words the
 * call leaves zero (calls that share them must be ordered on their streams).  done_flag (optional): set to 1 as soon as dY_ts is
 * complete, for a consumer on another stream that waits with immtsf_flag_wait (the rest of the kernel then overlaps with it). */
size_t immtsf_mmf_xrank_q_train_scratch_bytes(const immtsf_fusion_cfg* cfg);
int immtsf_mmf_xrank_q_train(const immtsf_fusion_cfg* cfg, const float* ln_w, const float* ln_b, const float* Y_ts, const float* P,
                             const float* bHO, const uint8_t* M_txt, const float* truth, const float* mask, const float* cnt,
                             float grad_scale, float* Y_out, float* loss, float* dY_ts, float* dP, float* dbHO, float* d_ln_w,
                             float* d_ln_b, void* scratch, size_t scratch_bytes, uint32_t* ticket, int32_t* done_flag,
                             immtsf_stream_t stream);

/* ---- a9, short sequences with wide heads (round 3; csrc/attn_mid.hip): softmax(scale Q K^T) V per (batch, head) on (B, L, H, E) /
 * (B, S, H, E) / (B, S, H, D) tensors for L, S <= 32 and E, D <= 256 (multiples of 4) -- PatchTST's FullAttention over the patches
 * of a variable (layers/SelfAttention_Family.py:50-77).  One launch per direction instead of batched GEMMs + row softmax
 * (3 forward, 5 backward); exact fp32; P (B, H, L, S) = the probabilities before dropout, the dropout indexing is that of
 * immtsf_softmax_rows_* (site, ((b H + h) L + l) S + s).  causal != 0: TriangularCausalMask. */
int32_t immtsf_attn_mid_supported(int32_t L, int32_t S, int32_t E, int32_t D);
int immtsf_attn_mid_forward(const float* q, const float* k, const float* v, int32_t B, int32_t L, int32_t S, int32_t H, int32_t E, int32_t D,
                            float scale, int32_t causal, float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, float* P,
                            float* out, immtsf_stream_t stream);
int immtsf_attn_mid_backward(const float* q, const float* k, const float* v, const float* P, const float* dout, int32_t B, int32_t L, int32_t S,
                             int32_t H, int32_t E, int32_t D, float scale, float p_drop, uint64_t seed, uint64_t site,
                             const uint64_t* seed_step_dev, float* dq, float* dk, float* dv, immtsf_stream_t stream);

/* ---- a7: MMF_GR_Add.forward (fusions/MMF_GR_Add.py:31-61; nn.GRU gate order r,z,n; hidden_dim = Hd) */
typedef struct immtsf_gr_params {
    float *w_ih, *w_hh, *b_ih, *b_hh; /* (3Hd, C+d),(3Hd,Hd),(3Hd),(3Hd)  gru.*_l0 */
    float *res_w, *res_b;             /* (C,Hd),(C) */
    float *gate_w, *gate_b;           /* (C, C+d),(C) */
    float *ln_w, *ln_b;               /* (C) */
} immtsf_gr_params;

size_t immtsf_mmf_gr_add_workspace_bytes(const immtsf_fusion_cfg* cfg, int32_t hidden);
size_t immtsf_mmf_gr_add_scratch_bytes(const immtsf_fusion_cfg* cfg, int32_t hidden);
int immtsf_mmf_gr_add_forward(const immtsf_fusion_cfg* cfg, int32_t hidden, const immtsf_gr_params* p,
                              const float* Y_ts, const float* E_txt, const uint8_t* M_txt, float* Y_out, void* workspace,
                              size_t workspace_bytes, immtsf_stream_t stream);
int immtsf_mmf_gr_add_backward(const immtsf_fusion_cfg* cfg, int32_t hidden, const immtsf_gr_params* p,
                               const float* Y_ts, const float* E_txt, const uint8_t* M_txt, const float* dY_out,
                               float* dY_ts, float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch,
                               size_t scratch_bytes, const immtsf_gr_params* grads, immtsf_stream_t stream);

/* MMF_GR_Add in SPLIT form (round 5; csrc/gr_train.hip; reference fusions/MMF_GR_Add.py:41-60).  The GRU's input map and the gate net
 * are linear in x = [Y_ts ; E_txt], so their text columns run where the text is -- the P half, ahead of the backbone:
 *   P (B T, pw) = E_txt [W_ih[:, C:] ; W_g[:, C:]]^T + [b_ih ; b_g],   pw = immtsf_mmf_gr_pw = 3 hidden + C rounded up to 8 --
 * and what is left between the backbone's forward and backward (the C-column products of Y_ts, the recurrence, residual head,
 * LayerNorm(C), dropout, gate, blend; for a training step the masked MSE of immtsf_masked_mse_counted and the backward of all of it
 * through time) is ONE launch, a wave per window: immtsf_mmf_gr_q_train.  Limits: T <= 64, C <= 16, hidden <= 16, d % 8 == 0
 * (immtsf_mmf_gr_pw returns 0 outside them: use the as-written form above).  Parameter gradients: the P half's backward WRITES the text
 * columns of d W_ih / d W_g and d b_ih / d b_g; q_train ADDS (atomics) the Y columns of d W_ih / d W_g and d W_hh, d b_hh, d res_w,
 * d res_b, d ln_w, d ln_b into buffers the caller hands in zeroed.  truth == NULL: q_train only writes Y_out (inference).  scratch: B
 * floats; ticket: two zero-initialised device words the call leaves zero; done_flag (optional): set to 1 as soon as dY_ts is complete
 * (a consumer on another stream waits with immtsf_flag_wait).  (ABI 6) */
int32_t immtsf_mmf_gr_pw(const immtsf_fusion_cfg* cfg, int32_t hidden);
size_t immtsf_mmf_gr_p_workspace_bytes(const immtsf_fusion_cfg* cfg, int32_t hidden);
size_t immtsf_mmf_gr_p_scratch_bytes(const immtsf_fusion_cfg* cfg, int32_t hidden);
int immtsf_mmf_gr_p_forward(const immtsf_fusion_cfg* cfg, int32_t hidden, const immtsf_gr_params* p, const float* E_txt, float* P, void* workspace,
                            size_t workspace_bytes, immtsf_stream_t stream);
int immtsf_mmf_gr_p_backward(const immtsf_fusion_cfg* cfg, int32_t hidden, const immtsf_gr_params* p, const float* E_txt, const float* dP,
                             float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                             const immtsf_gr_params* grads, immtsf_stream_t stream);
int immtsf_mmf_gr_q_train(const immtsf_fusion_cfg* cfg, int32_t hidden, const immtsf_gr_params* p, const float* Y_ts, const float* P,
                          const uint8_t* M_txt, const float* truth, const float* mask, const float* cnt, float grad_scale, float* Y_out,
                          float* loss, float* dY_ts, float* dP, const immtsf_gr_params* grads, float* scratch, uint32_t* ticket,
                          int32_t* done_flag, immtsf_stream_t stream);

/* ---- a14: tPatchGNN time-aware patch encoder, LearnableTE + TTCN (models/tPatchGNN.py:176-195), fused.
 * x, tt, mask: (P, L) with P = B*N*M patches (the reference's (B*N*M, L, 1) tensors); F = 1 + te_dim,
 * K = ttcn_dim = hid_dim - 1.  out: (P, K) = relu(pooled + T_bias).  Limit: F*K <= 1024 (IMMTSF_EUNSUPPORTED). */
typedef struct immtsf_ttcn_params {
    float *te_scale_w, *te_scale_b; /* (1),(1)          te_scale */
    float *te_per_w, *te_per_b;     /* (te_dim-1) each  te_periodic */
    float *W1, *b1;                 /* (K,F),(K)        Filter_Generators.0 */
    float *W2, *b2;                 /* (K,K),(K)        Filter_Generators.2 */
    float *W3, *b3;                 /* (F*K,K),(F*K)    Filter_Generators.4 */
    float* T_bias;                  /* (K)              T_bias (1,K) */
} immtsf_ttcn_params;

size_t immtsf_ttcn_workspace_bytes(int32_t P, int32_t L, int32_t te_dim, int32_t ttcn_dim);
size_t immtsf_ttcn_scratch_bytes(int32_t P, int32_t L, int32_t te_dim, int32_t ttcn_dim);
/* (B, M, L, N) batch tensors (values, time stamps, observation mask) -> the (B*N*M, L) patch rows of
 * models/tPatchGNN.py:271-275 (three permute(0, 3, 1, 2).reshape copies there), in one launch. */
int immtsf_patch_flatten3(const float* x, const float* tt, const float* mask, int32_t B, int32_t M, int32_t L, int32_t N, float* ox,
                          float* ott, float* omask, immtsf_stream_t stream);

/* bf16 mode with L <= 64, te_dim <= 15, ttcn_dim <= 32: the whole encoder of a patch runs on one CU (time embedding,
 * three filter-generator layers and the filter logits as MFMAs on LDS tiles, masked softmax and pooling in the
 * accumulator registers), one kernel per direction; the backward recomputes the forward on chip.  Otherwise the three
 * layers run on the MFMA GEMM over all P*L slots (`precision` as everywhere) and the masked softmax + pooling are
 * streaming kernels.  `workspace` holds the saved-for-backward state in both cases.  out / dout have leading
 * dimension out_ld >= K; flag_col >= K (or -1): column that receives the patch-non-empty flag (any mask > 0), so the
 * caller's [embedding ; flag] concatenation (models/tPatchGNN.py:268-270) needs no extra kernels */
int immtsf_ttcn_forward(int32_t P, int32_t L, int32_t te_dim, int32_t ttcn_dim, int32_t precision, const float* x,
                        const float* tt, const float* mask, const immtsf_ttcn_params* p, float* out, int32_t out_ld,
                        int32_t flag_col, void* workspace, size_t workspace_bytes, immtsf_stream_t stream);
/* dout (P,K) -> every parameter gradient (overwritten; te_accumulate != 0: the four time-embedding gradients are ADDED to
 * what their buffers hold -- the decoder's LearnableTE shares those parameters and its backward adds into the same, zero-
 * initialised buffers).  No gradient flows to x / tt / mask (data).
 * NOTE: consumes the workspace (the saved softmax weights are overwritten): one backward per forward. */
int immtsf_ttcn_backward(int32_t P, int32_t L, int32_t te_dim, int32_t ttcn_dim, int32_t precision, const float* x,
                         const float* tt, const float* mask, const immtsf_ttcn_params* p, const float* out,
                         const float* dout, int32_t out_ld, void* workspace, size_t workspace_bytes, void* scratch,
                         size_t scratch_bytes, const immtsf_ttcn_params* grads, int32_t te_accumulate, immtsf_stream_t stream);

/* ---- tPatchGNN adaptive-graph stage (models/tPatchGNN.py:205-238; gcn/nconv/linear :29-84), one (window, patch)
 * cell per workgroup, everything in LDS.  x, out, dout, dx: (B, N, M, D) contiguous.  Pointers in state_dict order. */
typedef struct immtsf_gcn_params {
    float *nodevec1, *nodevec2;     /* (N,nd), (nd,N) */
    float *gate1_w, *gate1_b;       /* (1,D+nd),(1)   nodevec_gate1[l].0 */
    float *gate2_w, *gate2_b;       /*                nodevec_gate2[l].0 */
    float *lin1_w, *lin1_b;         /* (nd,D),(nd)    nodevec_linear1[l] */
    float *lin2_w, *lin2_b;         /*                nodevec_linear2[l] */
    float *mlp_w, *mlp_b;           /* (D,(order+1)*D[,1,1]),(D)  gconv[l].mlp.mlp (adaptive support only) */
} immtsf_gcn_params;
/* LDS bytes one cell needs, 0 when it does not fit a CU's 160 KB (the caller keeps its own formulation then) */
size_t immtsf_tpatchgnn_gcn_lds_bytes(int32_t N, int32_t D, int32_t nd, int32_t order);
int immtsf_tpatchgnn_gcn_forward(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* x,
                                 const immtsf_gcn_params* p, float* out, immtsf_stream_t stream);
/* the same forward, also leaving what the backward reads of every cell -- X, the hop slabs, the mixing layer's pre-activation, the
 * node-vector intermediates, adjacency and gates: immtsf_tpatchgnn_gcn_saved_floats(...) floats (5 KB per cell at N 8, D 32, nd 10) -- in
 * `saved`; immtsf_tpatchgnn_gcn_backward_saved reads them back instead of recomputing the cell (a third of the recomputing backward's
 * time: every phase of the cell is a barrier and a chain of dependent LDS reads) */
size_t immtsf_tpatchgnn_gcn_saved_floats(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order);
int immtsf_tpatchgnn_gcn_forward_saved(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* x,
                                       const immtsf_gcn_params* p, float* out, float* saved, immtsf_stream_t stream);
int immtsf_tpatchgnn_gcn_backward_saved(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* saved,
                                        const immtsf_gcn_params* p, const float* dout, float* dx, const immtsf_gcn_params* grads,
                                        immtsf_stream_t stream);
/* recomputes the cell's forward from x; dx overwritten; parameter gradients are ACCUMULATED (atomics) into `grads`,
 * which the caller zeroes (or lets run on as a running sum) */
int immtsf_tpatchgnn_gcn_backward(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* x,
                                  const immtsf_gcn_params* p, const float* dout, float* dx, const immtsf_gcn_params* grads,
                                  immtsf_stream_t stream);

/* ---- tPatchGNN forecast decoder (models/tPatchGNN.py:168-174 as applied at :283-291): Linear(D+E, H) -> ReLU ->
 * Linear(H, H) -> ReLU -> Linear(H, 1) on [h[b, n] ; te[b, lp]] for every (b, n, lp), one kernel per direction (the
 * first layer is separable in h and te; exact fp32 in both precision modes).  h: (B, N, D) encoder state, te: (B, Lp, E)
 * time embedding of the prediction times, out / dout: (B, Lp, N).  H must be 32 (the reference's hid_dim) and N, Lp small enough for one
 * window's vectors to sit in LDS: immtsf_tpatchgnn_decoder_lds_bytes returns 0 otherwise (callers then use the GEMM
 * chain).  Pointers in nn.Sequential order: decoder.0 / .2 / .4 */
typedef struct immtsf_decoder_params {
    float *W1, *b1; /* (H, D+E), (H) */
    float *W2, *b2; /* (H, H), (H)   */
    float *W3, *b3; /* (1, H), (1)   */
} immtsf_decoder_params;
size_t immtsf_tpatchgnn_decoder_lds_bytes(int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H);
int immtsf_tpatchgnn_decoder_forward(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, const float* h,
                                     const float* te, const immtsf_decoder_params* p, float* out, immtsf_stream_t stream);
/* dh (B, N, D) and dte (B, Lp, E) are overwritten; parameter gradients are ACCUMULATED (atomics) into `grads`, which the
 * caller zeroes (or lets run on as a running sum).  Recomputes the forward: nothing is saved. */
int immtsf_tpatchgnn_decoder_backward(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, const float* h,
                                      const float* te, const immtsf_decoder_params* p, const float* dout, float* dh,
                                      float* dte, const immtsf_decoder_params* grads, immtsf_stream_t stream);
/* The same with a precision argument (0: the exact fp32 kernels above; 1: bf16 operands / fp32 accumulation on MFMA tiles for the
 * second layer and its gradients -- H = 32 is one K-step -- with the first layer and the (n, lp) bookkeeping unchanged).
 * Reference: models/tPatchGNN.py:168-174, 283-291. */
int immtsf_tpatchgnn_decoder_forward_p(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                                       const float* te, const immtsf_decoder_params* p, float* out, immtsf_stream_t stream);
int immtsf_tpatchgnn_decoder_backward_p(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                                        const float* te, const immtsf_decoder_params* p, const float* dout, float* dh,
                                        float* dte, const immtsf_decoder_params* grads, immtsf_stream_t stream);
/* The decoder with tPatchGNN's LearnableTE of the prediction times inside (models/tPatchGNN.py:176-180 applied at :283-285): t (B, Lp)
 * instead of te; the kernels build te[b, lp, :] = [w0 t + b0 ; sin(w t + b)] while they stage a window, and the backward reduces
 * d te to the four parameter gradients on the spot (ACCUMULATED into tgrads by atomics, like `grads`: the same buffers may be shared
 * with the patch encoder's time embedding).  No (B, Lp, E) tensor, no separate Time2Vec launches.  E <= 16. */
typedef struct immtsf_time2vec_params {
    float *w0, *b0; /* Linear(1, 1): (1, 1), (1)          */
    float *w, *b;   /* Linear(1, E - 1): (E - 1, 1), (E - 1); NULL when E == 1 */
} immtsf_time2vec_params;
int immtsf_tpatchgnn_decoder_forward_te(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                                        const float* t, const immtsf_time2vec_params* tp, const immtsf_decoder_params* p, float* out,
                                        immtsf_stream_t stream);
int immtsf_tpatchgnn_decoder_backward_te(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                                         const float* t, const immtsf_time2vec_params* tp, const immtsf_decoder_params* p, const float* dout,
                                         float* dh, const immtsf_decoder_params* grads, const immtsf_time2vec_params* tgrads,
                                         immtsf_stream_t stream);

/* ---- a16: masked per-variable MSE, compute_error(truth, pred, mask, "MSE", "mean") lib/evaluation.py:17-62.
 * pred/truth/mask (rows, C).  err_sum, cnt: (C) device buffers (outputs of the local reduction; under data
 * parallelism the caller all-reduces them before calling _finish).  scratch: >= 128*C floats.  loss: device scalar.
 * dpred: (rows, C). */
int immtsf_masked_mse_sums(const float* truth, const float* pred, const float* mask, int32_t rows, int32_t C,
                           float* err_sum, float* cnt, float* scratch, immtsf_stream_t stream);
int immtsf_masked_mse_finish(const float* truth, const float* pred, const float* mask, int32_t rows, int32_t C,
                             const float* err_sum, const float* cnt, float* loss, float* dpred, float grad_scale,
                             immtsf_stream_t stream);

/* the two calls above as ONE single-workgroup kernel, for rows*C <= IMMTSF_MSE_SMALL_MAX and no reduction across
 * ranks between the sums and the divide (IMMTSF_EUNSUPPORTED otherwise: use the pair).  cnt_global (may be NULL):
 * per-variable observation counts of the global batch, used for the divide instead of the local counts (data
 * parallelism with the counts reduced when the batch was built).  err_sum, cnt, dpred may be NULL. */
#define IMMTSF_MSE_SMALL_MAX (1 << 17)
int immtsf_masked_mse(const float* truth, const float* pred, const float* mask, int32_t rows, int32_t C, const float* cnt_global,
                      float* err_sum, float* cnt, float* loss, float* dpred, float grad_scale, immtsf_stream_t stream);

/* the same loss when the per-variable observation counts are known before the step (they depend on the mask only; the
 * reference recomputes them inside compute_error, lib/evaluation.py:36-47): no element waits for a reduction, so the
 * work spreads over up to 64 workgroups and the last one to finish adds the partial losses in index order
 * (deterministic).  C <= 64 (IMMTSF_EUNSUPPORTED above).  scratch: IMMTSF_MSE_COUNTED_SCRATCH floats owned by the caller,
 * zero-initialised ONCE; word 0 is a ticket counter that every call leaves at zero, so the buffer can be reused by the
 * next call on the same stream (not by concurrent calls). */
#define IMMTSF_MSE_COUNTED_SCRATCH 65
int immtsf_masked_mse_counted(const float* truth, const float* pred, const float* mask, int32_t rows, int32_t C,
                              const float* cnt_global, float* scratch, float* loss, float* dpred, float grad_scale,
                              immtsf_stream_t stream);

/* ---- device-side batch builder: the reference's collate functions over a dataset that is
 * resident in HBM.  Replaces lib/parse_datasets.py:252-295 (variable_time_collate_fn), :298-366 +
 * lib/utils.py:359-413 (patch_variable_time_collate_fn / split_and_patch_batch) and :764-824 (multimodal wrapper).
 * All arrays are device memory owned by the caller; window w owns rows row_off[w]..row_off[w+1]) of tt/vals/mask
 * (times ascending, chunk-relative; the first hist_len[w] rows have tt < history) and notes
 * note_off[w]..note_off[w+1]) of note_tau/note_src (note_src = row of the note in the resident matrix `emb`). */
typedef struct immtsf_store {
    const float* tt;            /* [rows] */
    const float* vals;          /* [rows, C] */
    const float* mask;          /* [rows, C] 0/1 */
    const int64_t* row_off;     /* [W+1] */
    const int32_t* hist_len;    /* [W] */
    const float* note_tau;      /* [notes] chunk-relative note times */
    const int64_t* note_src;    /* [notes] */
    const int64_t* note_off;    /* [W+1] */
    const float* emb;           /* [*, d_m] resident text-embedding matrix */
    int32_t C, d_m;
} immtsf_store;
/* outputs zero-padded like pad_sequence: obs_* (B,Lmax[,C]) (all three NULL to skip, as tPatchGNN does),
 * pred_* (B,Lpmax[,C]); times divided by time_max (= history + pred_window) as normalize_masked_tp does */
int immtsf_collate_series(const immtsf_store* s, const int32_t* window_ids, int32_t B, int32_t Lmax, int32_t Lpmax,
                          float time_max, float* obs_tp, float* obs_data, float* obs_mask, float* pred_tp,
                          float* pred_data, float* pred_mask, immtsf_stream_t stream);
/* tPatchGNN patches: out (B,npatch,Lp,C); patch i covers [i*stride, i*stride+size), the last one up to `history`;
 * per (b,i,c) the observed points in time order in slots 0.., zeros behind */
int immtsf_collate_patches(const immtsf_store* s, const int32_t* window_ids, int32_t B, int32_t npatch, float patch_size,
                           float patch_stride, float history, int32_t Lp, float time_max, float* obs_tp, float* obs_data,
                           float* obs_mask, immtsf_stream_t stream);
/* tau (B,Nmax), notes (B,Nmax,d_m) zero padded (either may be NULL), and the packed ragged index: lengths[B],
 * offsets[B+1] (int32, bit-exact with immtsf_ragged_index on the padded tensor), rowmap[sum] -> row in `emb` (or NULL) */
int immtsf_collate_notes(const immtsf_store* s, const int32_t* window_ids, int32_t B, int32_t Nmax, float* tau, float* notes,
                         int32_t* lengths, int32_t* offsets, int32_t* rowmap, immtsf_stream_t stream);

/* ---- building blocks exported for tests and for the layers/ mirror ------------------------------------------- */
/* C = act(alpha * op(A) op(B)^T + bias); layout 0 = NT (A:(M,K), B:(N,K)), 1 = NN (B:(K,N)), 2 = TN (A:(K,M), B:(K,N)) */
int immtsf_gemm(int32_t layout, int32_t precision, const float* A, int32_t lda, const float* B, int32_t ldb, float* C,
                int32_t ldc, const float* bias, int32_t M, int32_t N, int32_t K, float alpha, int32_t accumulate,
                int32_t act, immtsf_stream_t stream);
/* nn.Linear autograd in one call (x:(M,K), W:(N,K), dy:(M,N)): dx = dy W, optionally masked with relu_x (dx[m,k] = 0
 * where relu_x[m,k] <= 0: x was a ReLU output, so dx is the gradient of the PRE-activation); dW = dy^T x; db = column
 * sums of dy (reduced inside the dW GEMM).  Any of dx / dW+db may be NULL; db needs dW.  grads_prezeroed: dW and db
 * already hold zeros (one caller-side fill for many layers), so a split-K weight gradient skips its own memsets. */
int immtsf_linear_backward(int32_t precision, const float* x, const float* W, const float* dy, int32_t M, int32_t N,
                           int32_t K, float* dx, const float* relu_x, float* dW, float* db, int32_t grads_prezeroed,
                           immtsf_stream_t stream);
/* Time2Vec / tPatchGNN LearnableTE rows (fusions/TTF_T2V_XAttn.py:7-24, models/tPatchGNN.py:176-180):
 * out[r,0] = w0*t[r]+b0, out[r,j] = sin(w[j-1]*t[r]+b[j-1]).  backward: parameter gradients from dout (t is data);
 * scratch >= 2*64*d floats. */
int immtsf_time2vec_forward(const float* t, int32_t rows, int32_t d, const float* w0, const float* b0, const float* w,
                            const float* b, float* out, immtsf_stream_t stream);
int immtsf_time2vec_backward(const float* t, int32_t rows, int32_t d, const float* w, const float* b, const float* dout,
                             float* dw0, float* db0, float* dw, float* db, float* scratch, int32_t accumulate,
                             immtsf_stream_t stream);
/* bf16 twins.  In bf16 precision the MFMA operands are rounded to bf16 anyway; a registered fp32 range
 * [base, base+count) whose owner keeps a bf16 copy (same element offsets) lets every GEMM whose B operand lies inside
 * it -- the weights of the forward (NT) and data-gradient (NN) GEMMs -- fetch half as many bytes with no conversion.
 * immtsf_adam_step[_dev] on a registered parameter range writes the twin together with the parameters; after any other
 * write to the range call immtsf_f32_to_bf16 (FlatTrainer.refresh_twins).  Process-wide registry (<= 16 ranges), reader /
 * writer locked: register / unregister from any thread while other threads launch; an unregistered twin must stay
 * allocated until the work already enqueued against it has run. */
int immtsf_bf16_twin_register(const float* base, void* twin, size_t count);
int immtsf_bf16_twin_unregister(const float* base);
int immtsf_bf16_twin_enable(int32_t on);            /* A/B switch for measurements; default on */
int immtsf_f32_to_bf16(const float* src, void* dst, size_t n, immtsf_stream_t stream);
/* the reverse (exact widening): used with immtsf_f32_to_bf16 around a bf16 gradient all-reduce (FlatTrainer grad_wire) */
int immtsf_bf16_to_f32(const void* src, float* dst, size_t n, immtsf_stream_t stream);
/* nn.Linear in the bf16 dataflow (precision 1) with caller workspaces -- the layers/ mirror's projections (layers/SelfAttention_Family.py:
 * 199-214 query / key / value / out projection of AttentionLayer; models/PatchTST.py FlattenHead): the input is cast ONCE (x16: M*K bf16,
 * kept for the backward's weight gradients), the weights come from their registered bf16 twins (else are cast into w16[i]: N*K bf16, may be
 * NULL when twins exist), and the products run on the bf16-in-HBM kernels (csrc/gemm2.hip / gemm3.hip) instead of converting fp32 tiles
 * while staging.  nl (1..3) layers that share ONE input -- a self-attention's q | k | v -- are ONE forward launch (y[i]: M x N each) and,
 * backward, one cast of the nl upstream gradients (dy16: nl*M*N bf16), nl accumulating data-gradient launches (dx = sum_i dy_i W_i;
 * dx NULL: not needed) and ONE grouped weight-gradient launch (dW[i] = dy_i^T x, db[i] = column sums; db NULL or db[i] NULL: no bias).
 * act: 0 / 1 (ReLU, nl == 1 only).  grads_prezeroed: bit 0 = dW / db are zero already, bit 1 = dy16 holds the bf16 images already (a second
 * call that only forms the weight gradients -- dx NULL -- on any stream ordered behind the first: the data gradient stays on the dependent
 * chain, the weight gradients leave it).  W / w16 / b / y / dy / dW / db: HOST arrays of nl device pointers.  IMMTSF_EUNSUPPORTED: K or N
 * not a multiple of 8, nl > 3 (the caller falls back to immtsf_gemm / immtsf_linear_backward). */
int immtsf_linear_bf16_forward(int32_t nl, const float* x, void* x16, const float* const* W, void* const* w16, const float* const* b,
                               float* const* y, int32_t M, int32_t N, int32_t K, int32_t act, immtsf_stream_t stream);
int immtsf_linear_bf16_backward(int32_t nl, const void* x16, const float* const* W, void* const* w16, const float* const* dy, void* dy16,
                                float* dx, float* const* dW, float* const* db, int32_t M, int32_t N, int32_t K,
                                int32_t grads_prezeroed, immtsf_stream_t stream);
/* batched over (outer, inner) with element strides, used by FullAttention (layers/SelfAttention_Family.py:50-77) */
int immtsf_gemm_batched(int32_t layout, int32_t precision, const float* A, int32_t lda, int64_t sA_o, int64_t sA_i,
                        const float* B, int32_t ldb, int64_t sB_o, int64_t sB_i, float* C, int32_t ldc, int64_t sC_o,
                        int64_t sC_i, int32_t n_outer, int32_t n_inner, int32_t M, int32_t N, int32_t K, float alpha,
                        immtsf_stream_t stream);
/* softmax over the last dim of (B,H,L,S) scores in place (-> P), A = dropout(P) (A may alias sc when p == 0);
 * causal != 0 applies the TriangularCausalMask of utils/masking.py (key s visible to query l iff s <= l);
 * seed_step_dev: optional device counter added to `seed` (see immtsf_fusion_cfg.seed_step_dev) */
int immtsf_softmax_rows_forward(float* sc, float* A, int32_t B, int32_t H, int32_t L, int32_t S, const uint8_t* live,
                                float p_drop, uint64_t seed, uint64_t site, int32_t causal, const uint64_t* seed_step_dev,
                                immtsf_stream_t stream);
int immtsf_softmax_rows_backward(float* dA, const float* P, int32_t B, int32_t H, int32_t L, int32_t S, float p_drop,
                                 uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, immtsf_stream_t stream);
/* FullAttention over SHORT sequences (L <= IMMTSF_ATTN_SHORT_MAX; tPatchGNN attends over M = 2 patches) on the packed
 * in-projection output qkv (B, L, 3, H, E): out (B, L, H, E) = dropout(softmax(scale q k^T [causal])) v with one thread per
 * (sequence, head, position) -- one launch per direction instead of 2 batched GEMMs + softmax (3 + 5 launches).  Same
 * dropout stream as immtsf_softmax_rows_* (site, ((b*H+h)*L + l)*L + s).  The backward recomputes the softmax and
 * overwrites dqkv (B, L, 3, H, E).  Exact fp32.  IMMTSF_EUNSUPPORTED when L is larger, E > 64, E % 4 != 0 or a pointer is
 * not 16-byte aligned. */
#define IMMTSF_ATTN_SHORT_MAX 8
int immtsf_attention_short_forward(const float* qkv, int32_t B, int32_t L, int32_t H, int32_t E, float scale, int32_t causal,
                                   float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, float* out,
                                   immtsf_stream_t stream);
int immtsf_attention_short_backward(const float* qkv, const float* dout, int32_t B, int32_t L, int32_t H, int32_t E, float scale,
                                    int32_t causal, float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev,
                                    float* dqkv, immtsf_stream_t stream);
/* LayerNorm(+dropout) rows: xhat,rstd may be NULL in forward when no backward follows */
int immtsf_layernorm_forward(const float* x, int32_t rows, int32_t d, const float* gamma, const float* beta, float eps,
                             float* xhat, float* rstd, float* z, float p_drop, uint64_t seed, uint64_t site,
                             immtsf_stream_t stream);
/* backward: dz_dy is overwritten with dy; scratch >= 64*d floats (both parameter gradients come from one pass) */
int immtsf_layernorm_backward(float* dz_dy, int32_t rows, int32_t d, const float* gamma, const float* xhat,
                              const float* rstd, float* dx, float* dgamma, float* dbeta, float* scratch,
                              float p_drop, uint64_t seed, uint64_t site, immtsf_stream_t stream);
/* keep-mask (1 = kept) of `n` consecutive elements of a dropout site: lets tests feed the oracle identical masks */
int immtsf_dropout_mask(uint64_t seed, uint64_t site, uint64_t n, float p_drop, uint8_t* out, immtsf_stream_t stream);
/* dropout site ids used by the fusion blocks */
#define IMMTSF_SITE_T2V_ATTN 1  /* index ((b*T+t)*H+h)*N+n */
#define IMMTSF_SITE_T2V_OUT 2   /* index (b*T+t)*d+e */
#define IMMTSF_SITE_REC_OUT 3   /* index (b*T+t)*d+e */
#define IMMTSF_SITE_XADD_ATTN 4 /* index ((b*H+h)*T+l)*T+s */
#define IMMTSF_SITE_XADD_OUT 5  /* index (b*T+t)*C+c */
#define IMMTSF_SITE_GR_OUT 6    /* index (b*T+t)*C+c */
#define IMMTSF_SITE_LAYER_BASE 16

/* fused clip-by-global-norm + Adam on a flat parameter buffer (main.py:1098-1101: clip_grad_norm_(1.0) then
 * Adam(lr, weight_decay) with torch's L2-style weight decay).  norm_scratch: >= 1024 floats. */
int immtsf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                     float* norm_scratch, immtsf_stream_t stream);
/* graph-friendly form: *step_dev (device int64, >= 0) is incremented by the kernel and used as the step number;
 * dropout_step_dev (device uint64, may be NULL) is incremented too (see immtsf_fusion_cfg.seed_step_dev). */
int immtsf_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int64_t* step_dev, float max_norm,
                         float* norm_scratch, uint64_t* dropout_step_dev, immtsf_stream_t stream);

/* immtsf_adam_step_dev that also leaves `grad` ZERO behind the update: the next step's zero-fill of the gradient buffer (a
 * 32 MB memset in front of everything at the benchmark configuration) rides on the pass Adam makes over it anyway. */
int immtsf_adam_step_dev_zero(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t* step_dev, float max_norm, float* norm_scratch,
                              uint64_t* dropout_step_dev, immtsf_stream_t stream);

/* immtsf_adam_step_dev[_zero] behind a guard word: *skip_flag != 0 (device int32, e.g. the time-out word of immtsf_flag_wait)
 * DROPS the step -- parameters, moments and *step_dev stay as they are; the dropout counter still advances and, with zero_grad,
 * the gradient is still left zero -- so a captured step whose flag hand-over failed cannot apply half-finished gradients.
 * No reference counterpart: launch-model plumbing (immtsf.train.FlagStep). */
int immtsf_adam_step_guarded(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t* step_dev, float max_norm, float* norm_scratch,
                             uint64_t* dropout_step_dev, const int32_t* skip_flag, int32_t zero_grad, immtsf_stream_t stream);

/* The same step as two calls, for a sharded optimizer (immtsf.train.FlatTrainer(shard_optimizer=True): each rank owns
 * 1/W of the flat buffers): adam_sqnorm writes 1024 partial sums of squares of `grad` (this rank's shard of the
 * reduce-scattered gradient) to norm_scratch and bumps the device counters (both may be NULL); the caller sum-all-reduces
 * norm_scratch[0..1024) over the ranks; adam_apply then clips by the global norm and updates param / moments (and `twin`,
 * the bf16 image of `param`: NULL = the registered twin, if any).  step_dev NULL: bias corrections from the host `step`. */
int immtsf_adam_sqnorm(const float* grad, uint64_t n, float* norm_scratch, int64_t* step_dev, uint64_t* dropout_step_dev,
                       immtsf_stream_t stream);
int immtsf_adam_apply(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int32_t step, const int64_t* step_dev, float max_norm,
                      const float* norm_scratch, void* twin, immtsf_stream_t stream);

/* clip + Adam as launches the CALLER places (ABI 5; immtsf.train.FlagStep puts them at the head of the NEXT step's graph, the update
 * of every parameter range on the branch that reads those parameters first):
 *   adam_prepare: 1024 partial sums of squares of the whole gradient -- fp32 `grad`, or, when `grad_h` is given, its bf16 wire image
 *     (what a bf16 all-reduce left behind: no widening pass) -- into norm_scratch; the step DECISION: *skip_out = 1 (drop: no update,
 *     step not counted) when *pending == 0 (no gradient waits: the first replay, or behind a flush), when *err != 0 (a device-flag
 *     wait of this rank timed out) or when the guard slot is non-zero (guard_h: one bf16 value, guard_f: one float -- the time-out
 *     words of ALL ranks summed by the step's last collective, so every rank drops the same step); *pending is cleared;
 *     *step_dev += 1 unless dropped; *dropout_step_dev += 1 always.  pending / err / guard_* / step_dev / dropout_step_dev may be NULL.
 *   adam_range: the update of elements [lo, hi) of the flat buffers (lo a multiple of 8), clipped by the norm adam_prepare left in
 *     norm_scratch; the gradient is read from grad_h (bf16) when given, else from grad; zero_grad != 0 leaves grad[lo, hi) zero;
 *     the registered bf16 twin of `param` is kept current; *skip != 0: only the zero-fill happens. */
int immtsf_adam_prepare(const float* grad, const void* grad_h, uint64_t n, float* norm_scratch, int64_t* step_dev,
                        uint64_t* dropout_step_dev, int32_t* pending, const int32_t* err, const void* guard_h, const float* guard_f,
                        int32_t* skip_out, immtsf_stream_t stream);
int immtsf_adam_range(float* param, float* grad, const void* grad_h, float* exp_avg, float* exp_avg_sq, uint64_t n, uint64_t lo,
                      uint64_t hi, float lr, float beta1, float beta2, float eps, float weight_decay, const int64_t* step_dev,
                      float max_norm, const float* norm_scratch, int32_t zero_grad, const int32_t* skip, immtsf_stream_t stream);
/* *slot = (*err != 0) in the gradient wire's element type (is_bf16 != 0: one bf16, else one float): the guard word of this rank,
 * written into the slot behind the last bucket so that the step's last all-reduce sums it over the ranks (see adam_prepare) */
int immtsf_guard_pack(const int32_t* err, void* slot, int32_t is_bf16, immtsf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * One post-norm transformer encoder layer over short sequences: nn.TransformerEncoderLayer(d_model = D, nhead = H,
 * dim_feedforward = F, activation relu, batch_first) as tPatchGNN applies it to the M patches of every variable
 * (reference models/tPatchGNN.py:118-121 construction, :200-205 use).  x, out (Bs, S, D); S <= 8, D / H <= 64, D % 4 == 0.
 *   qkv = x in_w^T + in_b ; a = softmax(q k^T / sqrt(D/H)) v ; x1 = LN1(x + drop(a out_w^T + out_b))
 *   out = LN2(x1 + drop(drop(relu(x1 w1^T + b1)) w2^T + b2))
 * Dropout (training != 0): p_attn on the attention weights, p_drop on the three nn.Dropout sites, Philox sites
 * site_base + {0 attention, 1 dropout1, 2 dropout, 3 dropout2}.  7 launches forward, 15 backward.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct immtsf_encoder_layer_cfg {
    int32_t Bs, S, D, H, F;
    int32_t precision, training;
    float p_attn, p_drop, eps;
    uint64_t seed;
    const uint64_t* seed_step_dev; /* as in immtsf_fusion_cfg */
    uint64_t site_base;
    int32_t grads_prezeroed;       /* backward: the buffers in `grads` read zero already */
} immtsf_encoder_layer_cfg;
typedef struct immtsf_encoder_layer_params {
    float *in_w, *in_b;   /* (3D, D), (3D)  self_attn.in_proj_* */
    float *out_w, *out_b; /* (D, D), (D)    self_attn.out_proj.* */
    float *ln1_w, *ln1_b; /* (D), (D)       norm1 */
    float *w1, *b1;       /* (F, D), (F)    linear1 */
    float *w2, *b2;       /* (D, F), (D)    linear2 */
    float *ln2_w, *ln2_b; /* (D), (D)       norm2 */
} immtsf_encoder_layer_params;
size_t immtsf_encoder_layer_workspace_bytes(const immtsf_encoder_layer_cfg* cfg);
size_t immtsf_encoder_layer_scratch_bytes(const immtsf_encoder_layer_cfg* cfg);
/* the forward workspace holds what backward needs (qkv, a, x1, h with its dropout applied, the LayerNorm statistics) */
int immtsf_encoder_layer_forward(const immtsf_encoder_layer_cfg* cfg, const immtsf_encoder_layer_params* p, const float* x, float* out,
                                 void* workspace, size_t workspace_bytes, immtsf_stream_t stream);
/* dout (Bs, S, D) -> dx and every parameter gradient (`grads`: same layout as the parameters, all non-NULL) */
int immtsf_encoder_layer_backward(const immtsf_encoder_layer_cfg* cfg, const immtsf_encoder_layer_params* p, const float* x,
                                  const float* dout, float* dx, void* workspace, size_t workspace_bytes, void* scratch,
                                  size_t scratch_bytes, const immtsf_encoder_layer_params* grads, immtsf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * The two joints of the reference's post-norm EncoderLayer (layers/Transformer_EncDec.py:27-61; PatchTST, TimeLLM's
 * patch path, TimesNet's callers), each one block call per direction instead of dropout / add / LayerNorm /
 * activation kernels around the GEMMs:
 *   residual_layernorm:  out = LayerNorm(x + Dropout(branch))                                    (:52-54, the attention joint)
 *   ffn_block:           out = LayerNorm(x + Dropout(conv2(Dropout(act(conv1(x))))))            (:56-61; conv k=1 = Linear)
 * act: 1 ReLU, 2 GELU(erf) -- activation and dropout live in the first GEMM's epilogue, the activation derivative (from
 * the saved pre-activation for GELU) and the regenerated dropout mask in the epilogue of the data-gradient GEMM.
 * D % 4 == 0, D <= 1024.  Philox sites: residual_layernorm `site`; ffn_block site_base (hidden) and site_base + 1 (output).
 * ---------------------------------------------------------------------------------------------------------- */
int immtsf_residual_layernorm_forward(const float* x, const float* branch, int32_t rows, int32_t d, const float* gamma, const float* beta,
                                      float eps, int32_t training, float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev,
                                      float* xhat, float* rstd, float* out, immtsf_stream_t stream);
/* dout (read only) -> dx (gradient wrt x), dbranch (gradient wrt branch), dgamma, dbeta (written); scratch >= 64 (d + 8) floats */
int immtsf_residual_layernorm_backward(const float* dout, int32_t rows, int32_t d, const float* gamma, const float* xhat, const float* rstd,
                                       int32_t training, float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, float* dx,
                                       float* dbranch, float* dgamma, float* dbeta, float* scratch, immtsf_stream_t stream);
typedef struct immtsf_ffn_block_cfg {
    int32_t R, D, F;               /* rows, d_model, d_ff */
    int32_t act;                   /* 1 relu, 2 gelu */
    int32_t precision, training;
    float p_drop, eps;
    uint64_t seed;
    const uint64_t* seed_step_dev;
    uint64_t site_base;
    int32_t grads_prezeroed;       /* bit 0: the gradient buffers are zero already.  Backward, bf16-in-HBM path only (else IMMTSF_EUNSUPPORTED,
                                      nothing launched): bit 1 = the data path alone (everything but the two weight-gradient products),
                                      bit 2 = the two weight-gradient products alone, from the images the bit-1 call left in workspace /
                                      scratch, on any stream ordered behind it.  (ABI 6) */
} immtsf_ffn_block_cfg;
typedef struct immtsf_ffn_block_params {
    float *w1, *b1;     /* (F, D), (F)   conv1 (kernel size 1) */
    float *w2, *b2;     /* (D, F), (D)   conv2 */
    float *ln_w, *ln_b; /* (D), (D)      norm2 */
} immtsf_ffn_block_params;
size_t immtsf_ffn_block_workspace_bytes(const immtsf_ffn_block_cfg* cfg);
size_t immtsf_ffn_block_scratch_bytes(const immtsf_ffn_block_cfg* cfg);
int immtsf_ffn_block_forward(const immtsf_ffn_block_cfg* cfg, const immtsf_ffn_block_params* p, const float* x, float* out, void* workspace,
                             size_t workspace_bytes, immtsf_stream_t stream);
int immtsf_ffn_block_backward(const immtsf_ffn_block_cfg* cfg, const immtsf_ffn_block_params* p, const float* x, const float* dout, float* dx,
                              void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                              const immtsf_ffn_block_params* grads, immtsf_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * TimesNet's Inception block (reference layers/Conv_Blocks.py:5-31, used by models/TimesNet.py:21-68): the mean of n
 * same-padded 2-D convolutions with kernel sizes 1, 3, ..., 2n-1 is ONE convolution with the averaged, zero-padded
 * kernel.  merge builds that kernel as a GEMM weight W_eff (Cout, KS*KS*Cin), KS = 2n-1, columns ordered (dy, dx, ci);
 * unmerge is its backward (each W_i receives the centre crop of dW_eff / n, each b_i db_eff / n; written).  W / b / dW /
 * db: HOST arrays of n device pointers, W_i (Cout, Cin, 2i+1, 2i+1) as nn.Conv2d stores it.
 * conv2d_same_cl: the merged convolution on channels-last images (B, H, W, C) -- TimesBlock's (B, length, d_model)
 * activations viewed as (B, length/period, period, d_model) -- as im2col + MFMA GEMM (bias / GELU epilogue); backward:
 * the same convolution of dz with the flipped kernel for dx, GEMM with bias-gradient reduction for dW_eff / db_eff.
 * act: 0 none, 2 GELU(erf).
 * ---------------------------------------------------------------------------------------------------------- */
#define IMMTSF_INCEPTION_MAX 8
int immtsf_inception_merge(int32_t n, int32_t Cin, int32_t Cout, const float* const* W, const float* const* b, float* W_eff, float* b_eff,
                           immtsf_stream_t stream);
int immtsf_inception_unmerge(int32_t n, int32_t Cin, int32_t Cout, const float* dW_eff, const float* db_eff, float* const* dW, float* const* db,
                             immtsf_stream_t stream);
/* col: B*H*W * KS*KS*Cin floats (kept for backward); z_pre: B*H*W*Cout floats, required when act == 2 (kept for backward) */
int immtsf_conv2d_same_cl_forward(int32_t precision, const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t KS,
                                  const float* W_eff, const float* b_eff, int32_t Cout, int32_t act, float* col, float* z_pre, float* y,
                                  immtsf_stream_t stream);
/* dx may be NULL (then only the weight gradients are formed); dW_eff, db_eff are written; dx = the same im2col + GEMM convolution
 * applied to dz with the flipped kernel.  scratch: immtsf_conv2d_same_cl_scratch_floats(...) floats */
size_t immtsf_conv2d_same_cl_scratch_floats(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t KS, int32_t Cout);
int immtsf_conv2d_same_cl_backward(int32_t precision, const float* col, const float* z_pre, const float* y, const float* dy, int32_t B,
                                   int32_t H, int32_t W, int32_t Cin, int32_t KS, const float* W_eff, int32_t Cout, int32_t act, float* dx,
                                   float* dW_eff, float* db_eff, float* scratch, immtsf_stream_t stream);

/* The same convolution on TimesNet's period images with the period ON THE DEVICE (reference models/TimesNet.py:9-18 reads the top-k
 * periods on the host, :44-62 shapes the images by them: two host syncs per step, and a step that cannot be replayed from a hipGraph).
 * Images are position-major: row l * B + b of a (Lmax * B, C) matrix is position l of window b (rows beyond the series: zero), so an
 * image of any length is a prefix of the buffer.  period_rows: top (k int64 device values: the selected frequency indices) -> period[j] =
 * total / top[j], rows[j] = B * (total rounded up to a multiple of period[j]).  conv2d_period_*: conv2d_same_cl_* on the image
 * (rows[0] / B / period[0]) x period[0] that the first rows[0] rows hold; `period` / `rows`: ONE device int32 each; rows beyond rows[0]
 * are neither read nor written (dx rows beyond it are left as they are).  Lmax >= the largest possible length (2 * total).
 * w16 (Cout * KS*KS*Cin bf16, may be NULL): with precision 1 and channel counts that are multiples of 8 the im2col image is written as
 * bf16 (`col` then holds R * K bf16 values) and the products run on the bf16-in-HBM kernels; forward and backward get the same choice;
 * w16_ready != 0: w16 already holds W_eff's bf16 image (a caller that convolves several period images with one kernel casts it once). */
int immtsf_period_rows(const int64_t* top, int32_t k, int32_t total, int32_t B, int32_t* period, int32_t* rows, immtsf_stream_t stream);
int immtsf_conv2d_period_forward(int32_t precision, const float* x, int32_t B, int32_t Lmax, const int32_t* period, const int32_t* rows, int32_t Cin,
                                 int32_t KS, const float* W_eff, const float* b_eff, int32_t Cout, int32_t act, float* col, float* z_pre, float* y,
                                 void* w16, int32_t w16_ready, immtsf_stream_t stream);
size_t immtsf_conv2d_period_scratch_floats(int32_t B, int32_t Lmax, int32_t Cin, int32_t KS, int32_t Cout);
int immtsf_conv2d_period_backward(int32_t precision, const float* col, const float* z_pre, const float* dy, int32_t B, int32_t Lmax,
                                  const int32_t* period, const int32_t* rows, int32_t Cin, int32_t KS, const float* W_eff, int32_t Cout, int32_t act,
                                  float* dx, float* dW_eff, float* db_eff, float* scratch, void* w16, immtsf_stream_t stream);
/* k period images in ONE call (TimesBlock: the top-k periods share one merged kernel; reference models/TimesNet.py:62-79 loops over them).
 * x: one input shared by every image (x_stride == 0) or k inputs x + j x_stride floats; period / rows: k device numbers each; y, z_pre:
 * (k, B*Lmax, Cout); col: k im2col images (bf16 in bf16 mode, else fp32).  In bf16 mode the k images ride along grid.z of ONE im2col and ONE
 * product launch (bias + GELU in its epilogue); otherwise the call is the k single-image calls.  k <= 16.  Backward: dx is (B*Lmax, Cin),
 * summed over the images, when dx_shared, else (k, B*Lmax, Cin); dW_eff / db_eff receive the SUM over the images by accumulation -- the
 * caller hands them in ZEROED.  (ABI 6) */
int immtsf_conv2d_periods_forward(int32_t precision, const float* x, int64_t x_stride, int32_t B, int32_t Lmax, int32_t k, const int32_t* period,
                                  const int32_t* rows, int32_t Cin, int32_t KS, const float* W_eff, const float* b_eff, int32_t Cout, int32_t act,
                                  float* col, float* z_pre, float* y, void* w16, int32_t w16_ready, immtsf_stream_t stream);
size_t immtsf_conv2d_periods_scratch_floats(int32_t B, int32_t Lmax, int32_t k, int32_t Cin, int32_t KS, int32_t Cout);
int immtsf_conv2d_periods_backward(int32_t precision, const float* col, const float* z_pre, const float* dy, int32_t B, int32_t Lmax, int32_t k,
                                   const int32_t* period, const int32_t* rows, int32_t Cin, int32_t KS, const float* W_eff, int32_t Cout,
                                   int32_t act, float* dx, int32_t dx_shared, float* dW_eff, float* db_eff, float* scratch, void* w16,
                                   immtsf_stream_t stream);
/* the IMPLICIT form of the batched period convolution (bf16 mode; channel counts multiples of 8 up to 64, Lmax <= 128:
 * immtsf_conv2d_periods_implicit_ok): a period image is small enough for a workgroup to hold in LDS, so the rows of the im2col matrix are
 * formed there as MFMA operands and the image itself (161 MB bf16 per convolution at TimesNet's cfg4 shape, written and read back at HBM
 * rate) never exists -- immtsf_conv2d_periods_forward with col == NULL.  Its backward works from the INPUT x: phase bit 0 = the data
 * path (dx through the same kernel with the flipped kernel matrix), bit 1 = the kernel's gradient (an im2col image of x in scratch for
 * this one product; dW_eff / db_eff handed in ZEROED), from the images the bit-0 call left in `scratch`, on any stream ordered behind it.
 * (ABI 6) */
int immtsf_conv2d_periods_implicit_ok(int32_t precision, int32_t Lmax, int32_t Cin, int32_t KS, int32_t Cout);
size_t immtsf_conv2d_periods_backward_x_scratch_floats(int32_t B, int32_t Lmax, int32_t k, int32_t Cin, int32_t KS, int32_t Cout);
int immtsf_conv2d_periods_backward_x(int32_t precision, const float* x, int64_t x_stride, const float* z_pre, const float* dy, int32_t B, int32_t Lmax,
                                     int32_t k, const int32_t* period, const int32_t* rows, int32_t Cin, int32_t KS, const float* W_eff,
                                     int32_t Cout, int32_t act, float* dx, int32_t dx_shared, float* dW_eff, float* db_eff, float* scratch,
                                     int32_t phase, immtsf_stream_t stream);
/* TimesBlock's adaptive aggregation on the position-major images (reference models/TimesNet.py:80-86: stack the cropped images, weight them
 * with softmax(amplitude), sum, add the residual): out[b, t, :] = x[b, t, :] + sum_j w[b, j] Y[j, t B + b, :], t < total; Y (k, Lmax*B, N), w
 * (B, k) the softmax weights, x / out (B, total, N).  Backward: dY (k, Lmax*B, N) -- zero on the rows the crop dropped --, dw (B, k); the
 * residual's gradient is dout itself.  (ABI 6) */
int immtsf_period_aggregate_forward(const float* Y, const float* w, const float* x, int32_t B, int32_t total, int32_t Lmax, int32_t N, int32_t k,
                                    float* out, immtsf_stream_t stream);
int immtsf_period_aggregate_backward(const float* Y, const float* w, const float* dout, int32_t B, int32_t total, int32_t Lmax, int32_t N, int32_t k,
                                     float* dY, float* dw, immtsf_stream_t stream);

/* ---- measurement aid (bench.py roofline leg): when enabled, every GEMM launch is bracketed by hipEvents on the
 * stream it is launched on.  collect() synchronises those events and fills HOST arrays meta[10*max] = (layout,
 * precision, M, N, K, nprob, nbatch, dyn, grid threads, kernel path) and ms[max]; returns the number of records and resets
 * the tap.  Off by default; while it is on, GEMM launches from all threads are serialised by the tap's mutex. */
int immtsf_timing_enable(int32_t on);
/* Backward calls enqueue weight-gradient GEMMs on a library-owned side stream forked from / joined into `stream`
 * inside the call (concurrency with the data-gradient GEMMs; valid under hipGraph capture).  0 (the default) disables it.
 * The stream and its two events belong to the calling host thread (one set per thread and device). */
int immtsf_set_side_stream(int32_t on);

/* Device-side flags between two streams of one captured step (csrc/sync.hip; immtsf.train.FlagStep): flag_set behind the
 * producer's last kernel, flag_wait (a one-lane spin with a timeout, *err = 1 when it gave up) in front of the consumer's first,
 * flags_clear (n <= 64 consecutive flags) once both are done.  No reference counterpart: launch-model plumbing. */
int immtsf_flag_set(int32_t* flag, immtsf_stream_t stream);
int immtsf_flag_wait(int32_t* flag, int32_t* err, int32_t timeout_ms, immtsf_stream_t stream);
int immtsf_flags_clear(int32_t* flags, int32_t n, immtsf_stream_t stream);
/* flags_clear, and *set_flag = 1 in the same launch (the end of a captured step: clear the hand-over flags, mark a gradient pending) */
int immtsf_flags_clear_set(int32_t* flags, int32_t n, int32_t* set_flag, immtsf_stream_t stream);
/* Counting form, for a consumer that is NOT part of the captured step (the communication stream of the data-parallel step, enqueued
 * eagerly beside the replaying graph): flag_bump adds 1 (release) behind the producer's last kernel on every replay; flag_wait_ge
 * spins until *flag - target >= 0 (the consumer of replay k passes k).  Never cleared: a late consumer cannot miss a hand-over. */
int immtsf_flag_bump(int32_t* flag, immtsf_stream_t stream);
int immtsf_flag_wait_ge(int32_t* flag, int32_t target, int32_t* err, int32_t timeout_ms, immtsf_stream_t stream);
/* flag_wait_ge on n (1..4) flags in ONE launch (a collective over several buckets that complete together; flags: HOST array of device
 * pointers), followed -- guard_slot != NULL -- by immtsf_guard_pack(err, guard_slot, is_bf16): the wait in front of a step's LAST collective */
int immtsf_flag_wait_ge_multi(int32_t n, int32_t* const* flags, int32_t target, int32_t* err, int32_t timeout_ms, void* guard_slot,
                              int32_t is_bf16, immtsf_stream_t stream);
/* Trace of the flag kernels (a diagnostic: who waited for whom inside a replayed step, on the device's 100 MHz wall clock, without a
 * profiler serialising the branches).  immtsf_flag_trace(1) empties the ring (1024 entries) and starts recording, (0) stops;
 * immtsf_flag_trace_read copies up to max_entries entries of three int64 -- flag address, kind (0 set, 1 wait entered, 2 wait left,
 * 3 clear), wall clock -- and returns their count (< 0: error).  Both synchronise the device. */
int immtsf_flag_trace(int32_t enable);
int immtsf_flag_trace_read(int64_t* out, int32_t max_entries);
int immtsf_side_stream_enabled(void);
/* tuning aid for tools/gemm_bench.py: force a GEMM tile variant (1..6) and/or split-K factor; 0 = heuristic */
int immtsf_debug_gemm_config(int32_t variant, int32_t splitk);
/* members_host (optional, int32[24 * max]): for a GROUPED weight-gradient launch -- recorded as one record with layout code 3, kernel
 * path 3 and nprob = the number of member products -- the members' (M, N, K, K-is-a-device-value) quadruples, zero otherwise. */
int immtsf_timing_collect(int32_t max, int32_t* meta_host, float* ms_host, int32_t* members_host);

/* ---- a12 / a13: the time-aware patch embedding and the data embedding as one gather + skinny product + positional
 * table + dropout kernel per direction (csrc/embed.hip), exact fp32.
 *   mode 0 = PatchEmbedding.forward (layers/Embed.py:165-190): x (R, L) rows, R = B * n_vars;
 *            out[r,p,:] = drop(sum_k W[:,k] x[r, min(p*stride + k, L-1)] + pe[p,:]), W (D, K = patch_len), P patches.
 *   mode 1 = TokenEmbedding + PositionalEmbedding of DataEmbedding (layers/Embed.py:29-42,109-126): x (R = B, L, c_in);
 *            out[b,l,:] = drop(sum_{c,t} W[:,c,t] x[b,(l+t-1) mod L,c] + pe[l,:]), W (D, c_in, 3) = (D, K = 3 c_in), P = L.
 * pe: the (>= P, D) sinusoid table.  Dropout: Philox keyed by (seed [+ *seed_step_dev], site, element index), p_drop = 0
 * off.  backward: dW (D, K) accumulates by atomics onto zeros (dw_prezeroed != 0: the caller zero-filled it; else the
 * call does); dx (shape of x) is optional (NULL: the series are data).  IMMTSF_EUNSUPPORTED: K > 64 or D > 512. */
int immtsf_embed_forward(int32_t mode, const float* x, int32_t R, int32_t L, int32_t c_in, int32_t P, int32_t K, int32_t stride,
                         int32_t D, const float* W, const float* pe, float* out, float p_drop, uint64_t seed, uint64_t site,
                         const uint64_t* seed_step_dev, immtsf_stream_t stream);
int immtsf_embed_backward(int32_t mode, const float* x, int32_t R, int32_t L, int32_t c_in, int32_t P, int32_t K, int32_t stride,
                          int32_t D, const float* W, const float* dout, float* dW, int32_t dw_prezeroed, float* dx, float p_drop,
                          uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, immtsf_stream_t stream);

/* The bf16-in-memory GEMM the bf16 mode runs its projections on (csrc/gemm2.hip): A and B are bf16 in HBM and reach
 * LDS by LDS-DMA through a multi-stage ring; results as fp32 (C, may be NULL) and/or bf16 (Ch, may be NULL; row pitch
 * ldch).  layout 0 NT: C = A(M,K) B(N,K)^T (a linear layer's forward, layers/<module>.py nn.Linear call sites and
 * fusions/<module>.py projections); 1 NN: C = A(M,K) B(K,N) (its data gradient); 2 TN: C = A(K,M)^T B(K,N) (its weight
 * gradient; bias_grad (M) = column sums of A when given).  dyn: optional device int32 overriding M (dyn_which 0, NT/NN)
 * or K (dyn_which 1, TN) -- the ragged note count; a_rowmap: optional source row per logical row of A (NT/NN).
 * IMMTSF_EUNSUPPORTED: operands not 16-byte aligned / leading dimensions not multiples of 8 / K (NT, NN) or M, N (TN)
 * not multiples of 8. */
int immtsf_gemm_bf16(int32_t layout, const void* A, int32_t lda, const void* B, int32_t ldb, float* C, int32_t ldc, void* Ch,
                     int32_t ldch, const float* bias, float* bias_grad, int32_t M, int32_t N, int32_t K, float alpha,
                     int32_t accumulate, int32_t act, const int32_t* dyn, int32_t dyn_which, const int32_t* a_rowmap,
                     immtsf_stream_t stream);
/* n (2 .. 6) TN products C_i (M_i, N_i) = A_i (K_i, M_i)^T B_i (K_i, N_i) of different shapes as ONE launch of gemm2's grouped kernel --
 * what a block's backward does with its weight gradients (csrc/gemm2.hip immtsf_launch_gemm2_group_tn).  Measurement entry: bench.py
 * re-times the step's grouped launch through it.  IMMTSF_EUNSUPPORTED: a member the grouped kernel does not take. */
int immtsf_gemm_bf16_group_tn(int32_t n, const void* const* A, const int32_t* lda, const void* const* B, const int32_t* ldb,
                              float* const* C, const int32_t* ldc, const int32_t* M, const int32_t* N, const int32_t* K,
                              immtsf_stream_t stream);
/* tuning aid for tools/gemm2_bench.py: force a tile variant, a split-K factor, the XCD tile order (-1 = heuristic) */
int immtsf_debug_gemm2_config(int32_t variant, int32_t splitk, int32_t xcd);

/* The persistent many-rows GEMM (csrc/gemm3.hip): same operands and layouts as immtsf_gemm_bf16, for M >> 256 with
 * K > 64 -- the fusion projections at >= 512 windows per GPU (fusions/TTF_T2V_XAttn.py:70-84 input_proj / KV_proj /
 * attn in-proj, fusions/MMF_XAttn_Add.py:36-47).  One 512-thread workgroup per CU walks 256 x 256 (or 128 x 256) tiles;
 * results as fp32 (C, may be NULL) and/or bf16 (Ch, may be NULL), x = rowflag(alpha * acc + bias) + add_vec, where
 * row_flag is int32 per group of row_flag_div rows (zero: the rows are zeroed before add_vec).  dyn_rows: optional device
 * int32 overriding M (NT / NN).  IMMTSF_EUNSUPPORTED (use immtsf_gemm_bf16): K <= 64, act != 0, misaligned operands. */
int immtsf_gemm3_bf16(int32_t layout, const void* A, int32_t lda, const void* B, int32_t ldb, float* C, int32_t ldc, void* Ch,
                      int32_t ldch, const float* bias, const float* add_vec, const int32_t* row_flag, int32_t row_flag_div,
                      int32_t M, int32_t N, int32_t K, float alpha, int32_t act, const int32_t* dyn_rows, immtsf_stream_t stream);
/* TN with a long reduction and few tiles -- the weight gradients dW = dY^T X at >= 256 windows per GPU (K = rows of dY >= 8192):
 * C (M, N, fp32) = alpha * A(K, M)^T B(K, N) (+ C if accumulate), bias_grad (M, may be NULL) = alpha * column sums of A.  The
 * reduction is split over the persistent workgroups, fp32 partial tiles go to `ws` (immtsf_gemm3_tn_workspace_bytes bytes, 0 =
 * the product is too small for this path), one reduce launch follows.  dyn_k: optional device int32 overriding K. */
size_t immtsf_gemm3_tn_workspace_bytes(int32_t M, int32_t N, int32_t K);
int immtsf_gemm3_tn_bf16(const void* A, int32_t lda, const void* B, int32_t ldb, float* C, int32_t ldc, float* bias_grad, int32_t M,
                         int32_t N, int32_t K, float alpha, int32_t accumulate, const int32_t* dyn_k, void* ws, size_t ws_bytes,
                         immtsf_stream_t stream);
/* tuning aid for tools/gemm3_bench.py: force the tile height (256 / 128, 0 = heuristic) and the grid (0 = 256 workgroups) */
int immtsf_debug_gemm3_config(int32_t bm, int32_t grid);

#ifdef __cplusplus
}
#endif
#endif /* IMMTSF_H */
