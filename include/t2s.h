/*
 * t2s.h -- C ABI of libt2s_hip.so: the MI355X (gfx950) implementation of the
 * T2S diffusion hot path of Bill9125/T2MS.
 *
 * The reference has no FFI layer: its boundary is a set of Python classes
 * (SURVEY.md section 8b).  The host-side mirrors of those classes live in
 * t2ms_amd/model/... and call the entry points below through ctypes; each
 * entry point cites the reference interface it replaces (paths relative to
 * the reference repo root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *     tensors are dense, row-major fp32 in exactly the reference's layout;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     no entry point synchronises the device or allocates in the launch path
 *     (so callers may capture them into a hipGraph); only *_create/_destroy
 *     allocate / free;
 *   - return value: 0 on success, negative T2S_E_* on failure, with a
 *     human-readable message available from t2s_last_error();
 *   - the caller owns every buffer it passes in; the library owns only its
 *     packed weight copies, workspaces and hipGraph handles;
 *   - threads: a handle (t2s_dit, t2s_vae, t2s_sampler) is driven by ONE thread at a
 *     time -- a t2s_sampler together with the t2s_dit it was created on (it runs in
 *     that handle's workspace).  Different handles may be driven by different threads
 *     of one process on one device (what the library serialises for them: "Threads"
 *     at t2s_sampler_run).  The handle-free entry points (t2s_ddpm_*, t2s_rf_*,
 *     t2s_philox_*, t2s_mse_ws, t2s_mse_backward, t2s_adamw_*, t2s_attn_fwd*,
 *     t2s_time_embedding_freqs, t2s_eval_*, t2s_ts2vec_encode, t2s_mlp_*) keep no state between
 *     calls and may be called from any thread on any stream; t2s_mse lends a scratch
 *     per (device, stream), see there.  The deployment model is one process per GPU.
 */
#ifndef T2S_H
#define T2S_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2S_OK 0
#define T2S_E_INVALID (-1)  /* bad argument (shape, NULL, range)          */
#define T2S_E_HIP (-2)      /* a HIP runtime call failed                  */
#define T2S_E_STATE (-3)    /* handle not in a state that allows the call */

#define T2S_D_MODEL 128
#define T2S_N_TOK 480  /* (30/2) * (64/2) patches, transformer.py:132-136 */
#define T2S_N_HEADS 4
#define T2S_HEAD_DIM 32
#define T2S_N_BLOCKS 4
#define T2S_LAT_C 64
#define T2S_LAT_W 30
#define T2S_LAT_ELEMS (T2S_LAT_C * T2S_LAT_W)

const char* t2s_last_error(void);
/* library build id + the gfx arch it was compiled for, e.g. "t2s 0.1 gfx950" (host string) */
const char* t2s_version(void);

/* ------------------------------------------------------------------------ *
 * DiT denoiser: model/denoiser/transformer.py:128-193 (Transformer),
 * :94-124 (Transformerlayer), timm 1.0.11 Attention/Mlp call sites :104-105.
 * ------------------------------------------------------------------------ */
typedef struct t2s_dit t2s_dit;

/* Device pointers to the reference state-dict tensors. */
typedef struct t2s_dit_block_weights {
    const float* qkv_w;  /* layers.i.attn.qkv.weight            (384,128) */
    const float* qkv_b;  /* layers.i.attn.qkv.bias              (384)     */
    const float* proj_w; /* layers.i.attn.proj.weight           (128,128) */
    const float* proj_b; /* layers.i.attn.proj.bias             (128)     */
    const float* fc1_w;  /* layers.i.mlp.fc1.weight             (256,128) */
    const float* fc1_b;  /* layers.i.mlp.fc1.bias               (256)     */
    const float* fc2_w;  /* layers.i.mlp.fc2.weight             (128,256) */
    const float* fc2_b;  /* layers.i.mlp.fc2.bias               (128)     */
    const float* ada_w;  /* layers.i.adaLN_modulation.1.weight  (768,128) */
    const float* ada_b;  /* layers.i.adaLN_modulation.1.bias    (768)     */
} t2s_dit_block_weights;

typedef struct t2s_dit_weights {
    const float* conv_w;    /* conv.weight                (4,1,2,2) */
    const float* conv_b;    /* conv.bias                  (4)       */
    const float* patch_w;   /* patch_emb.weight           (128,4)   */
    const float* patch_b;   /* patch_emb.bias             (128)     */
    const float* pos_embed; /* pos_embed                  (1,480,128) */
    const float* ln_w;      /* ln.weight                  (128)     */
    const float* ln_b;      /* ln.bias                    (128)     */
    const float* out_w;     /* linear_emb_to_patch.weight (4,128)   */
    const float* out_b;     /* linear_emb_to_patch.bias   (4)       */
    const float* time_freqs; /* 10000^linspace(0,1,64), transformer.py:34 (64) */
    t2s_dit_block_weights blk[T2S_N_BLOCKS];
} t2s_dit_weights;

/* Create a denoiser able to run up to `max_seqs` sequences per call (a CFG
 * sampling step over a batch of B series uses 2*B sequences).  Packs the
 * weights into the kernels' MFMA-fragment layout (synchronous). */
int t2s_dit_create(const t2s_dit_weights* w, int max_seqs, t2s_dit** out);
/* t2s_dit_weights is T2S_DIT_N_TENSORS device pointers: 10 top-level fields in declaration order, then the 10 fields of each
 * of the 4 blocks.  The ABI carries no sizes, so: n_floats[i] (HOST array, n_entries == T2S_DIT_N_TENSORS, or NULL) = the
 * number of floats the caller holds behind pointer i; the call returns T2S_E_INVALID -- naming the tensor by its state-dict
 * key -- when one is NULL, holds fewer floats than the kernels read (the shapes in the struct comments above), or lies in a
 * device allocation that ends before those floats do (hipMemGetAddressRange; pointers HIP cannot place are passed through).
 * t2s_dit_create and t2s_dit_update_weights run the allocation-extent part themselves: an undersized buffer is an error
 * code, never a memory fault.  The host mirror calls this with the numel() of every parameter before it hands a new struct over. */
#define T2S_DIT_N_TENSORS (10 + 10 * T2S_N_BLOCKS)
int t2s_dit_weights_check(const t2s_dit_weights* w, const uint64_t* n_floats, int n_entries);
/* Re-pack after the caller changed the weights (load_state_dict, optimizer step). */
int t2s_dit_update_weights(t2s_dit* h, const t2s_dit_weights* w, void* stream);
void t2s_dit_destroy(t2s_dit* h);
int t2s_dit_max_seqs(const t2s_dit* h);
/* Matrix arithmetic of t2s_dit_forward[_cfg] / the sampler (attention and the row chain; the adaLN
 * linear and the small layers always run in f32).
 *   T2S_MATH_F32 (a new handle's mode)  v_mfma_f32_32x32x2_f32: exact fp32 multiply-add chains.
 *   T2S_MATH_BF16X3         every fp32 operand split into three bf16 terms, each product evaluated as
 *                           the six bf16 MFMAs of weight >= 2^-16 with fp32 accumulation: the same
 *                           accuracy as an fp32 product (dropped terms <= 3 * 2^-24 relative; measured
 *                           against fp64 the attention kernel is as close as the f32 one) at 2.67x
 *                           fewer matrix cycles.  Allocates 2 x max_seqs x 368,640 B + 3.1 MB of split
 *                           weights on first use; weights follow t2s_dit_update_weights.
 *                           Since round 5 the host-side Sampler and infer.py SELECT this mode unless told otherwise
 *                           (its error against fp64 is not larger than the reference's own CPU fp32 arithmetic at any of
 *                           17 table entries, profiles/r05_accuracy.md); the class API and the bench headline stay on F32.
 * Not capturable; a hipGraph captured under one mode keeps replaying that mode's kernels. */
#define T2S_MATH_F32 0
#define T2S_MATH_BF16X3 1
int t2s_dit_set_math(t2s_dit* h, int math);

/* TimeEmbedding.forward, transformer.py:30-40.  t: (B) fp32 (int64 timesteps are
 * converted to fp32 by the host mirror exactly as `t * 100.0` promotes them);
 * out: (B,128) = [sin(100 t / f) | cos(100 t / f)]. */
int t2s_time_embedding(const t2s_dit* h, const float* t, float* out, int B, void* stream);
/* The same without a handle (the module TimeEmbedding is parameter-free, transformer.py:25-40): freqs (64) =
 * 10000^linspace(0,1,64) as the caller evaluated it (the host mirror uses the reference's own fp32 torch ops, :34). */
int t2s_time_embedding_freqs(const float* freqs, const float* t, float* out, int B, void* stream);

/* Transformer.forward, transformer.py:158-193.
 *   x    (B,64,30)   latent
 *   temb (temb_rows,128), temb_rows in {1,B}: output of t2s_time_embedding
 *   text (B,128) or NULL (text_input=None -> unconditional)
 *   out  (B,64,30)
 * B <= max_seqs. */
int t2s_dit_forward(t2s_dit* h, const float* x, const float* temb, int temb_rows,
                    const float* text, float* out, int B, void* stream);

/* Both classifier-free-guidance branches of one sampling step in ONE pass
 * (infer.py:79-80 / 85-86): sequences [0,B) are the unconditional branch
 * (c = temb), [B,2B) the conditional one (c = temb + text); both read x[b].
 *   temb (1,128); out_uncond, out_cond (B,64,30).  2*B <= max_seqs. */
int t2s_dit_forward_cfg(t2s_dit* h, const float* x, const float* temb, const float* text,
                        float* out_uncond, float* out_cond, int B, void* stream);
/* The same pass with one time-embedding row PER series (temb_rows == B; 1 = shared): what the pair of calls
 * `model(x_t, t, None)`, `model(x_t, t, emb)` at infer.py:79-80 / 85-86 computes for a per-row t.  The class-API mirror
 * runs such a pair as ONE pass once it has seen the pattern (t2ms_amd/model/denoiser/transformer.py) -- speculation on
 * torch tensor identity, which cannot see a raw-pointer write: a caller that updates x_t IN PLACE through this ABI
 * (t2s_ddpm_step, t2s_rf_step) between the two calls of a mirror Transformer switches the pairing off
 * (Transformer.set_pairing(False), or T2S_NO_PAIRING=1 in the environment). */
int t2s_dit_forward_cfg_rows(t2s_dit* h, const float* x, const float* temb, int temb_rows, const float* text,
                             float* out_uncond, float* out_cond, int B, void* stream);

/* In-situ kernel timing with HIP events recorded on the launching stream around every launch of
 * the forward (not usable while the stream is being captured).  _begin arms it; after running
 * forwards, _end synchronises and returns {attention ms, attention launches, row-chain ms,
 * row-chain launches, other ms, other launches} in out6 and disarms it. */
int t2s_dit_timing_begin(t2s_dit* h);
int t2s_dit_timing_end(t2s_dit* h, double* out6);
/* The same with the launches of the training step (t2s_dit_train_forward / _backward) in classes 3..8: out holds
 * n_classes pairs {ms, launches}: 0 attention, 1 row chain, 2 other (inference forward); 3 streaming GEMMs / fused row
 * kernels, 4 attention forward, 5 attention backward, 6 weight gradients, 7 gate / LayerNorm elementwise kernels,
 * 8 everything else of the step (packs, patchify, final layer, adaLN linear); 9 / 10 the first (block 0's qkv with the
 * patchify prologue) and the last (block 3's proj + MLP with the final layer) row-chain launch of an inference forward,
 * which class 1 contains as well.  n_classes <= 11. */
int t2s_dit_timing_end_ex(t2s_dit* h, double* out, int n_classes);

/* Test tap: copy out the residual stream (S,480,128) the last forward left in the
 * workspace (post block 3, before the final LayerNorm).  Used by tests to bisect. */
int t2s_dit_read_stream(const t2s_dit* h, float* out, int S, void* stream);

/* Fused softmax(q k^T / sqrt(32)) v for N=480, head_dim=32 (timm Attention.forward
 * core; call site transformer.py:116).  q,k,v: (BH,480,32); o: (BH,480,32). */
int t2s_attn_fwd(const float* q, const float* k, const float* v, float* o, int BH, void* stream);
/* The same attention on the library's internal "fragment-major" tensors -- the kernel the DiT
 * forward actually launches (exposed for benchmarking / tests).  With tile = row/32, i = row%32,
 * a (480,32) per-head matrix X is stored as float4 fragments
 *     P[((tile*4 + g)*64 + 32*h + i)*4 + e] = X[32*tile + i][8*g + 4*h + e]
 * q, k: P of the head's Q and K;  vT: P of V TRANSPOSED, i.e.
 *     vT[((tile*4 + g)*64 + 32*h + d)*4 + e] = V[32*tile + 8*g + 4*h + e][d];
 * heads ordered (seq*4 + head).  o: (n_seq*480, 128) with
 *     o[(((row/32)*16 + G)*64 + 32*h + row%32)*4 + e] = O[row][8*G + 4*h + e]  (G = col/8). */
int t2s_attn_fwd_packed(const float* q, const float* k, const float* vT, float* o, int n_seq,
                        void* stream);

/* The "bf16x3" attention kernel (T2S_MATH_BF16X3, see t2s_dit_set_math) on plain tensors: same
 * contract as t2s_attn_fwd (q, k, v, o: (BH, 480, 32) fp32, BH a multiple of 4), fp32-accurate
 * products evaluated as six bf16 MFMAs each.  Packs / unpacks its operands internally and
 * synchronises the stream (tests, benchmarking). */
int t2s_attn_fwd_x3(const float* q, const float* k, const float* v, float* o, int BH, void* stream);

/* ------------------------------------------------------------------------ *
 * Training step of the DiT: train.py:101-127 (pred = model(x_t, t, emb); loss.backward();
 * optimizer.step()).  The host mirror wraps these in a torch.autograd.Function.
 * ------------------------------------------------------------------------ */
typedef struct t2s_dit_block_grads { /* same shapes as t2s_dit_block_weights */
    float *qkv_w, *qkv_b, *proj_w, *proj_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ada_w, *ada_b;
} t2s_dit_block_grads;
typedef struct t2s_dit_grads { /* every tensor that receives a gradient (925,592 values) */
    float *conv_w, *conv_b, *patch_w, *patch_b, *ln_w, *ln_b, *out_w, *out_b;
    t2s_dit_block_grads blk[T2S_N_BLOCKS];
} t2s_dit_grads;

/* Arithmetic of the training path.  T2S_TRAIN_F32 (default): fp32 end to end, gradients equal
 * autograd through the fp32 reference to ~1e-6 relative.  T2S_TRAIN_BF16 (BASELINE config 4,
 * "train.py bf16"): every contraction on bf16 MFMA with fp32 accumulation, saved activations in
 * bf16; master weights, residual stream, LayerNorm / softmax statistics, gradients and the
 * optimizer stay fp32.  Takes effect at the next t2s_dit_train_forward. */
#define T2S_TRAIN_F32 0
#define T2S_TRAIN_BF16 1
int t2s_dit_set_train_dtype(t2s_dit* h, int dtype);
/* Transformer.forward that keeps the activations for a following backward (plain layouts).
 * `w` are the caller's CURRENT weights (call t2s_dit_update_weights first if they changed since
 * the handle was created/updated); arguments otherwise as t2s_dit_forward.  Allocates its
 * workspace on first use (not capturable). */
int t2s_dit_train_forward(t2s_dit* h, const t2s_dit_weights* w, const float* x, const float* temb,
                          int temb_rows, const float* text, float* out, int B, void* stream);
/* The attention forward of the T2S_TRAIN_BF16 path on fp32 buffers (converted to bf16 on the way
 * in, back on the way out; synchronises the stream): q, k, v (n_seq*4, 480, 32) heads ->
 * o_rows (n_seq*480, 128) token rows, lse (n_seq*4, 480) = log2-domain log-sum-exp of the scaled
 * scores.  For tests / benchmarking of that kernel (incl. its stale-reference branch). */
int t2s_attn_fwd_bf16(const float* q, const float* k, const float* v, float* o_rows, float* lse,
                      int n_seq, void* stream);
/* Backward of the last t2s_dit_train_forward: dout (B,64,30) = dLoss/dout; writes (overwrites)
 * every gradient tensor of `g`.  Every gradient is reduced in a fixed order (bit-reproducible from run to run): the block
 * weights / biases per row slab in slab order, the small final-layer and patchify gradients (ln, linear_emb_to_patch,
 * patch_emb, conv) per workgroup in workgroup order. */
int t2s_dit_train_backward(t2s_dit* h, const float* dout, const t2s_dit_grads* g, int B, void* stream);
/* The gradient with respect to the denoiser's INPUT latent, dinput (B,64,30) (patchify backwards, transformer.py:166-172), after
 * t2s_dit_train_backward of the same batch.  train.py needs it only when something upstream of the latent trains: the LA-VAE
 * encoder un-frozen (train.py:31-33, `usepretrainedvae` false). */
int t2s_dit_train_input_grad(t2s_dit* h, float* dinput, int B, void* stream);
/* One fused AdamW update (torch.optim.AdamW semantics; train.py:37 uses lr 1e-4, weight_decay 0):
 * p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).  step >= 1. */
int t2s_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   void* stream);
/* The same update for up to 64 tensors in ONE launch (the DiT has 48 trainable tensors).  table_dev: device array
 * of n_tensors entries; total_chunks = sum over entries of ceil(n / 1024).  All entries share lr / betas / eps /
 * weight_decay / step. */
typedef struct t2s_adamw_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    uint64_t n;
} t2s_adamw_tensor;
int t2s_adamw_step_multi(const t2s_adamw_tensor* table_dev, int n_tensors, uint64_t total_chunks, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int step, void* stream);
/* Backward of t2s_mse: da = 2 (a-b) g / n, db = -da (either may be NULL); grad_out: device scalar. */
int t2s_mse_backward(const float* a, const float* b, const float* grad_out, float* da, float* db,
                     uint64_t n, void* stream);

/* ------------------------------------------------------------------------ *
 * Diffusion backbones: model/backbone/DDPM.py, model/backbone/rectified_flow.py
 * ------------------------------------------------------------------------ */
/* DDPM.p_sample (DDPM.py:28-36) fused with the CFG combine (infer.py:87):
 *   pred = u + cfg*(c-u);  x <- coef[t][0]*(x - coef[t][1]*pred) + coef[t][2]*z
 * coef: (T,3) = {1/sqrt(alpha), (1-alpha)/sqrt(1-alpha_bar), sqrt(beta)} (host-built
 * with the reference's own fp32 ops); t_index in [0,T).
 * z = noise[n] if noise != NULL, else the library's Philox N(0,1) stream
 * (seed, stream_id, row0 + row) -- see t2s_philox_normal.
 * eps_c may be NULL (then pred = eps_u and cfg is ignored).  In-place on x. */
int t2s_ddpm_step(float* x, const float* eps_u, const float* eps_c, const float* noise,
                  const float* coef, int t_index, float cfg, uint64_t seed, uint32_t stream_id,
                  uint32_t row0, int B, void* stream);
/* DDPM.p_sample as the class exposes it (DDPM.py:28-36): per-row timestep t (B) int32,
 * injected draws `noise` (B,64,30) (the host mirror supplies torch.randn like DDPM.py:35),
 * out of place.  n_steps = T, the length of the coefficient tables: a row whose t is outside [0,T) (the reference's
 * `gather` raises for it, DDPM.py:7-9) reads no table entry and comes out as NaN -- memory-safe and loud, without a
 * host sync in the sampling loop. */
int t2s_ddpm_p_sample(const float* xt, const float* eps_hat, const int32_t* t, const float* noise,
                      const float* coef, float* out, int B, int n_steps, void* stream);
/* The same two class methods on latents of another size, rows of `row_elems` floats (a multiple of 4): the MLP denoiser
 * of BASELINE configs[0] runs on the (B,64,L/4) pre-interpolation latent, not on (B,64,30). */
int t2s_ddpm_p_sample_n(const float* xt, const float* eps_hat, const int32_t* t, const float* noise,
                        const float* coef, float* out, int B, int row_elems, int n_steps, void* stream);
int t2s_ddpm_q_sample_n(const float* x0, const float* eps, const int32_t* t, const float* sqrt_ab,
                        const float* sqrt_1mab, float* out, int B, int row_elems, int n_steps, void* stream);
/* DDPM.loss / RectifiedFlow.loss = F.mse_loss(a, b) (DDPM.py:37-38, rectified_flow.py:13-16):
 * mean over n elements into out[0]; fixed summation order (deterministic, two launches: per-workgroup partials, then
 * their sum in index order).  scratch: T2S_MSE_SCRATCH_FLOATS floats of the CALLER's device memory, contents irrelevant
 * before and after; no state is kept anywhere else, so concurrent calls (other streams, other threads) only need
 * different scratch. */
#define T2S_MSE_SCRATCH_FLOATS 1024
int t2s_mse_ws(const float* a, const float* b, float* out, uint64_t n, float* scratch, void* stream);
/* The same with a scratch the library lends per (device, stream): allocated (hipMalloc, kept for the life of the
 * process) the first time a stream calls -- that first call is not capturable, later ones are.  Calls on one stream
 * are ordered by the stream; calls on different streams use different scratch. */
int t2s_mse(const float* a, const float* b, float* out, uint64_t n, void* stream);
/* RectifiedFlow.euler (rectified_flow.py:5-7) fused with the CFG combine (infer.py:81-82):
 *   x <- x + (u + cfg*(c-u)) * dt */
int t2s_rf_step(float* x, const float* v_u, const float* v_c, float cfg, float dt, int B,
                void* stream);
/* DDPM.q_sample (DDPM.py:19-27): out = sqrt_ab[t[b]]*x0 + sqrt_1mab[t[b]]*eps.
 * sqrt_ab, sqrt_1mab: (T) host-built tables, T = n_steps; t: (B) int32 (out of range -> NaN row, see p_sample). */
int t2s_ddpm_q_sample(const float* x0, const float* eps, const int32_t* t, const float* sqrt_ab,
                      const float* sqrt_1mab, float* out, int B, int n_steps, void* stream);
/* RectifiedFlow.create_flow (rectified_flow.py:8-12): out = t*x1 + (1-t)*x0, t: (B) fp32. */
int t2s_rf_create_flow(const float* x1, const float* x0, const float* t, float* out, int B,
                       void* stream);
/* N(0,1) draws, Philox4x32-10 + Box-Muller: element e of GLOBAL row r uses counter
 * (e/4, r, stream_id, 0), key = seed; identical for any sharding of rows over GPUs.
 * out: (n_rows,row_elems), row_elems % 4 == 0.  (replaces torch.randn at
 * DDPM.py:35 / infer.py:75 in perf mode; parity mode injects noise instead.) */
int t2s_philox_normal(float* out, uint64_t seed, uint32_t stream_id, uint32_t row0, int n_rows,
                      int row_elems, void* stream);
/* U[0,1) draws of the same stream: element e of GLOBAL row r = lane e % 4 of counter (e/4, r, stream_id, 0), key = seed,
 * as the 24-bit uniform (x >> 8) * 2^-24 (exact in fp32, never 1.0).  out: (n_rows,row_elems), any row_elems >= 1.
 * (replaces torch.rand at train.py:109,113: the per-row diffusion time of a training step, drawn on the device as a
 * function of (seed, step, global row) -- the same for any number of GPUs, and no host -> device copy per step.) */
int t2s_philox_uniform(float* out, uint64_t seed, uint32_t stream_id, uint32_t row0, int n_rows,
                       int row_elems, void* stream);

/* ------------------------------------------------------------------------ *
 * LA-VAE codec: model/pretrained/vqvae.py:36-105 (Encoder, Decoder)
 * ------------------------------------------------------------------------ */
typedef struct t2s_vae t2s_vae;

typedef struct t2s_vae_stack_weights { /* ResidualStack, vqvae.py:7-33 */
    const float* conv3_w[4]; /* _layers.i._block.1.weight (res_hidden,hidden,3), no bias */
    const float* conv1_w[4]; /* _layers.i._block.3.weight (hidden,res_hidden,1), no bias */
} t2s_vae_stack_weights;

typedef struct t2s_vae_weights {
    int hidden;       /* block_hidden_size (128)   */
    int res_hidden;   /* res_hidden_size (256)     */
    int n_res_layers; /* num_residual_layers (2), <= 4 */
    int emb;          /* embedding_dim (64)        */
    /* decoder.* (may all be NULL for an encode-only handle) */
    const float* dec_conv1_w; /* (hidden,emb,3) */
    const float* dec_conv1_b;
    t2s_vae_stack_weights dec_stack;
    const float* dec_ct1_w; /* ConvTranspose1d (hidden,hidden/2,4) */
    const float* dec_ct1_b;
    const float* dec_ct2_w; /* ConvTranspose1d (hidden/2,1,4) */
    const float* dec_ct2_b;
    /* encoder.* (may all be NULL for a decode-only handle) */
    const float* enc_conv1_w; /* (hidden/2,1,4) */
    const float* enc_conv1_b;
    const float* enc_conv2_w; /* (hidden,hidden/2,4) */
    const float* enc_conv2_b;
    const float* enc_conv3_w; /* (hidden,hidden,3) */
    const float* enc_conv3_b;
    t2s_vae_stack_weights enc_stack;
    const float* enc_prevq_w; /* (emb,hidden,1) */
    const float* enc_prevq_b;
} t2s_vae_weights;

int t2s_vae_create(const t2s_vae_weights* w, t2s_vae** out);
void t2s_vae_destroy(t2s_vae* h);
/* Decoder.forward, vqvae.py:97-105: z (B,64,30) -> recon (B,L), after (B,64,L/4) (may be NULL).
 * Any L % 4 == 0: up to 128 a series is one LDS-resident tile, longer ones (the reference's SUSHI set is 2048 long) run
 * as time tiles with recomputed halos -- the same values bit for bit.  (The host mirror applies torch.squeeze's rule.) */
int t2s_vae_decode(t2s_vae* h, const float* z, float* recon, float* after, int B, int L,
                   void* stream);
/* Decoder.forward on a latent of another width: z (B,64,latent_w), latent_w <= 32 (F.interpolate at vqvae.py:98 accepts any;
 * BASELINE configs[0] decodes the (B,64,L/4) latent of the MLP denoiser, for which the interpolation is the identity). */
int t2s_vae_decode_w(t2s_vae* h, const float* z, float* recon, float* after, int B, int L, int latent_w,
                     void* stream);
/* Encoder.forward, vqvae.py:57-71: x (B,L) -> z (B,64,30), before (B,64,L/4) (may be NULL for L <= 128; for longer
 * series the time tiles write `before` and a second launch interpolates it to the latent, so the buffer is required). */
int t2s_vae_encode(t2s_vae* h, const float* x, float* z, float* before, int B, int L,
                   void* stream);

/* Re-copy the weights into an existing handle (same hyper-parameters, same encoder / decoder presence): what a training
 * step does after the optimizer changed encoder.* (train.py:31-33 with usepretrainedvae false) -- no allocation, stream-ordered. */
int t2s_vae_update_weights(t2s_vae* h, const t2s_vae_weights* w, void* stream);

/* Backward of Encoder.forward (vqvae.py:57-71) for the one reference configuration that TRAINS the LA-VAE encoder (train.py:31-33,
 * `usepretrainedvae` false: the encoder is grafted into the denoiser and its parameters join the optimizer).
 *   x (B,L) the forward's input; dz (B,64,30) = dLoss/dz; dbefore (B,64,L/4) = dLoss/dbefore or NULL (train.py uses z only);
 *   g: one gradient tensor per encoder.* parameter, same shapes as t2s_vae_weights' enc_* fields; every one is OVERWRITTEN.
 * The forward is recomputed from x (nothing is saved by t2s_vae_encode); data gradients run per series in LDS, weight gradients
 * as exact-fp32 MFMA GEMMs over all (series, position) rows with a fixed two-stage reduction: bit-reproducible.
 * Default LA-VAE shape only (hidden 128, res_hidden 128 / 256, emb 64) and L <= 128; allocates its row blocks on first use /
 * when B * L grows (not capturable then). */
typedef struct t2s_vae_enc_grads {
    float *conv1_w, *conv1_b;   /* encoder._conv_1 (hidden/2,1,4), (hidden/2) */
    float *conv2_w, *conv2_b;   /* encoder._conv_2 (hidden,hidden/2,4), (hidden) */
    float *conv3_w, *conv3_b;   /* encoder._conv_3 (hidden,hidden,3), (hidden) */
    float* stack_conv3_w[4];    /* encoder._residual_stack._layers.i._block.1.weight (res_hidden,hidden,3) */
    float* stack_conv1_w[4];    /* encoder._residual_stack._layers.i._block.3.weight (hidden,res_hidden,1) */
    float *prevq_w, *prevq_b;   /* encoder._pre_vq_conv (emb,hidden,1), (emb) */
} t2s_vae_enc_grads;
int t2s_vae_encode_backward(t2s_vae* h, const float* x, const float* dz, const float* dbefore, const t2s_vae_enc_grads* g, int B,
                            int L, void* stream);

/* ------------------------------------------------------------------------ *
 * Fused sampling loop: infer.py:75-95 (x_T -> steps x [2 DiT forwards + CFG +
 * sampler update] -> LA-VAE decode), one hipGraph per step replayed `steps`
 * times with a device-side step counter.
 * ------------------------------------------------------------------------ */
typedef struct t2s_sampler t2s_sampler;

#define T2S_MODE_DDPM 0
#define T2S_MODE_RF 1

typedef struct t2s_sample_config {
    int mode;            /* T2S_MODE_DDPM | T2S_MODE_RF                                  */
    int steps;           /* infer.py --total_step                                         */
    float cfg_scale;     /* infer.py --cfg_scale                                          */
    int batch;           /* series per call on this GPU (2*batch <= dit max_seqs)         */
    int length;          /* decoded series length L (24/48/96)                            */
    int use_graph;       /* 1: capture one step into a hipGraph and replay it             */
    uint64_t seed;       /* Philox key (perf mode)                                        */
    uint32_t row0;       /* global index of this shard's first series (multi-GPU)         */
    const float* ddpm_coef; /* HOST (steps,3), see t2s_ddpm_step; NULL for RF             */
    const float* t_values;  /* HOST (steps): the t passed to the denoiser at loop index j:
                               DDPM steps-1-j (infer.py:84), RF round(j/steps*steps)/steps (:78) */
} t2s_sample_config;

int t2s_sampler_create(t2s_dit* dit, t2s_vae* vae /* may be NULL: no decode */,
                       const t2s_sample_config* cfg, t2s_sampler** out);
void t2s_sampler_destroy(t2s_sampler* s);
/* Run the whole loop.
 *   x      (B,64,30) in: x_T (perf mode: fill it with t2s_philox_normal, stream_id 0xFFFFFFFF);
 *          out: final latent
 *   text   (B,128)
 *   noise  (steps,B,64,30) injected per-step draws (parity mode) or NULL (Philox, perf mode)
 *   series (B,L) decoded output, or NULL
 *   trace0 (steps,L) or NULL: decode of row 0 after every step (infer.py:90-93)
 *   stream NULL = the default stream.  With use_graph = 1 the graphs are then captured and replayed on a stream
 *          the sampler owns (the default stream cannot be captured), ordered after everything queued on the default
 *          stream before the call and joined back to it before the call returns -- same semantics, never an eager
 *          fallback.  trace0 runs are eager by design (a decode between the steps).  Runs with more than one lane
 *          (t2s_sampler_set_lanes) execute on the library's own pool of streams in the same way for ANY `stream`: the
 *          run is ordered after everything queued on `stream` before the call and joined back to it before the call
 *          returns.
 * Threads: (the header's convention: one thread per handle at a time.)  Two threads may drive two samplers on two t2s_dit
 *          handles of one device concurrently.  Every run that opens a stream capture (use_graph) or uses the library's
 *          stream pool (several lanes, or stream NULL with use_graph) holds a per-device lock for the length of its
 *          ENQUEUE (host work of a few ms; the GPU work stays asynchronous), and t2s_sampler_create / _destroy take the
 *          same lock.  Why (measured): while ANY thread has a stream capture open on the device, HIP
 *          answers a synchronous hipMemcpy and a hipDeviceSynchronize from any other thread with an error
 *          (hipErrorStreamCaptureImplicit / hipErrorStreamCaptureUnsupported) and INVALIDATES the open capture, thread-
 *          local capture mode notwithstanding.  The library itself issues neither outside that lock (t2s_sampler_create
 *          uploads on a non-blocking stream of its own; no entry point copies synchronously), and the Python mirrors
 *          build / grow / destroy their handles under the same per-device lock (t2ms_amd._lib.device_lock).  What no lock
 *          of the library can cover is the CALLER's own device-wide calls on other threads while a run is being enqueued
 *          -- hipDeviceSynchronize (torch.cuda.synchronize()), synchronous hipMemcpy -- and a `stream` handed to a
 *          single-lane run, which -- like any HIP stream under capture -- must not be used by another thread meanwhile.
 *          (Stream-ordered work of other threads, including torch's default-stream kernels, asynchronous copies and
 *          stream synchronisations, is fine: tools/stress_threads.py runs it against open captures by the thousand.)
 *          The pool is created and calibrated by the first t2s_sampler_create on a device (not by a run), so a run
 *          never allocates or synchronises for it.
 * Memory:  t2s_sampler_create also allocates a whole-run adaLN table (steps x (batch + 1) x 3072 floats: 3.2 GB at
 *          256 series x 1000 steps, ~2 % of a step) when that is at most 1/8 of the device memory currently free and
 *          16 GB; otherwise -- or with T2S_ADALN_TABLE=0 in the environment -- the per-step adaLN kernel runs instead
 *          (bitwise the same results).
 */
int t2s_sampler_run(t2s_sampler* s, float* x, const float* text, const float* noise,
                    float* series, float* trace0, void* stream);
/* Lanes of t2s_sampler_run: the rows of a batch never interact (infer.py:76-88), so the loop may run as two (up to four)
 * part batches, each a complete chain with its own hipGraph on its own stream (lanes 1.. on streams the sampler owns,
 * forked from / joined to `stream` inside the call), so that one chain's kernels fill the chip while the other's
 * drain.  Bitwise the same result.  lanes: 0 = automatic (equal shares: two when the batch is a multiple of 64 or 32 series, three for 96; env
 * T2S_SAMPLER_LANES=<n> overrides), or 1 .. 4 chains of equal shares.  trace0 runs always use one lane. */
int t2s_sampler_set_lanes(t2s_sampler* s, int lanes);
/* What a lane's hipGraph holds (use_graph = 1): whole_loop = 1 -> the WHOLE loop, steps x (forward + update) kernel nodes,
 * launched once per run ("whole loop in one hipGraph"); 0 -> ONE step, replayed `steps` times from the
 * host; -1 -> the library's default (env T2S_SAMPLER_LOOP_GRAPH=0|1 overrides it).  Every node reads its loop index from
 * the lane's device counter, so both forms run the same kernels in the same order: bitwise the same result.  Takes effect
 * at the next t2s_sampler_run (the graphs are re-captured when the form changes). */
int t2s_sampler_set_loop_graph(t2s_sampler* s, int whole_loop);
/* Move the sampler to another shard position: global index of its first series (the Philox key of row r is
 * row0 + r).  Takes effect at the next t2s_sampler_run; the captured hipGraphs are kept (the kernels read the
 * value from device memory next to the step counter).  infer.py:66 loops over batches with one sampler. */
int t2s_sampler_set_row0(t2s_sampler* s, uint32_t row0);
/* Number of lanes the sampler currently holds an instantiated hipGraph for (0: the last run was eager / nothing run
 * yet).  Lets a caller (and tests/test_hip_parity.py) check that use_graph = 1 really replays a graph. */
int t2s_sampler_graph_lanes(const t2s_sampler* s);
/* How many MUTUALLY CONCURRENT streams the calibration of the current device's lane-stream pool found (0: the pool has
 * not been built yet -- it is built by the first multi-lane or NULL-stream graph run; 4: every lane has a hardware queue
 * of its own).  Diagnostic: HIP streams that share a hardware queue execute one after the other. */
int t2s_sampler_lane_pool(void);

/* ------------------------------------------------------------------------ *
 * Evaluation metrics: evaluation.py:166-206 (calculate_mse, calculate_wape), :21-45 (calculate_mrr)
 * ------------------------------------------------------------------------ */
/* ori, gen: (n, len) device arrays, len = L * n_series of the (N, L, n_series) arrays infer.py writes;
 * per_sample: (n, 2) output = [mse_i, wape_i (NaN when sum |ori_i| == 0)]; out: [MSE, WAPE] =
 * [mean_i mse_i, nanmean_i wape_i].  Deterministic summation order. */
int t2s_eval_mse_wape(const float* ori, const float* gen, float* per_sample, float* out, int n, int len,
                      void* stream);

/* MRR over repeated generations: evaluation.py:21-45 (calculate_mrr) with cosine_similarity of
 * Dataset_Construction_Pipeline/Evaluate_Datasets.py:6-15, as evaluation.py:298-314 feeds it (x_1.npy against the
 * x_t.npy of run_0..run_{runs-1}).
 * ori (n, len); gen (runs, n, len), run-major (the reference stacks the runs on a trailing axis; the host mirror
 * passes them in the order it loads them); sims (n, runs) output: the cosine similarities (0 where not finite);
 * score (n) output: 1 / (g + 1) for the best run g when its similarity exceeds `threshold` (the reference uses
 * 0.5, evaluation.py:300), else 0 -- g is the run INDEX, as the reference computes it; ties go to the largest
 * index; out[0] = mean score. */
int t2s_eval_mrr(const float* ori, const float* gen, float* sims, float* score, float* out, int n, int len,
                 int runs, float threshold, void* stream);

/* ED: evaluation.py:137-150 (calculate_ed).  ori, gen: (n, L, n_series); per_sample (n): mean over series of the
 * Euclidean distance over time; out[0] = mean over samples. */
int t2s_eval_ed(const float* ori, const float* gen, float* per_sample, float* out, int n, int L, int n_series,
                void* stream);
/* CRPS: evaluation.py:51-83 (calculate_crps) over the repeated generations evaluation.py:298-314 stacks.
 * ori (n, L, n_series); gen (runs, n, L, n_series) run-major; per_sample (n); out[0] = mean. */
int t2s_eval_crps(const float* ori, const float* gen, float* per_sample, float* out, int n, int L, int n_series,
                  int runs, void* stream);
/* DTW: evaluation.py:152-163 (calculate_dtw = dtaidistance 2.3 dtw_ndim.distance, no window, no penalty): sqrt of the
 * minimal warping-path cost with squared-Euclidean point costs between the (L, n_series) sequences of each sample.
 * per_sample (n); out[0] = mean.  L <= 4096. */
int t2s_eval_dtw(const float* ori, const float* gen, float* per_sample, float* out, int n, int L, int n_series,
                 void* stream);

/* TS2Vec encoder of the C-FID metric: evaluate/ts2vec.py:352-399 (TSEncoder.forward, eval mode, mask 'all_true') followed
 * by the 'full_series' pooling of TS2Vec.encode (:236-245).  Device pointers to the state-dict tensors:
 * input_fc.{weight (hidden,input_dims),bias}; block i in [0,depth]: feature_extractor.net.i.conv{1,2}.conv.{weight
 * (co,ci,3),bias} with dilation 2^i (blocks 0..depth-1: hidden -> hidden, block depth: hidden -> output_dims with the 1x1
 * projector feature_extractor.net.depth.projector.{weight (output_dims,hidden,1),bias}). */
#define T2S_TS2VEC_MAX_BLOCKS 16
typedef struct t2s_ts2vec_weights {
    int input_dims, hidden, output_dims, depth;
    const float *fc_w, *fc_b;
    const float* conv1_w[T2S_TS2VEC_MAX_BLOCKS];
    const float* conv1_b[T2S_TS2VEC_MAX_BLOCKS];
    const float* conv2_w[T2S_TS2VEC_MAX_BLOCKS];
    const float* conv2_b[T2S_TS2VEC_MAX_BLOCKS];
    const float *proj_w, *proj_b;
} t2s_ts2vec_weights;
/* x (B, T, input_dims) -> rep (B, T, output_dims) per-step representations (may be NULL) and full (B, output_dims) =
 * their maximum over time.  Time steps of x holding a NaN are zeroed as the reference does.  3 * max(hidden,
 * output_dims) * T * 4 bytes must fit in 160 KB of LDS (T <= 128 at the evaluation's 64 / 100 channels). */
int t2s_ts2vec_encode(const t2s_ts2vec_weights* w, const float* x, float* rep, float* full, int B, int T, void* stream);

/* ------------------------------------------------------------------------ *
 * MLP denoiser of BASELINE configs[0]: model/denoiser/mlp.py:49-94 (MLPlayer x 8 on a
 * (64, 6) latent; forward, and the backward for train.py --denoiser MLP).
 * ------------------------------------------------------------------------ */
#define T2S_MLP_LAYERS 8
#define T2S_MLP_WIDTH 64      /* channels, mlp.py:52-56 */
#define T2S_MLP_POSITIONS 6   /* latent width the layer is built for, mlp.py:55,67 */
#define T2S_MLP_TEXT_DIM 128
#define T2S_MLP_HIDDEN 256
#define T2S_MLP_PACKED_FLOATS 397888 /* 8 x 49,736: what t2s_mlp_pack writes */
/* Device pointers to the state-dict tensors of layers.<i> that the forward reads (torch layouts).  cross_attn.query /
 * cross_attn.key are not among them: the six keys are the same row (mlp.py:77 repeats the text over the positions), every
 * softmax row is uniform and the attended value is value(text) whatever they hold.  norm1, norm3, pos_emb, self_attn,
 * self_attn2 are constructed and never called (mlp.py:53-60). */
typedef struct t2s_mlp_layer_weights {
    const float *value_w, *value_b; /* cross_attn.value (64,128), (64) */
    const float *proj_w, *proj_b;   /* cross_attn.proj  (64,64), (64)  */
    const float *norm2_w, *norm2_b; /* (64) */
    const float *mlp0_w, *mlp0_b;   /* mlp.0  (256,64), (256) */
    const float *mlp2_w, *mlp2_b;   /* mlp.2  (64,256), (64)  */
    const float *pos0_w, *pos0_b;   /* mlp2.0 (256,6), (256)  */
    const float *pos2_w, *pos2_b;   /* mlp2.2 (6,256), (6)    */
} t2s_mlp_layer_weights;
typedef struct t2s_mlp_weights {
    t2s_mlp_layer_weights layer[T2S_MLP_LAYERS];
} t2s_mlp_weights;
/* Transposes the weights into `packed` (T2S_MLP_PACKED_FLOATS floats, caller-owned device buffer).  Every pointer's
 * allocation extent is checked against the shape above.  Stateless: pack again after the weights change. */
int t2s_mlp_pack(const t2s_mlp_weights* w, float* packed, void* stream);
/* MLP.forward (mlp.py:90-94): x (B,64,6); t (B) fp32 (int64 timesteps converted by the host mirror exactly as
 * `t * 100.0` promotes them) and freqs (32) = 10000^linspace(0,1,32) as the caller evaluated it (mlp.py:12) for
 * TimeEmbedding(64), mlp.py:5-18; text (B,128) or NULL (mlp.py:75 skips the cross attention) -> out (B,64,6); out may
 * alias x.  One launch, one workgroup per series. */
int t2s_mlp_forward(const float* packed, const float* x, const float* t, const float* freqs, const float* text, float* out,
                    int B, void* stream);
/* Gradients of the same forward (train.py:123-125 with --denoiser MLP): every tensor of t2s_mlp_layer_weights, in its own layout. */
typedef struct t2s_mlp_layer_grads {
    float *value_w, *value_b, *proj_w, *proj_b, *norm2_w, *norm2_b, *mlp0_w, *mlp0_b, *mlp2_w, *mlp2_b, *pos0_w, *pos0_b, *pos2_w,
        *pos2_b;
} t2s_mlp_layer_grads;
typedef struct t2s_mlp_grads {
    t2s_mlp_layer_grads layer[T2S_MLP_LAYERS];
} t2s_mlp_grads;
#define T2S_MLP_GRAD_PART_FLOATS 391744 /* 8 x 48,968: one series' contributions, what `scratch` holds per series */
/* Backward of t2s_mlp_forward for dout (B,64,6): the forward is recomputed from x (nothing else is saved); every series writes
 * its contribution to `scratch` (B x T2S_MLP_GRAD_PART_FLOATS floats, caller-owned) and the contributions are added in series
 * order -- bit-reproducible.  grads receives d loss / d parameter (overwritten, not accumulated); dx (B,64,6) the gradient of
 * x, or NULL.  `w` are the tensors `packed` was made from.  cross_attn.query / key get no gradient here: exact zeros are theirs. */
int t2s_mlp_backward(const t2s_mlp_weights* w, const float* packed, const float* x, const float* t, const float* freqs,
                     const float* text, const float* dout, float* dx, const t2s_mlp_grads* grads, float* scratch,
                     uint64_t scratch_floats, int B, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* T2S_H */
