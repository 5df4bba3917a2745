// Internal definition of the t2s_dit handle, shared by t2s_dit.hip (inference) and t2s_train.hip.
#pragma once
#include <vector>

#include "t2s_common.h"

struct t2s_train_ws;   // training workspace (t2s_train.hip), created on first use

struct t2s_dit {
    int max_seqs = 0;
    // parameters (device)
    float* arena = nullptr;  // all parameters, offsets below
    float *conv_w, *conv_b, *patch_w, *patch_b, *pos, *ln_w, *ln_b, *out_w, *out_b, *freqs;
    float *qkv_b[t2s::NBLK], *proj_b[t2s::NBLK], *fc1_b[t2s::NBLK], *fc2_b[t2s::NBLK], *ada_b;
    t2s::f32x4 *qkv_p[t2s::NBLK], *proj_p[t2s::NBLK], *fc1_p[t2s::NBLK], *fc2_c[t2s::NBLK], *ada_p;
    // the same row-chain weights in the 16-token kernel's fragment order (t2s_rows16.h; small launches)
    t2s::f32x4 *qkv_p16[t2s::NBLK], *proj_p16[t2s::NBLK], *fc1_p16[t2s::NBLK], *fc2_c16[t2s::NBLK];
    // workspace (device), activations fragment-major
    float *h = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *ao = nullptr;
    float* h0 = nullptr;         // patchified tokens of the B distinct sequences of a CFG pass (both branches share them)
    float* mod = nullptr;        // (S, MODROW) adaLN modulation of all blocks (adaln_kernel)
    // T2S_MATH_BF16X3: k and V^T of the running block as split bf16 planes (t2s_x3.h), allocated on first use
    int math = 0;
    __bf16 *k3 = nullptr, *v3 = nullptr;
    __bf16* w3 = nullptr;        // split (3 x bf16) weights of the row chain in chunk order (t2s_rows_x3.h)
    __bf16 *qkv3[t2s::NBLK], *proj3[t2s::NBLK], *fc13[t2s::NBLK], *fc2c3[t2s::NBLK];
    // optional in-situ kernel timing (HIP events on the launching stream; never under capture)
    t2s_train_ws* train = nullptr;
    int train_dtype = 0;         // T2S_TRAIN_F32 / T2S_TRAIN_BF16 (t2s_dit_set_train_dtype)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_class;   // class of interval i = [ev_pool[2i], ev_pool[2i+1]]
};


namespace t2s {
// launch classes of the in-situ timing (t2s_dit_timing_begin / _end[_ex]): inference forward 0-2, training step 3-8
// 9 / 10: the first (<qkv only>, with patchify) and the last (<proj + MLP> + final layer) row-chain launch of a forward; they
// are ALSO counted in class 1, which stays "every row-chain launch"
enum { TC_ATTN = 0, TC_ROWS = 1, TC_OTHER = 2, TC_TR_GEMM = 3, TC_TR_ATTN_FWD = 4, TC_TR_ATTN_BWD = 5, TC_TR_WGRAD = 6,
       TC_TR_ELEM = 7, TC_TR_TAIL = 8, TC_ROWS_FIRST = 9, TC_ROWS_LAST = 10, TC_COUNT = 11 };
struct TimeScope {   // records an event pair around the launches issued in its scope when timing is on
    t2s_dit* h; hipStream_t st; bool on;
    TimeScope(t2s_dit* h_, int cls, hipStream_t st_) : h(h_), st(st_), on(h_->timing) {
        if (!on) return;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
        h->ev_pool.push_back(a); h->ev_pool.push_back(b); h->ev_class.push_back(cls);
        (void)hipEventRecord(a, st);
    }
    ~TimeScope() { if (on) (void)hipEventRecord(h->ev_pool.back(), st); }
};
}  // namespace t2s
