// Host-side sanitizer driver of the C ABI (tools/host_asan/build.sh): links the host-only ASan / UBSan objects of csrc/*.hip against
// hip_stub.cpp and walks the entry points of include/gsdd.h with
//   (1) valid descriptors at the workload's sizes (fake device pointers: the wrappers must never dereference them on the host) -- every
//       grid / LDS / workspace computation runs under the sanitizers and the stub validates each launch configuration;
//   (2) every workspace contract violated by one byte: the call must return GSDD_E_ARG and launch NOTHING;
//   (3) null pointers, bad sizes, unknown variant / mode values: GSDD_E_ARG, nothing launched;
//   (4) a failing hipFuncSetAttribute: reported as GSDD_E_HIP by that call and retried (successfully) by the next one.
// Exit code 0 = all expectations met and no sanitizer report.  Test infrastructure only.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/gsdd.h"

extern "C" long gsdd_stub_launches();
extern "C" void gsdd_stub_fail_next_set_attribute(long n);

static int g_failed = 0, g_checked = 0;
#define EXPECT(expr, want_rc, want_launch)                                                                                     \
    do {                                                                                                                       \
        const long before_ = gsdd_stub_launches();                                                                             \
        const int rc_ = (expr);                                                                                                \
        const long n_ = gsdd_stub_launches() - before_;                                                                        \
        ++g_checked;                                                                                                           \
        if (rc_ != (want_rc) || ((want_launch) ? n_ <= 0 : n_ != 0)) {                                                         \
            std::fprintf(stderr, "FAIL %s:%d  %s -> rc %d (want %d), %ld launches (want %s) [%s]\n", __FILE__, __LINE__, #expr, rc_, \
                         (int)(want_rc), n_, (want_launch) ? ">0" : "0", gsdd_last_error());                                   \
            ++g_failed;                                                                                                        \
        }                                                                                                                      \
    } while (0)

// distinct fake "device" addresses, 256-byte aligned, in a range no host allocation lives in
static uintptr_t g_next = 0x7000000000ull;
template <class T = float>
static T* devp(size_t bytes = 1 << 20) {
    T* p = reinterpret_cast<T*>(g_next);
    g_next += (bytes + 255) / 256 * 256 + 4096;
    return p;
}

int main() {
    void* st = nullptr;
    const int B = 16, L = 4096, H = 16, K = 4096, T = 100;
    const int64_t M = (int64_t)B * L;

    // ---------------------------------------------------------------- attention (sampler)
    {
        float *q = devp(), *k = devp(), *v = devp(), *out = devp();
        const int64_t need = gsdd_d3pm_attention_workspace_bytes(B, L, H);
        void* ws = devp<void>(need);
        uint64_t* redo = devp<uint64_t>();
        for (int mode = GSDD_ATTN_AUTO; mode <= GSDD_ATTN_KC256; ++mode)
            EXPECT(gsdd_d3pm_attention(q, k, v, B, L, H, out, ws, need, redo, mode, st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_attention(q, nullptr, nullptr, B, L, H, out, ws, need, redo, GSDD_ATTN_A8, st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_attention(q, k, v, B, L, H, out, ws, need - 1, redo, GSDD_ATTN_AUTO, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention(q, nullptr, nullptr, B, L, H, out, ws, need - 1, redo, GSDD_ATTN_AUTO, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention(q, nullptr, nullptr, B, L, H, out, nullptr, 0, redo, GSDD_ATTN_AUTO, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention(q, nullptr, nullptr, B, L, H, out, ws, need, redo, GSDD_ATTN_F32PV, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention(q, k, v, B, L, H, out, ws, need, redo, 99, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention(q, k, v, B, L, H, out, ws, need, redo, -1, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention(q, k, nullptr, B, L, H, out, ws, need, redo, 0, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention(q, k, v, 0, L, H, out, ws, need, redo, 0, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention(q, k, v, 1 << 20, L, H, out, ws, need, redo, 0, st), GSDD_E_ARG, false);      // grid beyond 2^31
        EXPECT(gsdd_d3pm_attention(q, k, v, 2, 48, H, out, nullptr, 0, nullptr, 0, st), GSDD_OK, true);          // ragged: workspace-free kernel
        EXPECT(gsdd_d3pm_attention(q, k, v, 2, 50, H, out, nullptr, 0, nullptr, 0, st), GSDD_OK, true);          // L % 16 != 0: vector kernel
        // training forward + backward
        float *lse = devp(), *o = devp(), *dO = devp(), *dqkv = devp(), *scratch = devp();
        EXPECT(gsdd_d3pm_attention_train(q, k, v, B, L, H, out, lse, ws, need, GSDD_ATTN_AUTO, st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_attention_train(q, k, v, B, L, H, out, lse, ws, need, GSDD_ATTN_P22, st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_attention_train(q, k, v, B, L, H, out, lse, ws, need, GSDD_ATTN_A8, st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_attention_train(q, k, v, B, L, H, out, lse, nullptr, 0, GSDD_ATTN_AUTO, st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_attention_train(q, k, v, B, L, H, out, lse, ws, need - 1, GSDD_ATTN_AUTO, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention_train(q, k, v, B, L, H, out, lse, ws, need, GSDD_ATTN_P11, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention_train(q, k, v, B, L, H, out, nullptr, ws, need, GSDD_ATTN_AUTO, st), GSDD_E_ARG, false);
        const int64_t bneed = gsdd_d3pm_attention_bwd_workspace_bytes(B, L, H);
        void* bws = devp<void>(bneed);
        for (int variant = GSDD_ATTN_BWD_AUTO; variant <= GSDD_ATTN_BWD_DEV_LAST; ++variant)
            EXPECT(gsdd_d3pm_attention_bwd(q, k, v, o, dO, lse, B, L, H, dqkv, scratch, bws, bneed, variant, st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_attention_bwd(q, k, v, o, dO, lse, B, L, H, dqkv, scratch, bws, bneed - 1, 0, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention_bwd(q, k, v, o, dO, lse, B, L, H, dqkv, nullptr, nullptr, 0, 0, st), GSDD_E_ARG, false);   // vector kernels need scratch
        EXPECT(gsdd_d3pm_attention_bwd(q, k, v, o, dO, lse, B, L, H, dqkv, scratch, nullptr, 0, 0, st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_attention_bwd(q, k, v, o, dO, lse, B, L, H, dqkv, scratch, bws, bneed, 99, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_attention_bwd(q, k, v, o, nullptr, lse, B, L, H, dqkv, scratch, bws, bneed, 0, st), GSDD_E_ARG, false);
    }

    // ---------------------------------------------------------------- fused layer
    {
        gsdd_layer_desc d;
        std::memset(&d, 0, sizeof d);
        d.y = devp(); d.x = devp(); d.M = 2 * M; d.L = L; d.n_embd = 64; d.hidden = 256; d.cvec = devp();
        d.wproj = devp(); d.bproj = devp(); d.ln2_g = devp(); d.ln2_b = devp(); d.w1 = devp(); d.b1 = devp(); d.w2 = devp(); d.b2 = devp();
        d.ada = devp(); d.t2 = devp<int64_t>(); d.wqkv = devp(); d.bqkv = devp(); d.qkv = devp();
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_E_ARG, false);                          // no fragment images at all
        d.layer_h2 = devp<void>(); d.wqkv_h2 = devp<void>();
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_OK, true);
        d.variant = GSDD_LAYER_X3P;
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_E_ARG, false);                          // asks for the bf16x3 kernel, has only the f16 images
        d.w2_x3 = devp<void>(); d.wqkv_x3 = devp<void>();
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_OK, true);
        d.variant = 1;
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_E_ARG, false);                          // the removed split-on-the-fly kernel's old number
        d.variant = GSDD_LAYER_AUTO;
        const int64_t need = gsdd_d3pm_attention_workspace_bytes(2 * B, L, 16);
        d.kv_img = devp<void>(need); d.kv_img_bytes = need;
        d.range_flag = devp<int>();
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_OK, true);
        d.kv_img_bytes = need - 1;
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_E_ARG, false);
        d.kv_img_bytes = need;
        d.L = 4090;
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_E_ARG, false);                          // images need L % 32 == 0
        d.L = L; d.n_embd = 128;
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_E_ARG, false);
        d.n_embd = 64;
        const float* keep = d.y; d.y = nullptr;                                       // q|k|v stage only (block 0)
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_OK, true);
        d.qkv = nullptr;
        EXPECT(gsdd_d3pm_layer(&d, st), GSDD_E_ARG, false);
        d.y = keep;
        EXPECT(gsdd_d3pm_layer(nullptr, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_logits(devp(), 2 * M, 64, devp(), devp(), devp(), devp(), K, devp(), st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_logits(devp(), 2 * M, 64, devp(), devp(), devp(), devp(), K + 2, devp(), st), GSDD_E_ARG, false);
        // a failing per-device attribute call is reported and retried (first use of the bf16x3 q|k|v-only instantiation set)
        gsdd_stub_fail_next_set_attribute(1);
        EXPECT(gsdd_rows_linear(devp(), M, 64, devp<void>(), 256, devp(), nullptr, 0, nullptr, devp(), 0, st), GSDD_E_HIP, false);
        EXPECT(gsdd_rows_linear(devp(), M, 64, devp<void>(), 256, devp(), nullptr, 0, nullptr, devp(), 0, st), GSDD_OK, true);
        EXPECT(gsdd_rows_linear(devp(), M, 64, devp<void>(), 320, devp(), nullptr, 0, nullptr, devp(), 0, st), GSDD_E_ARG, false);
    }

    // ---------------------------------------------------------------- posterior step / q_sample / training objective
    {
        gsdd_step_desc d;
        std::memset(&d, 0, sizeof d);
        d.logits_c = devp(); d.logits_u = devp(); d.tok_in = devp<int64_t>(); d.tok_out = devp<int64_t>();
        d.B = B; d.L = L; d.K = K; d.T = T; d.guidance = 2.f;
        const float* sched[8];
        for (int i = 0; i < 8; ++i) d.sched[i] = sched[i] = devp();
        d.t_dev = devp<int64_t>(); d.stream_dev = devp<int64_t>(); d.seed = 1;
        EXPECT(gsdd_d3pm_step(&d, st), GSDD_OK, true);
        d.occupancy = 3;
        EXPECT(gsdd_d3pm_step(&d, st), GSDD_OK, true);
        d.occupancy = 7;
        EXPECT(gsdd_d3pm_step(&d, st), GSDD_E_ARG, false);
        d.occupancy = 0;
        for (int k : {4, 32, 768, 1024, 2048, 4092, 8192}) { d.K = k; EXPECT(gsdd_d3pm_step(&d, st), GSDD_OK, true); }
        d.K = 8196;
        EXPECT(gsdd_d3pm_step(&d, st), GSDD_E_ARG, false);
        d.K = 30;
        EXPECT(gsdd_d3pm_step(&d, st), GSDD_E_ARG, false);
        d.K = K; d.sched[3] = nullptr;
        EXPECT(gsdd_d3pm_step(&d, st), GSDD_E_ARG, false);
        EXPECT(gsdd_d3pm_q_sample(devp<int64_t>(), devp<int64_t>(), B, L, K, T, sched, devp<int64_t>(), 1, devp<int64_t>(), 0, st), GSDD_OK, true);
        gsdd_train_desc t;
        std::memset(&t, 0, sizeof t);
        t.logits = devp(); t.x0 = devp<int64_t>(); t.xt = devp<int64_t>(); t.t_dev = devp<int64_t>(); t.pt = devp();
        t.B = B; t.L = L; t.K = K; t.T = T;
        for (int i = 0; i < 8; ++i) t.sched[i] = sched[i];
        t.mask_weight[0] = t.mask_weight[1] = 1.f; t.aux_weight = 5e-4f; t.adaptive_aux = 1;
        t.kl = devp(); t.nll = devp(); t.aux = devp(); t.x0_recon = devp<int64_t>(); t.xt1_recon = devp<int64_t>();
        t.Lt_history = devp(); t.Lt_count = devp(); t.loss = devp(); t.per_sample = devp();
        EXPECT(gsdd_d3pm_train_loss(&t, st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_train_loss_grad(&t, devp(), st), GSDD_OK, true);
        EXPECT(gsdd_d3pm_train_loss_bwd(&t, devp(), st), GSDD_OK, true);
        t.K = 8192;                                                          // J = 32: the 128 KB LDS attribute is requested here only
        EXPECT(gsdd_d3pm_train_loss_grad(&t, devp(), st), GSDD_OK, true);
        t.K = K;
        EXPECT(gsdd_d3pm_train_loss_grad(&t, nullptr, st), GSDD_E_ARG, false);
    }

    // ---------------------------------------------------------------- VQ-VAE pieces with workspaces
    {
        const int64_t Mz = 64 * 4096;
        const int64_t need = gsdd_nearest_code_workspace_bytes(4096);
        EXPECT(gsdd_nearest_code(devp(), Mz, 128, devp(), 4096, devp<int64_t>(), devp(), devp<void>(), need, st), GSDD_OK, true);
        EXPECT(gsdd_nearest_code(devp(), Mz, 128, devp(), 4096, devp<int64_t>(), devp(), devp<void>(), need - 1, st), GSDD_E_ARG, false);
        EXPECT(gsdd_nearest_code(devp(), Mz, 128, devp(), 4096, devp<int64_t>(), nullptr, nullptr, 0, st), GSDD_OK, true);     // vector kernel
        EXPECT(gsdd_nearest_code(devp(), Mz, 130, devp(), 4096, devp<int64_t>(), nullptr, nullptr, 0, st), GSDD_E_ARG, false);
        const int64_t bn = gsdd_bn_train_workspace_bytes(Mz, 256);
        EXPECT(gsdd_bn_train(devp(), Mz, 256, devp(), devp(), 1e-5f, 0.1f, devp(), devp(), devp(), devp(), devp(), devp<void>(), bn, st), GSDD_OK, true);
        EXPECT(gsdd_bn_train(devp(), Mz, 256, devp(), devp(), 1e-5f, 0.1f, devp(), devp(), devp(), devp(), devp(), devp<void>(), bn - 1, st), GSDD_E_ARG, false);
        const int64_t bb = gsdd_bn_relu_bwd_workspace_bytes(Mz, 256);
        EXPECT(gsdd_bn_relu_bwd(devp(), devp(), Mz, 256, devp(), devp(), devp(), nullptr, devp(), devp(), devp(), devp<void>(), bb, st), GSDD_OK, true);
        EXPECT(gsdd_bn_relu_bwd(devp(), devp(), Mz, 256, devp(), devp(), devp(), nullptr, devp(), devp(), devp(), devp<void>(), bb - 1, st), GSDD_E_ARG, false);
        EXPECT(gsdd_mse(devp(), devp(), 1 << 24, 1.f, devp(), devp<void>(), 8192, st), GSDD_OK, true);
        EXPECT(gsdd_mse(devp(), devp(), 1 << 24, 1.f, devp(), devp<void>(), 8191, st), GSDD_E_ARG, false);
        EXPECT(gsdd_axial_attention(devp(), 2, 16, 16, 16, 256, 2, devp(), GSDD_AXIAL_AUTO, st), GSDD_OK, true);
        EXPECT(gsdd_axial_attention(devp(), 2, 16, 16, 16, 256, 2, devp(), GSDD_AXIAL_VALU, st), GSDD_OK, true);
        EXPECT(gsdd_axial_attention(devp(), 2, 16, 16, 16, 256, 2, devp(), 5, st), GSDD_E_ARG, false);
        EXPECT(gsdd_axial_attention_bwd(devp(), devp(), 2, 16, 16, 16, 256, 2, devp(), GSDD_AXIAL_AUTO, st), GSDD_OK, true);
        EXPECT(gsdd_axial_attention_bwd(devp(), devp(), 2, 16, 16, 16, 256, 2, devp(), -2, st), GSDD_E_ARG, false);
        // the generic GEMM at the decoder's largest shape, both back ends, and the weight gradient
        gsdd_gemm_desc g;
        std::memset(&g, 0, sizeof g);
        g.in = devp(); g.N = 16; g.Di = 19; g.Hi = 66; g.Wi = 66; g.Cin = 256; g.in_pitch = 256;
        g.Do = 16; g.Ho = 64; g.Wo = 64; g.sd = g.sh = g.sw = 1; g.ntaps = 8; g.taps = devp<int>();
        g.w = devp(); g.Cout = 256; g.epi_shift = devp(); g.act = 1;
        g.out = devp(); g.oD = 16; g.oH = 128; g.oW = 128; g.osd = 1; g.osh = 2; g.osw = 2; g.out_pitch = 256;
        EXPECT(gsdd_gemm(&g, st), GSDD_OK, true);
        g.flags = GSDD_GEMM_EXACT_F32;
        EXPECT(gsdd_gemm(&g, st), GSDD_OK, true);
        EXPECT(gsdd_conv_wgrad(&g, devp(), 256, devp(), st), GSDD_OK, true);
        g.flags = 0;
        EXPECT(gsdd_conv_wgrad(&g, devp(), 256, devp(), st), GSDD_OK, true);
        g.flags = 6;
        EXPECT(gsdd_gemm(&g, st), GSDD_E_ARG, false);
        EXPECT(gsdd_conv_wgrad(&g, devp(), 256, devp(), st), GSDD_E_ARG, false);
        g.flags = 0; g.out_pitch = 100;
        EXPECT(gsdd_gemm(&g, st), GSDD_E_ARG, false);
    }

    // ---------------------------------------------------------------- graph capture misuse, events
    {
        void* exec = nullptr;
        EXPECT(gsdd_graph_end(st, &exec), GSDD_E_HIP, false);                         // end without begin
        EXPECT(gsdd_graph_begin(st), GSDD_OK, false);
        EXPECT(gsdd_graph_begin(st), GSDD_E_HIP, false);                              // nested begin
        EXPECT(gsdd_graph_end(st, &exec), GSDD_OK, false);
        EXPECT(gsdd_graph_launch(exec, st), GSDD_OK, false);
        EXPECT(gsdd_graph_launch(nullptr, st), GSDD_E_ARG, false);
        EXPECT(gsdd_graph_destroy(exec), GSDD_OK, false);
        EXPECT(gsdd_graph_end(st, nullptr), GSDD_E_ARG, false);
        void *e0 = nullptr, *e1 = nullptr;
        float ms = -1.f;
        EXPECT(gsdd_event_create(&e0), GSDD_OK, false);
        EXPECT(gsdd_event_create(&e1), GSDD_OK, false);
        EXPECT(gsdd_event_record(e0, st), GSDD_OK, false);
        EXPECT(gsdd_event_elapsed_ms(e0, e1, &ms), GSDD_OK, false);
        EXPECT(gsdd_event_elapsed_ms(e0, nullptr, &ms), GSDD_E_ARG, false);
        EXPECT(gsdd_event_destroy(e0), GSDD_OK, false);
        EXPECT(gsdd_event_destroy(e1), GSDD_OK, false);
        EXPECT(gsdd_event_create(nullptr), GSDD_E_ARG, false);
    }

    std::printf("host_asan driver: %d expectations, %d failed, %ld launches validated\n", g_checked, g_failed, gsdd_stub_launches());
    return g_failed == 0 ? 0 : 1;
}
