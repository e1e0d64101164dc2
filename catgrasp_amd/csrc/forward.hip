// HOST code (no kernel of its own): PointNetCls.forward (pointnet2.py:289-299) in eval mode as ONE call -- the twelve launches of the
// exact-f32 path (three fused per-point passes, nine dense layers) issued back to back from C.
//
// The python engine (catgrasp_amd/engine.py: cls_forward) issues the same launches one ctypes call at a time, each with its output
// allocation and argument marshalling: ~10 us of interpreter per launch, ~0.15 ms per forward -- invisible behind a 16,384-candidate
// chunk (200 ms of matrix time), a third of a predict_batch call of a few poses, which is what the reference issues per object
// (predicter.py:67-94: hundreds of poses, not 50,000).  Same kernels, same arguments, same order: the results are those of the
// python path bit for bit.  All intermediates live in a caller-owned workspace (no allocation here either).
#include <stddef.h>
#include "../../include/catgrasp_amd.h"

namespace {
inline size_t up4(size_t n) { return (n + 3) & ~(size_t)3; }
}  // namespace

extern "C" size_t cg_pointnet_cls_workspace_floats(int B) {
  const size_t b = (size_t)(B > 0 ? B : 0);
  return 3 * (up4(b * 1024) + up4(b * 512) + up4(b * 256)) + up4(b * 9) + up4(b * 4096);
}

extern "C" int cg_pointnet_cls_forward(const float* x, int B, int N, const cg_cls_weights* w, int nsplit, float* ws, float* logits,
                                       float** trans_feat_t, void* stream) {
  if (B < 0 || N <= 0 || nsplit < 1) return CG_ERR_ARG;
  if (B == 0) return CG_OK;
  if (!x || !w || !ws || !logits || w->n_out <= 0) return CG_ERR_ARG;
  const size_t b = (size_t)B;
  float* g1 = ws;                 float* h1 = g1 + up4(b * 1024); float* h2 = h1 + up4(b * 512);
  float* t3 = h2 + up4(b * 256);  float* g2 = t3 + up4(b * 9);    float* h3 = g2 + up4(b * 1024);
  float* h4 = h3 + up4(b * 512);  float* t64 = h4 + up4(b * 256); float* g3 = t64 + up4(b * 4096);
  float* h5 = g3 + up4(b * 1024); float* h6 = h5 + up4(b * 512);
  int rc;
#define CG_TRY(call) do { rc = (call); if (rc != CG_OK) return rc; } while (0)
  // STN3d: conv1..conv3 + max, fc1, fc2, fc3 (+ I3)                                         pointnet2.py:170-185
  CG_TRY(cg_pointmlp_max(x, B, N, nullptr, w->stn_w1, w->stn_b1, 0, nullptr, nullptr, nullptr, w->stn_w2, w->stn_b2, w->stn_w3, w->stn_b3, 1, nsplit,
                         g1, nullptr, stream));
  CG_TRY(cg_gemm_bias_act(g1, B, 1024, 1024, w->stn_fc1, 512, w->stn_fc1b, nullptr, 1, 0, 1, 0, h1, 512, stream));
  CG_TRY(cg_gemm_bias_act(h1, B, 512, 512, w->stn_fc2, 256, w->stn_fc2b, nullptr, 1, 0, 1, 0, h2, 256, stream));
  CG_TRY(cg_gemm_bias_act(h2, B, 256, 256, w->stn_fc3, 9, w->stn_fc3b, nullptr, 1, 0, 0, 3, t3, 9, stream));
  // STNkd on top of the encoder's conv1: conv1..conv3 + max, fc1, fc2, fc3 (+ I64)           pointnet2.py:208-223, :243-252
  CG_TRY(cg_pointmlp_max(x, B, N, t3, w->enc_w1, w->enc_b1, 1, w->fstn_wm, w->fstn_bm, nullptr, w->fstn_w2, w->fstn_b2, w->fstn_w3, w->fstn_b3, 1,
                         nsplit, g2, nullptr, stream));
  CG_TRY(cg_gemm_bias_act(g2, B, 1024, 1024, w->fstn_fc1, 512, w->fstn_fc1b, nullptr, 1, 0, 1, 0, h3, 512, stream));
  CG_TRY(cg_gemm_bias_act(h3, B, 512, 512, w->fstn_fc2, 256, w->fstn_fc2b, nullptr, 1, 0, 1, 0, h4, 256, stream));
  CG_TRY(cg_gemm_bias_act(h4, B, 256, 256, w->fstn_fc3, 4096, w->fstn_fc3b, nullptr, 1, 0, 0, 64, t64, 4096, stream));
  // encoder: conv1, x.T64, conv2, conv3, max                                                 pointnet2.py:243-266
  CG_TRY(cg_pointmlp_max(x, B, N, t3, w->enc_w1, w->enc_b1, 2, nullptr, nullptr, t64, w->enc_w2, w->enc_b2, w->enc_w3, w->enc_b3, 0, nsplit, g3,
                         nullptr, stream));
  // head: fc1, fc2 (BN folded, ReLU; dropout is the identity in eval), fc3                   pointnet2.py:295-298
  CG_TRY(cg_gemm_bias_act(g3, B, 1024, 1024, w->head_fc1, 512, w->head_fc1b, nullptr, 1, 0, 1, 0, h5, 512, stream));
  CG_TRY(cg_gemm_bias_act(h5, B, 512, 512, w->head_fc2, 256, w->head_fc2b, nullptr, 1, 0, 1, 0, h6, 256, stream));
  CG_TRY(cg_gemm_bias_act(h6, B, 256, 256, w->head_fc3, w->n_out, w->head_fc3b, nullptr, 1, 0, 0, 0, logits, w->n_out, stream));
#undef CG_TRY
  if (trans_feat_t) *trans_feat_t = t64;
  return CG_OK;
}
