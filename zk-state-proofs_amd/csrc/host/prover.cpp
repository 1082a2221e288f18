// Replaces the body of the reference's `client.prove(&pk, stdin).run()`
// (prover/src/bin/main.rs:71-74) below the executor: sp1-prover / sp1-stark's
// commit -> quotient -> open -> FRI flow, here as one fixed sequence of HIP kernel
// launches per batch with the Fiat-Shamir transcript kept on the device.
#include "prover.hpp"

#include <algorithm>

namespace zksp {

Workspace::~Workspace() {
  for (void* p : allocs)
    if (p) (void)hipFree(p);
}

template <class T>
static bool dalloc(Workspace* ws, T** p, size_t count) {
  void* q = nullptr;
  if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) return false;
  ws->allocs.push_back(q);
  *p = static_cast<T*>(q);
  return true;
}

static int ceil_log2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}

int workspace_ensure(Context* ctx, int logh, int batch, int max_perms) {
  batch = std::max(batch, ctx->batch_hint);
  if (ctx->ws && ctx->ws->logh == logh && ctx->ws->batch >= batch && ctx->ws->max_perms >= max_perms) return 0;
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->ws.reset(new Workspace());
  Workspace* ws = ctx->ws.get();
  ws->logh = logh;
  ws->batch = batch;
  ws->max_perms = max_perms;
  const size_t B = (size_t)batch, h = (size_t)1 << logh, n = 2 * h, W = kTraceWidth;
  const uint32_t Q = ctx->params.num_queries;
  ws->body_words = proof_body_words(logh, Q);
  const size_t n_open_words = (2 * W + 8 + 2 * kPermWidth) * 4;
  ws->io_rows_log = (size_t)bus_io_log_rows(logh);
  const size_t RIO = (size_t)1 << ws->io_rows_log;
  ws->open_rows_log = (size_t)ceil_log2((n_open_words + 7) / 8);
  const size_t R = (size_t)1 << ws->open_rows_log;
  ws->fri_layer_stride = 0;
  ws->fri_tree_stride = 0;
  for (int k = 0; k < logh; ++k) {
    size_t hk = h >> k;
    ws->fri_layer_stride += 2 * hk * 4;
    ws->fri_tree_stride += (2 * hk - 1) * 8;
  }
  ws->fri_layer_stride += 2 * 4;  // the final 2-point layer
  const size_t scratch_words = std::max<size_t>(W * h, std::max<size_t>((size_t)13 * n * 4, (size_t)reduce_nchunks((int)W) * n * 4));
  bool ok = true;
  ok &= dalloc(ws, &ws->states, B * (size_t)max_perms * 25);
  ok &= dalloc(ws, &ws->n_perms, B);
  ok &= dalloc(ws, &ws->init_obs, B * kInitObs);
  ok &= dalloc(ws, &ws->trace, B * scratch_words);  // also quotient / reduce partial sums once the trace is dead
  ok &= dalloc(ws, &ws->coef_t, B * W * h);
  ok &= dalloc(ws, &ws->lde_t, B * W * n);
  ok &= dalloc(ws, &ws->tree_t, B * (2 * n - 1) * 8);
  ok &= dalloc(ws, &ws->ch, B);
  ok &= dalloc(ws, &ws->alpha, B * 4);
  ok &= dalloc(ws, &ws->alpha_pows, B * (size_t)kNumAllConstraints * 4);
  ok &= dalloc(ws, &ws->io, B * 8 * RIO);
  ok &= dalloc(ws, &ws->tree_io, B * (2 * RIO - 1) * 8);
  ok &= dalloc(ws, &ws->bus_ch, B * 8);
  ok &= dalloc(ws, &ws->beta_pows, B * (size_t)kBusTuple * 4);
  ok &= dalloc(ws, &ws->bus_terms, B * h * 4);
  ok &= dalloc(ws, &ws->phi, B * kPermWidth * h);
  ok &= dalloc(ws, &ws->coef_p, B * kPermWidth * h);
  ok &= dalloc(ws, &ws->lde_p, B * kPermWidth * n);
  ok &= dalloc(ws, &ws->tree_p, B * (2 * n - 1) * 8);
  ok &= dalloc(ws, &ws->cum_sum, B * 4);
  ok &= dalloc(ws, &ws->quot, B * 8 * h);
  ok &= dalloc(ws, &ws->coef_q, B * 8 * h);
  ok &= dalloc(ws, &ws->lde_q, B * 8 * n);
  ok &= dalloc(ws, &ws->tree_q, B * (2 * n - 1) * 8);
  ok &= dalloc(ws, &ws->zeta, B * 4);
  ok &= dalloc(ws, &ws->zpow, B * 2 * h * 4);
  ok &= dalloc(ws, &ws->opened, B * 8 * R);
  ok &= dalloc(ws, &ws->tree_o, B * (2 * R - 1) * 8);
  ok &= dalloc(ws, &ws->af, B * 4);
  ok &= dalloc(ws, &ws->af_pows, B * (2 * W + 16) * 4);
  ok &= dalloc(ws, &ws->bsum, B * 5 * 4);
  ok &= dalloc(ws, &ws->fri_layers, B * ws->fri_layer_stride);
  ok &= dalloc(ws, &ws->fri_trees, B * ws->fri_tree_stride);
  ok &= dalloc(ws, &ws->betas, B * (size_t)logh * 4);
  ok &= dalloc(ws, &ws->witness, B);
  ok &= dalloc(ws, &ws->indices, B * Q);
  ok &= dalloc(ws, &ws->body, B * ws->body_words);
  if (!ok) {
    ctx->ws.reset();
    return ctx->fail(3, "workspace: hipMalloc failed");
  }
  // the zero padding of the opened-value matrix is written once
  ZKSP_HIP_CHECK(ctx, hipMemsetAsync(ws->opened, 0, B * 8 * R * 4, ctx->stream));
  return 0;
}

int prove_resident(Context* ctx) {
  Workspace* ws = ctx->ws.get();
  if (!ws || ws->n == 0) return ctx->fail(1, "prove_resident: no batch loaded");
  const DeviceDomain* dom = ctx->domain(ws->logh);
  if (!dom) return 3;
  hipStream_t s = ctx->stream;
  const P2Consts* kc = ctx->d_consts;
  const int logh = ws->logh, logn = logh + 1, B = ws->n, W = kTraceWidth;
  const size_t h = (size_t)1 << logh, n = 2 * h;
  const int Q = (int)ctx->params.num_queries, pow_bits = (int)ctx->params.pow_bits;
  const size_t tree_stride = (2 * n - 1) * 8, root_off = (2 * n - 2) * 8;
  const size_t R = (size_t)1 << ws->open_rows_log;

  {
    ProfileSpan sp(ctx, "keccak_trace");
    launch_keccak_trace(s, ws->states, ws->max_perms, ws->n_perms, ws->trace, logh, B);
  }
  {
    ProfileSpan sp(ctx, "lde_trace");
    launch_lde(s, ws->trace, ws->coef_t, ws->lde_t, dom->twc_fwd, dom->twc_inv, dom->in_scale_br, 0, 0, dom->out_scale_br,
               logh, (size_t)B * W);
  }
  {
    ProfileSpan sp(ctx, "leaf_hash_trace");  // exactly one launch of leaf_hash_trace_kernel
    launch_merkle_commit(s, ws->lde_t, (size_t)W * n, W, logn, ws->tree_t, tree_stride, B, kc, false);
  }
  {
    ProfileSpan sp(ctx, "merkle_upper");
    launch_merkle_upper(s, logn, ws->tree_t, tree_stride, B, kc);
  }
  const size_t RIO = (size_t)1 << ws->io_rows_log;
  {
    ProfileSpan sp(ctx, "bus_io");
    // public I/O limbs (input || keccak-f(input)) and their Merkle root, absorbed before the trace root
    launch_keccak_io(s, ws->states, ws->max_perms, ws->n_perms, ws->io, 8 * RIO, B);
    launch_merkle_commit(s, ws->io, 8 * RIO, 8, (int)ws->io_rows_log, ws->tree_io, (2 * RIO - 1) * 8, B, kc);
  }
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_init(s, ws->ch, ws->init_obs, kInitObs, B, kc);
    launch_ch_observe_sample(s, ws->ch, ws->tree_io + (2 * RIO - 2) * 8, (2 * RIO - 1) * 8, 8, ws->bus_ch, 8, 0, B, kc);
    // trace root -> gamma, beta
    launch_ch_observe_sample(s, ws->ch, ws->tree_t + root_off, tree_stride, 8, ws->bus_ch, 8, 2, B, kc);
    launch_ext_powers(s, ws->bus_ch + 4, 8, kR1, ws->beta_pows, (size_t)kBusTuple * 4, kBusTuple, 0, B);
  }
  {
    ProfileSpan sp(ctx, "bus_trace");
    launch_bus_perm_trace(s, ws->trace, ws->bus_ch, ws->beta_pows, ws->bus_terms, ws->phi, ws->cum_sum, logh, B);
    launch_lde(s, ws->phi, ws->coef_p, ws->lde_p, dom->twc_fwd, dom->twc_inv, dom->in_scale_br, 0, 0, dom->out_scale_br,
               logh, (size_t)B * kPermWidth);
    launch_merkle_commit(s, ws->lde_p, (size_t)kPermWidth * n, kPermWidth, logn, ws->tree_p, tree_stride, B, kc);
  }
  {
    ProfileSpan sp(ctx, "transcript");
    // running-sum root, cumulative sum -> alpha
    launch_ch_observe_sample(s, ws->ch, ws->tree_p + root_off, tree_stride, 8, ws->alpha, 4, 0, B, kc);
    launch_ch_observe_sample(s, ws->ch, ws->cum_sum, 4, 4, ws->alpha, 4, 1, B, kc);
    launch_ext_powers(s, ws->alpha, 4, kR1, ws->alpha_pows, (size_t)kNumAllConstraints * 4, kNumAllConstraints, 0, B);
  }
  {
    ProfileSpan sp(ctx, "quotient");
    QuotientArgs qa;
    qa.lde = ws->lde_t;
    qa.lde_p = ws->lde_p;
    qa.alpha_pows = ws->alpha_pows;
    qa.bus_ch = ws->bus_ch;
    qa.beta_pows = ws->beta_pows;
    qa.cum_sum = ws->cum_sum;
    qa.sel_first = dom->sel_first;
    qa.sel_trans = dom->sel_trans;
    qa.sel_last = dom->sel_last;
    qa.zh_inv = dom->zh_inv;
    qa.partial = ws->trace;
    qa.quot = ws->quot;
    qa.logh = logh;
    qa.batch = B;
    launch_keccak_quotient(s, qa);
  }
  {
    ProfileSpan sp(ctx, "lde_quot");
    // columns 4c..4c+3 of every proof were evaluated over coset c: scale tables 1 and 2
    launch_lde(s, ws->quot, ws->coef_q, ws->lde_q, dom->twc_fwd, dom->twc_inv, dom->in_scale_br + h, 2, 1,
               dom->out_scale_br, logh, (size_t)B * 8);
  }
  {
    ProfileSpan sp(ctx, "merkle_quot");
    launch_merkle_commit(s, ws->lde_q, 8 * n, 8, logn, ws->tree_q, tree_stride, B, kc);
  }
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_observe_sample(s, ws->ch, ws->tree_q + root_off, tree_stride, 8, ws->zeta, 4, 1, B, kc);
    launch_ext_powers(s, ws->zeta, 4, kR1, ws->zpow, 2 * h * 4, (int)h, logh, B, /*centred=*/1);
    launch_ext_powers(s, ws->zeta, 4, dom->w_h, ws->zpow + h * 4, 2 * h * 4, (int)h, logh, B, /*centred=*/1);
  }
  {
    ProfileSpan sp(ctx, "open");
    launch_open(s, ws->coef_t, (size_t)W * h, W, logh, ws->zpow, 2 * h * 4, 2, ws->opened, 8 * R, (size_t)W, B);
    launch_open(s, ws->coef_q, 8 * h, 8, logh, ws->zpow, 2 * h * 4, 1, ws->opened + (size_t)2 * W * 4, 8 * R, 0, B);
    launch_open(s, ws->coef_p, (size_t)kPermWidth * h, kPermWidth, logh, ws->zpow, 2 * h * 4, 2,
                ws->opened + (size_t)(2 * W + 8) * 4, 8 * R, (size_t)kPermWidth, B);
  }
  {
    ProfileSpan sp(ctx, "merkle_open");
    launch_merkle_commit(s, ws->opened, 8 * R, 8, (int)ws->open_rows_log, ws->tree_o, (2 * R - 1) * 8, B, kc);
  }
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_observe_sample(s, ws->ch, ws->tree_o + (2 * R - 2) * 8, (2 * R - 1) * 8, 8, ws->af, 4, 1, B, kc);
    launch_ext_powers(s, ws->af, 4, kR1, ws->af_pows, (size_t)(2 * W + 16) * 4, 2 * W + 16, 0, B);
  }
  {
    ProfileSpan sp(ctx, "reduce_openings");
    ReduceArgs ra;
    ra.lde_t = ws->lde_t;
    ra.lde_q = ws->lde_q;
    ra.lde_p = ws->lde_p;
    ra.af_pows = ws->af_pows;
    ra.opened = ws->opened;
    ra.opened_stride = 8 * R;
    ra.zeta = ws->zeta;
    ra.xs = dom->xs;
    ra.partial = ws->trace;
    ra.bsum = ws->bsum;
    ra.out = ws->fri_layers;
    ra.out_stride = ws->fri_layer_stride;
    ra.w_h = dom->w_h;
    ra.width = W;
    ra.logh = logh;
    ra.batch = B;
    launch_reduce_openings(s, ra);
  }
  size_t loff = 0, toff = 0;
  // layers of more than 512 leaves: commit, transcript and fold as separate launches
  // ... and in a large batch every layer: the fused tail is one workgroup per proof, a latency form
  const int k_tail = B >= 64 ? logh : std::max(0, logh - kFriTailMaxLogLeaves);
  for (int k = 0; k < k_tail; ++k) {
    const int loghk = logh - k;
    const size_t hk = h >> k;
    {
      ProfileSpan sp(ctx, "fri_commit");
      launch_fri_commit(s, ws->fri_layers + loff, ws->fri_layer_stride, loghk, ws->fri_trees + toff * 8,
                        ws->fri_tree_stride, B, kc);
    }
    {
      ProfileSpan sp(ctx, "transcript");
      launch_ch_observe_sample(s, ws->ch, ws->fri_trees + (toff + 2 * hk - 2) * 8, ws->fri_tree_stride, 8,
                               ws->betas + (size_t)k * 4, (size_t)logh * 4, 1, B, kc);
    }
    {
      ProfileSpan sp(ctx, "fri_fold");
      launch_fri_fold(s, ws->fri_layers + loff, ws->fri_layer_stride, ws->fri_layers + loff + 2 * hk * 4,
                      ws->fri_layer_stride, ws->betas + (size_t)k * 4, (size_t)logh * 4, dom->tw_inv, k,
                      dom->fold_xinv[2 * k], dom->fold_xinv[2 * k + 1], loghk, B);
    }
    loff += 2 * hk * 4;
    toff += 2 * hk - 1;
  }
  // the remaining layers in one launch
  if (k_tail < logh) {
    ProfileSpan sp(ctx, "fri_commit");
    FriTailArgs ta;
    ta.layers = ws->fri_layers;
    ta.layer_stride = ws->fri_layer_stride;
    ta.trees = ws->fri_trees;
    ta.tree_stride = ws->fri_tree_stride;
    ta.ch = ws->ch;
    ta.betas = ws->betas;
    ta.beta_stride = (size_t)logh * 4;
    ta.tw_inv = dom->tw_inv;
    for (int i = 0; i < 48; ++i) ta.xinv[i] = i < 2 * logh ? dom->fold_xinv[i] : 0;
    for (auto& j : ta.join) j = nullptr;
    ta.logh = logh;
    ta.k_start = k_tail;
    ta.loff_start = loff;
    ta.toff_start = toff;
    launch_fri_tail(s, ta, B, kc);
    for (int k = k_tail; k < logh; ++k) {
      const size_t hk = h >> k;
      loff += 2 * hk * 4;
      toff += 2 * hk - 1;
    }
  }
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_observe_sample(s, ws->ch, ws->fri_layers + loff, ws->fri_layer_stride, 4, ws->alpha, 4, 0, B, kc);
  }
  {
    ProfileSpan sp(ctx, "grind");
    launch_ch_grind(s, ws->ch, ws->witness, pow_bits, B, kc);
  }
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_queries(s, ws->ch, ws->witness, ws->indices, Q, pow_bits, logn, B, kc);
  }
  {
    ProfileSpan sp(ctx, "assemble");
    AssembleArgs aa;
    aa.lde_t = ws->lde_t;
    aa.tree_t = ws->tree_t;
    aa.lde_q = ws->lde_q;
    aa.tree_q = ws->tree_q;
    aa.lde_p = ws->lde_p;
    aa.tree_p = ws->tree_p;
    aa.cum_sum = ws->cum_sum;
    aa.opened = ws->opened;
    aa.fri_layers = ws->fri_layers;
    aa.fri_trees = ws->fri_trees;
    aa.witness = ws->witness;
    aa.indices = ws->indices;
    aa.body = ws->body;
    aa.lde_t_stride = (size_t)W * n;
    aa.tree_t_stride = tree_stride;
    aa.lde_q_stride = 8 * n;
    aa.tree_q_stride = tree_stride;
    aa.lde_p_stride = (size_t)kPermWidth * n;
    aa.tree_p_stride = tree_stride;
    aa.opened_stride = 8 * R;
    aa.fri_layer_stride = ws->fri_layer_stride;
    aa.fri_tree_stride = ws->fri_tree_stride;
    aa.body_stride = ws->body_words;
    aa.width = W;
    aa.logh = logh;
    aa.n_queries = Q;
    aa.batch = B;
    // the body buffer may still be the source of the previous group's device-to-host copy
    if (ctx->body_free) (void)hipStreamWaitEvent(s, ctx->body_free, 0);
    launch_assemble(s, aa);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx->fail(3, std::string("prove_resident: ") + hipGetErrorString(e));
  return 0;
}

}  // namespace zksp
