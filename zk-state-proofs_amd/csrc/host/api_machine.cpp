// C ABI of the machine-proof side (include/zksp.h, "machine proof" section): traced execution
// records for tests and the CPU oracle.  Nothing here throws across the boundary.
#include "../../../include/zksp.h"

#include <cstring>
#include <memory>
#include <new>

#include <thread>

#include "api_types.hpp"
#include "machine_defs.hpp"
#include "mprover.hpp"
#include "zeta_program.hpp"

using namespace zksp;

static_assert(mach::kHeaderWords == ZKSP_MACHINE_HEADER_WORDS && mach::kNumChips == ZKSP_MACHINE_CHIPS &&
                  mach::kPubTupleWords == ZKSP_PUB_TUPLE_WORDS,
              "include/zksp.h describes the proof layout: keep it in step with machine_defs.hpp");

// Header (version, chip heights, exit code, digests, key digest, payload words), public values, body: the proof object.
int machine_proof_from_parts(const zksp_pk* pk, const ExecutionRecord& r, const int* lh, const uint32_t* handover_pc,
                             const std::vector<uint32_t>& agg_leaves, const std::vector<uint32_t>& agg_keys,
                             const LeafCheckLog* leaf_check, const uint32_t* body, size_t body_words, zksp_proof** out) {
  zksp_proof* p = new (std::nothrow) zksp_proof();
  if (!p) return ZKSP_ERR_INVALID_ARG;
  try {
    const size_t pvw = (r.public_values.size() + 3) / 4;
    p->bytes.assign(((size_t)mach::kHeaderWords + pvw + body_words) * 4, 0);
    uint32_t* w = reinterpret_cast<uint32_t*>(p->bytes.data());
    w[0] = kProofMagic;
    w[1] = mach::kMachineVersion;
    for (int c = 0; c < mach::kNumChips; ++c) w[2 + c] = (uint32_t)lh[c];
    w[2 + mach::kNumChips] = r.exit_code;
    w[3 + mach::kNumChips] = (uint32_t)r.public_values.size();
    memcpy(w + 4 + mach::kNumChips, r.pv_digest.data(), 32);
    memcpy(w + 12 + mach::kNumChips, r.deferred_digest.data(), 32);
    memcpy(w + 20 + mach::kNumChips, pk->mvk.digest, 32);
    constexpr int kHo = mach::kNumCpuInst - 1;
    for (int i = 0; i < kHo; ++i) w[28 + mach::kNumChips + i] = handover_pc[i];
    w[28 + mach::kNumChips + kHo] = (uint32_t)(agg_leaves.size() / 8);
    if (!machine_nodes_public(agg_keys.empty() ? nullptr : agg_keys.data(), agg_leaves.data(), agg_leaves.size() / 8,
                              w + 29 + mach::kNumChips + kHo, w + 37 + mach::kNumChips + kHo, nullptr)) {
      delete p;
      return ZKSP_ERR_INVALID_ARG;
    }
    if (leaf_check) {  // the statement of the leaf-proof check: how many public bus tuples, and their digest
      const size_t n_pub = leaf_check->pub_tuples.size() / mach::kPubTupleWords;
      w[45 + mach::kNumChips + kHo] = (uint32_t)n_pub;
      machine_pub_digest(leaf_check->pub_tuples.data(), n_pub, w + 46 + mach::kNumChips + kHo);
    }
    if (!r.public_values.empty()) memcpy(w + mach::kHeaderWords, r.public_values.data(), r.public_values.size());
    memcpy(w + mach::kHeaderWords + pvw, body, body_words * 4);
  } catch (...) {
    delete p;
    return ZKSP_ERR_INVALID_ARG;
  }
  std::string err;
  p->version = mach::kMachineVersion;
  if (!parse_machine_header(p->bytes.data(), p->bytes.size(), &p->mhdr, &err)) {
    delete p;
    return ZKSP_ERR_PROOF_FORMAT;
  }
  *out = p;
  return ZKSP_OK;
}

extern "C" {

int zksp_machine_trace(zksp_client* c, const zksp_pk* pk, const zksp_stdin* stdin_, zksp_mtrace** out) {
  if (!c || !pk || !stdin_ || !out) return ZKSP_ERR_INVALID_ARG;
  *out = nullptr;
  zksp_mtrace* t = new (std::nothrow) zksp_mtrace();
  if (!t) return ZKSP_ERR_INVALID_ARG;
  t->prog = &pk->mprog;
  try {
    trace_execute(pk->elf, pk->mprog, stdin_->entries, (uint64_t)1 << 21, &t->t);
    t->t.agg_leaves = stdin_->agg_leaves;
    t->t.agg_keys = stdin_->agg_keys;
    t->t.agg_rows = machine_agg_row_count(t->t.agg_keys.empty() ? nullptr : t->t.agg_keys.data(), t->t.agg_leaves.size() / 8);
    t->t.leaf_check = stdin_->leaf_check;
  } catch (...) {
    delete t;
    return c->ctx.fail(ZKSP_ERR_EXECUTOR, "executor: out of memory while tracing the guest");
  }
  if (!t->t.rec.error.empty()) {
    const int rc = c->ctx.fail(ZKSP_ERR_EXECUTOR, "executor: " + t->t.rec.error);
    delete t;
    return rc;
  }
  if (!t->t.rec.halted) {
    delete t;
    return c->ctx.fail(ZKSP_ERR_EXECUTOR, "executor: guest did not halt");
  }
  *out = t;
  return ZKSP_OK;
}

void zksp_mtrace_free(zksp_mtrace* t) { delete t; }

int zksp_mtrace_section(const zksp_mtrace* t, int which, const void** ptr, size_t* bytes) {
  if (!t || !ptr || !bytes) return ZKSP_ERR_INVALID_ARG;
  const MachineTrace& m = t->t;
  switch (which) {
    case ZKSP_MT_CYCLES: *ptr = m.cycles.data(); *bytes = m.cycles.size() * sizeof(CycleRec); break;
    case ZKSP_MT_KECCAK: *ptr = m.keccak.data(); *bytes = m.keccak.size() * sizeof(KeccakCall); break;
    case ZKSP_MT_MEMFINAL: *ptr = m.memfinal.data(); *bytes = m.memfinal.size() * sizeof(MemFinalRec); break;
    case ZKSP_MT_MULS: *ptr = m.muls.data(); *bytes = m.muls.size() * sizeof(MulRec); break;
    case ZKSP_MT_PROG_MULT: *ptr = m.prog_mult.data(); *bytes = m.prog_mult.size() * 4; break;
    case ZKSP_MT_ALU_IDX: *ptr = m.alu_idx.data(); *bytes = m.alu_idx.size() * 4; break;
    case ZKSP_MT_SUB_IDX: *ptr = m.sub_idx.data(); *bytes = m.sub_idx.size() * 4; break;
    case ZKSP_MT_BW_IDX: *ptr = m.bw_idx.data(); *bytes = m.bw_idx.size() * 4; break;
    case ZKSP_MT_ECALL_IDX: *ptr = m.ecall_idx.data(); *bytes = m.ecall_idx.size() * 4; break;
    case ZKSP_MT_DIV_IDX: *ptr = m.div_idx.data(); *bytes = m.div_idx.size() * 4; break;
    case ZKSP_MT_PROGRAM: *ptr = t->prog->rows.data(); *bytes = t->prog->rows.size() * sizeof(ProgramRow); break;
    case ZKSP_MT_IMAGE: *ptr = t->prog->image.data(); *bytes = t->prog->image.size() * sizeof(ImageRow); break;
    case ZKSP_MT_PUBLIC_VALUES: *ptr = m.rec.public_values.data(); *bytes = m.rec.public_values.size(); break;
    case ZKSP_MT_LEAF_P2_ROWS: *ptr = m.leaf_check ? m.leaf_check->p2_rows.data() : nullptr; *bytes = m.leaf_check ? m.leaf_check->p2_rows.size() * 4 : 0; break;
    case ZKSP_MT_LEAF_QR_ROWS: *ptr = m.leaf_check ? m.leaf_check->qr_rows.data() : nullptr; *bytes = m.leaf_check ? m.leaf_check->qr_rows.size() * 4 : 0; break;
    case ZKSP_MT_LEAF_TR_ROWS: *ptr = m.leaf_check ? m.leaf_check->tr_rows.data() : nullptr; *bytes = m.leaf_check ? m.leaf_check->tr_rows.size() * 4 : 0; break;
    case ZKSP_MT_LEAF_PUB_TUPLES: *ptr = m.leaf_check ? m.leaf_check->pub_tuples.data() : nullptr; *bytes = m.leaf_check ? m.leaf_check->pub_tuples.size() * 4 : 0; break;
    default: return ZKSP_ERR_INVALID_ARG;
  }
  return ZKSP_OK;
}

int zksp_mtrace_info(const zksp_mtrace* t, zksp_mtrace_info_t* info) {
  if (!t || !info) return ZKSP_ERR_INVALID_ARG;
  memset(info, 0, sizeof *info);
  info->cycles = t->t.rec.cycles;
  info->memory_ops = t->t.rec.memory_ops;
  info->exit_code = t->t.rec.exit_code;
  info->entry = t->prog->entry;
  info->log_prog = (uint32_t)t->prog->log_prog;
  info->log_image = (uint32_t)t->prog->log_image;
  info->keccak_mode = (uint32_t)t->prog->keccak_mode;
  memcpy(info->pv_digest, t->t.rec.pv_digest.data(), 32);
  memcpy(info->deferred_digest, t->t.rec.deferred_digest.data(), 32);
  info->uninit_reads = t->t.rec.uninit_reads;
  return ZKSP_OK;
}

int zksp_mtrace_heights(const zksp_mtrace* t, int32_t* lh) {
  if (!t || !lh) return ZKSP_ERR_INVALID_ARG;
  int v[mach::kNumChips];
  machine_heights(*t->prog, t->t, v);
  for (int c = 0; c < mach::kNumChips; ++c) lh[c] = v[c];
  return ZKSP_OK;
}

const char* zksp_machine_chip_widths(int chip, int32_t* widths3) {
  if (chip < 0 || chip >= mach::kNumChips || !widths3) return nullptr;
  const mach::ChipDef& d = mach::chip_def(chip);
  widths3[0] = d.prep_w; widths3[1] = d.main_w; widths3[2] = d.perm_width();
  return d.name;
}

int zksp_machine_cover_heights(const zksp_mtrace* const* traces, size_t n, int32_t* lh) {
  if (!traces || n == 0 || !lh) return ZKSP_ERR_INVALID_ARG;
  MachineCounts cover;
  for (size_t i = 0; i < n; ++i) {
    if (!traces[i] || traces[i]->prog != traces[0]->prog) return ZKSP_ERR_INVALID_ARG;
    cover.cover(traces[i]->t);
  }
  int v[mach::kNumChips];
  machine_heights(*traces[0]->prog, cover, v);
  for (int c = 0; c < mach::kNumChips; ++c) lh[c] = v[c];
  return ZKSP_OK;
}

size_t zksp_machine_body_words(const zksp_client* c, const int32_t* lh) {
  if (!c || !lh) return 0;
  int v[mach::kNumChips];
  for (int k = 0; k < mach::kNumChips; ++k) {
    if (lh[k] < 5 || lh[k] > 21) return 0;
    v[k] = lh[k];
  }
  return machine_proof_body_words(v, c->ctx.params.num_queries);
}

int zksp_hip_machine_load(zksp_client* c, const zksp_pk* pk, const zksp_mtrace* const* traces, size_t n) {
  if (!c || !pk || !traces || n == 0) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  if (!ctx->has_device()) return ctx->fail(ZKSP_ERR_NO_DEVICE, "machine_load: client has no GPU");
  ZKSP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  try {
    std::vector<const MachineTrace*> ts(n);
    for (size_t i = 0; i < n; ++i) {
      if (!traces[i] || traces[i]->prog != &pk->mprog) return ctx->fail(ZKSP_ERR_INVALID_ARG, "machine_load: trace of another key");
      ts[i] = &traces[i]->t;
    }
    return machine_load(ctx, pk->mprog, pk->mvk, ts.data(), n);
  } catch (...) {
    return ctx->fail(ZKSP_ERR_HIP, "machine_load: out of memory");
  }
}

int zksp_hip_release_workspace(zksp_client* c) {
  if (!c) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  if (!ctx->has_device()) return ZKSP_OK;
  ZKSP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->copy_stream) ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy_stream));
  ctx->mws.reset();
  ctx->retired.clear();
  if (ctx->arena) (void)hipFree(ctx->arena);
  ctx->arena = nullptr;
  ctx->arena_bytes = 0;
  for (int k = 0; k < 2; ++k) {
    if (ctx->rec_arena[k]) (void)hipFree(ctx->rec_arena[k]);
    ctx->rec_arena[k] = nullptr;
    ctx->rec_bytes[k] = 0;
  }
  for (int k = 0; k < 2; ++k) {
    if (ctx->h_stage2[k]) (void)hipHostFree(ctx->h_stage2[k]);
    ctx->h_stage2[k] = nullptr;
    ctx->h_stage2_words[k] = 0;
  }
  ctx->body_free = nullptr;
  return ZKSP_OK;
}

int zksp_hip_machine_prove(zksp_client* c) {
  if (!c) return ZKSP_ERR_INVALID_ARG;
  if (!c->ctx.has_device()) return c->ctx.fail(ZKSP_ERR_NO_DEVICE, "machine_prove: client has no GPU");
  ZKSP_HIP_CHECK(&c->ctx, hipSetDevice(c->ctx.device));
  return machine_prove_resident(&c->ctx);
}

int zksp_hip_machine_fetch_bodies(zksp_client* c, uint32_t* out, size_t cap_words) {
  if (!c || !out) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  MachineWorkspace* w = ctx->mws.get();
  if (!w || w->n == 0) return ctx->fail(ZKSP_ERR_INVALID_ARG, "machine_fetch_bodies: no batch");
  const size_t words = (size_t)w->n * w->body_words;
  if (cap_words < words) return ctx->fail(ZKSP_ERR_INVALID_ARG, "machine_fetch_bodies: buffer too small");
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(out, w->body, words * 4, hipMemcpyDeviceToHost, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ZKSP_OK;
}

int zksp_hip_machine_fetch_roots(zksp_client* c, uint32_t* out, size_t cap_words) {
  if (!c || !out) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  MachineWorkspace* w = ctx->mws.get();
  if (!w || w->n == 0) return ctx->fail(ZKSP_ERR_INVALID_ARG, "machine_fetch_roots: no batch");
  if (cap_words < (size_t)w->n * 8) return ctx->fail(ZKSP_ERR_INVALID_ARG, "machine_fetch_roots: buffer too small");
  // the main-trace commitment is the first 8 words of every proof body
  ZKSP_HIP_CHECK(ctx, hipMemcpy2DAsync(out, 32, w->body, w->body_words * 4, 32, (size_t)w->n, hipMemcpyDeviceToHost, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ZKSP_OK;
}

int zksp_hip_machine_fetch_stage(zksp_client* c, int chip, int stage, size_t proof_index, uint32_t* out, size_t cap_words) {
  if (!c || !out || chip < 0 || chip >= mach::kNumChips || stage < 0 || stage > 2) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  MachineWorkspace* w = ctx->mws.get();
  if (!w || w->n == 0 || proof_index >= (size_t)w->n) return ctx->fail(ZKSP_ERR_INVALID_ARG, "machine_fetch_stage: no such resident proof");
  const size_t words = (size_t)w->mat[chip][stage].w << w->logh[chip];
  if (cap_words < words) return ctx->fail(ZKSP_ERR_INVALID_ARG, "machine_fetch_stage: buffer too small");
  ZKSP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(out, w->mat[chip][stage].tr + proof_index * words, words * 4, hipMemcpyDeviceToHost, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < words; ++i) out[i] = Fp::raw(out[i]).to_canonical();
  return ZKSP_OK;
}

int zksp_hip_machine_fetch_challenges(zksp_client* c, size_t proof_index, uint32_t* out) {
  if (!c || !out) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  MachineWorkspace* w = ctx->mws.get();
  if (!w || w->n == 0 || proof_index >= (size_t)w->n) return ctx->fail(ZKSP_ERR_INVALID_ARG, "machine_fetch_challenges: no such resident proof");
  ZKSP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // alpha's buffer is reused for the last transcript squeeze: the powers table still starts with 1, alpha
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(out, w->bus_ch + proof_index * 8, 32, hipMemcpyDeviceToHost, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(out + 8, w->alpha_pows + proof_index * w->alpha_stride + 4, 16, hipMemcpyDeviceToHost, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(out + 12, w->zeta + proof_index * 4, 16, hipMemcpyDeviceToHost, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(out + 16, w->cum + proof_index * 4 * mach::kNumChips, 16 * mach::kNumChips, hipMemcpyDeviceToHost, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 16 + 4 * mach::kNumChips; ++i) out[i] = Fp::raw(out[i]).to_canonical();
  return ZKSP_OK;
}

int zksp_machine_proof_from_body(const zksp_pk* pk, const zksp_mtrace* t, const int32_t* log_heights, const uint32_t* body,
                                 size_t body_words, zksp_proof** out) {
  if (!pk || !t || !body || !out) return ZKSP_ERR_INVALID_ARG;
  int lh[mach::kNumChips];
  if (log_heights) {
    for (int c = 0; c < mach::kNumChips; ++c) {
      if (log_heights[c] < 5 || log_heights[c] > 21) return ZKSP_ERR_INVALID_ARG;
      lh[c] = log_heights[c];
    }
    if (!machine_fits(t->t, lh)) return ZKSP_ERR_INVALID_ARG;
  } else {
    machine_heights(*t->prog, t->t, lh);
  }
  uint32_t handover[mach::kNumCpuInst - 1];
  for (int k = 1; k < mach::kNumCpuInst; ++k) handover[k - 1] = machine_handover_pc(*t->prog, t->t, lh, k);
  return machine_proof_from_parts(pk, t->t.rec, lh, handover, t->t.agg_leaves, t->t.agg_keys, t->t.leaf_check.get(), body, body_words, out);
}

int zksp_stdin_set_aggregation(zksp_stdin* s, const uint32_t* leaves, size_t n) {
  if (n == 1 || (n & (n - 1))) return ZKSP_ERR_INVALID_ARG;  // the leaves of a full tree: a power of two of them
  return zksp_stdin_set_aggregation_keyed(s, nullptr, leaves, n);
}

int zksp_stdin_set_aggregation_keyed(zksp_stdin* s, const uint32_t* keys, const uint32_t* digests, size_t n) {
  if (!s || (n && !digests) || n == 1 || n > ((size_t)1 << 20)) return ZKSP_ERR_INVALID_ARG;
  for (size_t i = 0; i < 8 * n; ++i)
    if (digests[i] >= kP) return ZKSP_ERR_INVALID_ARG;
  if (n && machine_agg_row_count(keys, n) == SIZE_MAX) return ZKSP_ERR_INVALID_ARG;  // (every ancestor needs both children)
  try {
    s->agg_leaves.assign(digests, digests + 8 * n);
    if (keys) s->agg_keys.assign(keys, keys + n);
    else s->agg_keys.clear();
  } catch (...) {
    return ZKSP_ERR_INVALID_ARG;
  }
  return ZKSP_OK;
}

int zksp_proof_aggregation(const zksp_proof* p, uint32_t* n_leaves, uint32_t* root8) {
  if (!p || !n_leaves || !root8 || p->version != mach::kMachineVersion) return ZKSP_ERR_INVALID_ARG;
  *n_leaves = p->mhdr.agg_n;
  memcpy(root8, p->mhdr.agg_root, 32);
  return ZKSP_OK;
}

int zksp_verify_aggregate(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const uint32_t* leaves, size_t n) {
  return zksp_verify_aggregate_keyed(c, p, vk, nullptr, leaves, n);
}

int zksp_verify_aggregate_keyed(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const uint32_t* keys, const uint32_t* digests,
                                size_t n) {
  if (!c || !p || !vk || (n && !digests)) return ZKSP_ERR_INVALID_ARG;
  if (c->ctx.params.proof_mode != ZKSP_PROOF_MACHINE || p->version != mach::kMachineVersion)
    return c->ctx.fail(ZKSP_ERR_VERIFY, "verify: not a machine proof");
  std::string err;
  int rc;
  try {
    rc = verify_machine_proof(p->bytes.data(), p->bytes.size(), vk->machine, c->ctx.params.num_queries, c->ctx.params.pow_bits, &err,
                              digests, n, keys);
  } catch (...) {
    return c->ctx.fail(ZKSP_ERR_VERIFY, "verify: out of memory");
  }
  if (rc) return c->ctx.fail(rc, "verify: " + err);
  return ZKSP_OK;
}

// ---- leaf-proof check (stage 2b) ----
// Verifies `leaf` - completely, or (stub_only) everything but its query phase - and leaves the records / the public tuples of a
// proof ABOUT that verification in *out.  `index`: the leaf's place among the leaves checked beside one run.  own / n_own: the
// public tuples `leaf` itself closes its buses with (a leaf that checks leaves of its own: a node of a recursion tree).
// (no access to the client's error state: safe to run for several leaves at once)
static int leaf_check_run(const zksp_client* c, const zksp_proof* leaf, const zksp_vk* leaf_vk, std::shared_ptr<LeafCheckLog>* out,
                          uint32_t index, const uint32_t* own, size_t n_own, bool stub_only, std::string* err, unsigned max_threads = 0) {
  if (c->ctx.params.proof_mode != ZKSP_PROOF_MACHINE || leaf->version != mach::kMachineVersion) {
    *err = "leaf check: not a machine proof";
    return ZKSP_ERR_VERIFY;
  }
  if (leaf->mhdr.agg_n) { *err = "leaf check: the leaf proof carries an aggregation payload"; return ZKSP_ERR_UNSUPPORTED; }
  if (c->ctx.params.num_queries > mach::kLeafMaxQueries || index >= 4096) {
    *err = "leaf check: too many queries or leaves for the tag space";
    return ZKSP_ERR_UNSUPPORTED;
  }
  int rc;
  try {
    auto log = std::make_shared<LeafCheckLog>();
    log->leaf_index = index;
    log->n_leaves = 1;
    // a stub is recognised by its length; a complete proof is cut down to one where only the statement is wanted
    const size_t stub_len = leaf->mhdr.body_offset + machine_proof_body_words(leaf->mhdr.logh, 0) * 4;
    const bool is_stub = leaf->bytes.size() == stub_len;
    if (is_stub && !stub_only) { *err = "leaf check: a proof stub has no query phase to prove"; return ZKSP_ERR_INVALID_ARG; }
    std::string verr;
    rc = verify_machine_proof(leaf->bytes.data(), stub_only ? stub_len : leaf->bytes.size(), leaf_vk->machine, c->ctx.params.num_queries,
                              c->ctx.params.pow_bits, &verr, nullptr, 0, nullptr, own, n_own, log.get(), stub_only, max_threads);
    if (rc == 0) *out = std::move(log);
    else *err = "leaf check: the leaf proof does not verify: " + verr;
  } catch (...) {
    *err = "leaf check: out of memory";
    return ZKSP_ERR_VERIFY;
  }
  return rc;
}
static int leaf_check_of(zksp_client* c, const zksp_proof* leaf, const zksp_vk* leaf_vk, std::shared_ptr<LeafCheckLog>* out,
                         uint32_t index = 0, const uint32_t* own = nullptr, size_t n_own = 0, bool stub_only = false) {
  std::string err;
  const int rc = leaf_check_run(c, leaf, leaf_vk, out, index, own, n_own, stub_only, &err);
  if (rc) return c->ctx.fail(rc, err);
  return ZKSP_OK;
}

int zksp_proof_stub(const zksp_proof* p, zksp_proof** out) {
  if (!p || !out || p->version != mach::kMachineVersion) return ZKSP_ERR_INVALID_ARG;
  const size_t stub_len = p->mhdr.body_offset + machine_proof_body_words(p->mhdr.logh, 0) * 4;
  if (p->bytes.size() < stub_len) return ZKSP_ERR_INVALID_ARG;
  zksp_proof* q = new (std::nothrow) zksp_proof();
  if (!q) return ZKSP_ERR_INVALID_ARG;
  try {
    *q = *p;
    q->bytes.resize(stub_len);
  } catch (...) {
    delete q;
    return ZKSP_ERR_INVALID_ARG;
  }
  *out = q;
  return ZKSP_OK;
}

int zksp_stdin_set_verified_leaf(zksp_client* c, zksp_stdin* s, const zksp_proof* leaf, const zksp_vk* leaf_vk) {
  if (!c || !s) return ZKSP_ERR_INVALID_ARG;
  if (!leaf) { s->leaf_check.reset(); s->deferred.clear(); s->statement.clear(); return ZKSP_OK; }
  if (!leaf_vk) return ZKSP_ERR_INVALID_ARG;
  if (!s->deferred.empty()) return c->ctx.fail(ZKSP_ERR_INVALID_ARG, "leaf check: the stdin carries deferred leaves; defer all of a run's leaves or none");
  std::shared_ptr<LeafCheckLog> log;
  const int rc = leaf_check_of(c, leaf, leaf_vk, &log);
  if (rc) return rc;
  s->leaf_check = std::move(log);
  return ZKSP_OK;
}

int zksp_stdin_add_verified_node(zksp_client* c, zksp_stdin* s, const zksp_proof* leaf, const zksp_vk* leaf_vk, const uint32_t* own,
                                 size_t n_own) {
  if (!c || !s || !leaf || !leaf_vk || (n_own && !own)) return ZKSP_ERR_INVALID_ARG;
  if (!s->deferred.empty()) return c->ctx.fail(ZKSP_ERR_INVALID_ARG, "leaf check: the stdin carries deferred leaves; defer all of a run's leaves or none");
  const uint32_t have = s->leaf_check ? s->leaf_check->n_leaves : 0;
  std::shared_ptr<LeafCheckLog> one;
  const int rc = leaf_check_of(c, leaf, leaf_vk, &one, have, own, n_own);
  if (rc) return rc;
  try {
    std::shared_ptr<LeafCheckLog> all;
    if (s->leaf_check) {
      all = std::make_shared<LeafCheckLog>(*s->leaf_check);
      all->append(*one);
    } else {
      all = std::move(one);  // (the first leaf: its log as it is)
    }
    all->leaf_index = 0;
    all->n_leaves = have + 1;
    s->leaf_check = std::move(all);
  } catch (...) {
    return c->ctx.fail(ZKSP_ERR_VERIFY, "leaf check: out of memory");
  }
  return ZKSP_OK;
}

int zksp_stdin_add_verified_leaf(zksp_client* c, zksp_stdin* s, const zksp_proof* leaf, const zksp_vk* leaf_vk) {
  return zksp_stdin_add_verified_node(c, s, leaf, leaf_vk, nullptr, 0);
}

int zksp_stdin_add_verified_leaves(zksp_client* c, zksp_stdin* s, const zksp_proof* const* leaves, const zksp_vk* const* leaf_vks,
                                   const uint32_t* const* own, const size_t* n_own, size_t n) {
  if (!c || !s || !leaves || !leaf_vks || !n) return ZKSP_ERR_INVALID_ARG;
  for (size_t k = 0; k < n; ++k)
    if (!leaves[k] || !leaf_vks[k] || (own && n_own && n_own[k] && !own[k])) return ZKSP_ERR_INVALID_ARG;
  if (!s->deferred.empty()) return c->ctx.fail(ZKSP_ERR_INVALID_ARG, "leaf check: the stdin carries deferred leaves; defer all of a run's leaves or none");
  const uint32_t have = s->leaf_check ? s->leaf_check->n_leaves : 0;
  // the leaves are independent of one another: verified and logged side by side (each on several threads of its own for the
  // queries), appended in the order given
  std::vector<std::shared_ptr<LeafCheckLog>> logs(n);
  std::vector<int> rcs(n, ZKSP_OK);
  std::vector<std::string> errs(n);
  auto run = [&](size_t k) noexcept {
    try {
      rcs[k] = leaf_check_run(c, leaves[k], leaf_vks[k], &logs[k], have + (uint32_t)k, own ? own[k] : nullptr, own && n_own ? n_own[k] : 0,
                              false, &errs[k]);
    } catch (...) {
      rcs[k] = ZKSP_ERR_VERIFY;
    }
  };
  {
    struct Joiner {
      std::vector<std::thread> th;
      ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
    } pool;
    for (size_t k = 1; k < n; ++k) {
      try {
        pool.th.emplace_back(run, k);
      } catch (...) {
        run(k);
      }
    }
    run(0);
  }
  for (size_t k = 0; k < n; ++k)
    if (rcs[k]) return c->ctx.fail(rcs[k], errs[k].empty() ? std::string("leaf check: out of memory") : errs[k]);
  try {
    std::shared_ptr<LeafCheckLog> all;
    if (!s->leaf_check && n == 1) {
      all = std::move(logs[0]);
    } else {
      all = s->leaf_check ? std::make_shared<LeafCheckLog>(*s->leaf_check) : std::make_shared<LeafCheckLog>();
      std::vector<const LeafCheckLog*> parts(n);
      for (size_t k = 0; k < n; ++k) parts[k] = logs[k].get();
      all->append_all(parts.data(), n);
    }
    all->leaf_index = 0;
    all->n_leaves = have + (uint32_t)n;
    s->leaf_check = std::move(all);
  } catch (...) {
    return c->ctx.fail(ZKSP_ERR_VERIFY, "leaf check: out of memory");
  }
  return ZKSP_OK;
}

int zksp_leaf_public_at(zksp_client* c, const zksp_proof* leaf, const zksp_vk* leaf_vk, uint32_t index, const uint32_t* own, size_t n_own,
                        uint32_t* out, size_t cap_words, size_t* n_tuples) {
  if (!c || !leaf || !leaf_vk || !n_tuples || (n_own && !own)) return ZKSP_ERR_INVALID_ARG;
  std::shared_ptr<LeafCheckLog> log;
  const int rc = leaf_check_of(c, leaf, leaf_vk, &log, index, own, n_own, /*stub_only=*/true);
  if (rc) return rc;
  *n_tuples = log->pub_tuples.size() / mach::kPubTupleWords;
  if (out) {
    if (cap_words < log->pub_tuples.size()) return c->ctx.fail(ZKSP_ERR_INVALID_ARG, "leaf_public: buffer too small");
    memcpy(out, log->pub_tuples.data(), log->pub_tuples.size() * 4);
  }
  return ZKSP_OK;
}

// the statements of several leaves beside one run, one after the other, in the order the leaves were added
static int leaves_statement(zksp_client* c, const zksp_proof* const* leaves, const zksp_vk* const* leaf_vks, size_t n,
                            std::vector<uint32_t>* out) {
  out->clear();
  for (size_t k = 0; k < n; ++k) {
    if (!leaves[k] || !leaf_vks[k]) return ZKSP_ERR_INVALID_ARG;
    std::shared_ptr<LeafCheckLog> one;
    const int rc = leaf_check_of(c, leaves[k], leaf_vks[k], &one, (uint32_t)k, nullptr, 0, /*stub_only=*/true);
    if (rc) return rc;
    out->insert(out->end(), one->pub_tuples.begin(), one->pub_tuples.end());
  }
  return ZKSP_OK;
}

int zksp_leaves_public(zksp_client* c, const zksp_proof* const* leaves, const zksp_vk* const* leaf_vks, size_t n, uint32_t* out,
                       size_t cap_words, size_t* n_tuples) {
  if (!c || !leaves || !leaf_vks || !n || !n_tuples) return ZKSP_ERR_INVALID_ARG;
  std::vector<uint32_t> st;
  int rc;
  try {
    rc = leaves_statement(c, leaves, leaf_vks, n, &st);
  } catch (...) {
    return c->ctx.fail(ZKSP_ERR_VERIFY, "leaf check: out of memory");
  }
  if (rc) return rc;
  *n_tuples = st.size() / mach::kPubTupleWords;
  if (out) {
    if (cap_words < st.size()) return c->ctx.fail(ZKSP_ERR_INVALID_ARG, "leaves_public: buffer too small");
    memcpy(out, st.data(), st.size() * 4);
  }
  return ZKSP_OK;
}

int zksp_verify_with_leaves(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const zksp_proof* const* leaves,
                            const zksp_vk* const* leaf_vks, size_t n) {
  if (!c || !p || !vk || !leaves || !leaf_vks || !n) return ZKSP_ERR_INVALID_ARG;
  std::vector<uint32_t> st;
  int rc;
  try {
    rc = leaves_statement(c, leaves, leaf_vks, n, &st);
  } catch (...) {
    return c->ctx.fail(ZKSP_ERR_VERIFY, "leaf check: out of memory");
  }
  if (rc) return rc;
  return zksp_verify_public(c, p, vk, st.data(), st.size() / mach::kPubTupleWords);
}

int zksp_zeta_program_selftest(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const uint32_t* own, size_t n_own,
                               uint32_t info[8]) {
  if (!c || !p || !vk || !info || (n_own && !own)) return ZKSP_ERR_INVALID_ARG;
  if (c->ctx.params.proof_mode != ZKSP_PROOF_MACHINE || p->version != mach::kMachineVersion)
    return c->ctx.fail(ZKSP_ERR_VERIFY, "zeta program: not a machine proof");
  const size_t stub_len = p->mhdr.body_offset + machine_proof_body_words(p->mhdr.logh, 0) * 4;
  const bool is_stub = p->bytes.size() == stub_len;
  ZetaSelfTest zst;
  std::string err;
  int rc;
  try {
    rc = verify_machine_proof(p->bytes.data(), p->bytes.size(), vk->machine, c->ctx.params.num_queries, c->ctx.params.pow_bits, &err,
                              nullptr, 0, nullptr, own, n_own, nullptr, is_stub, 0, &zst);
  } catch (...) {
    return c->ctx.fail(ZKSP_ERR_VERIFY, "zeta program: out of memory");
  }
  memset(info, 0, 32);
  info[0] = zst.n_ops; info[1] = zst.n_cells; info[2] = zst.n_inputs; info[3] = zst.n_consts;
  info[4] = (uint32_t)zst.mismatch_chip;
  info[5] = zst.max_reads; info[6] = zst.inputs_read;
  if (rc) return c->ctx.fail(rc, "zeta program: " + err);
  return ZKSP_OK;
}

int zksp_stdin_defer_verified_leaves(zksp_client* c, zksp_stdin* s, const zksp_proof* const* leaves, const zksp_vk* const* leaf_vks,
                                     const uint32_t* const* own, const size_t* n_own, size_t n) {
  if (!c || !s || !leaves || !leaf_vks || !n) return ZKSP_ERR_INVALID_ARG;
  for (size_t k = 0; k < n; ++k)
    if (!leaves[k] || !leaf_vks[k] || (own && n_own && n_own[k] && !own[k])) return ZKSP_ERR_INVALID_ARG;
  if (s->leaf_check) return c->ctx.fail(ZKSP_ERR_INVALID_ARG, "leaf check: the stdin already carries verified leaves; defer all of a run's leaves or none");
  try {
    for (size_t k = 0; k < n; ++k) {
      zksp_stdin::Deferred d{leaves[k], leaf_vks[k], {}};
      if (own && n_own && n_own[k]) d.own.assign(own[k], own[k] + n_own[k] * mach::kPubTupleWords);
      s->deferred.push_back(std::move(d));
    }
  } catch (...) {
    return c->ctx.fail(ZKSP_ERR_INVALID_ARG, "leaf check: out of memory");
  }
  return ZKSP_OK;
}

int stdin_resolve_deferred(const zksp_client* c, zksp_stdin* s, std::string* err, unsigned budget) {
  if (s->deferred.empty()) return ZKSP_OK;
  const size_t n = s->deferred.size();
  if (budget == 0) budget = 1;
  try {
    std::vector<std::shared_ptr<LeafCheckLog>> logs(n);
    std::vector<int> rcs(n, ZKSP_OK);
    std::vector<std::string> errs(n);
    const bool side_by_side = budget >= n && n > 1;
    const unsigned per_leaf = side_by_side ? std::max(1u, budget / (unsigned)n) : budget;
    auto run = [&](size_t k) noexcept {
      try {
        const zksp_stdin::Deferred& d = s->deferred[k];
        rcs[k] = leaf_check_run(c, d.leaf, d.vk, &logs[k], (uint32_t)k, d.own.empty() ? nullptr : d.own.data(),
                                d.own.size() / mach::kPubTupleWords, false, &errs[k], per_leaf);
      } catch (...) {
        rcs[k] = ZKSP_ERR_VERIFY;
      }
    };
    if (side_by_side) {
      struct Joiner {
        std::vector<std::thread> th;
        ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
      } pool;
      for (size_t k = 1; k < n; ++k) {
        try {
          pool.th.emplace_back(run, k);
        } catch (...) {
          run(k);
        }
      }
      run(0);
    } else {
      for (size_t k = 0; k < n; ++k) {
        run(k);
        if (rcs[k]) break;
      }
    }
    for (size_t k = 0; k < n; ++k)
      if (rcs[k]) {
        *err = errs[k].empty() ? std::string("leaf check: out of memory") : errs[k];
        return rcs[k];
      }
    std::shared_ptr<LeafCheckLog> all;
    if (n == 1) {
      all = std::move(logs[0]);
    } else {
      all = std::make_shared<LeafCheckLog>();
      std::vector<const LeafCheckLog*> parts(n);
      for (size_t k = 0; k < n; ++k) parts[k] = logs[k].get();
      all->append_all(parts.data(), n);
    }
    all->leaf_index = 0;
    all->n_leaves = (uint32_t)n;
    s->statement = all->pub_tuples;
    s->leaf_check = std::move(all);
    s->deferred.clear();
  } catch (...) {
    *err = "leaf check: out of memory";
    return ZKSP_ERR_VERIFY;
  }
  return ZKSP_OK;
}

int zksp_stdin_public_tuples(const zksp_stdin* s, uint32_t* out, size_t cap_words, size_t* n_tuples) {
  if (!s || !n_tuples) return ZKSP_ERR_INVALID_ARG;
  // (the attached checks' list; after proving consumed them - or made them, if they were deferred - the copy kept of it)
  const std::vector<uint32_t>& st = s->leaf_check ? s->leaf_check->pub_tuples : s->statement;
  *n_tuples = st.size() / mach::kPubTupleWords;
  if (out) {
    if (cap_words < st.size()) return ZKSP_ERR_INVALID_ARG;
    if (!st.empty()) memcpy(out, st.data(), st.size() * 4);
  }
  return ZKSP_OK;
}

int zksp_leaf_public(zksp_client* c, const zksp_proof* leaf, const zksp_vk* leaf_vk, uint32_t* out, size_t cap_words, size_t* n_tuples) {
  return zksp_leaf_public_at(c, leaf, leaf_vk, 0, nullptr, 0, out, cap_words, n_tuples);
}

int zksp_verify_public(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const uint32_t* tuples, size_t n_tuples) {
  if (!c || !p || !vk || (n_tuples && !tuples)) return ZKSP_ERR_INVALID_ARG;
  if (c->ctx.params.proof_mode != ZKSP_PROOF_MACHINE || p->version != mach::kMachineVersion)
    return c->ctx.fail(ZKSP_ERR_VERIFY, "verify: not a machine proof");
  std::string err;
  int rc;
  try {
    rc = verify_machine_proof(p->bytes.data(), p->bytes.size(), vk->machine, c->ctx.params.num_queries, c->ctx.params.pow_bits, &err,
                              nullptr, 0, nullptr, tuples, n_tuples);
  } catch (...) {
    return c->ctx.fail(ZKSP_ERR_VERIFY, "verify: out of memory");
  }
  if (rc) return c->ctx.fail(rc, "verify: " + err);
  return ZKSP_OK;
}

int zksp_verify_with_leaf(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const zksp_proof* leaf, const zksp_vk* leaf_vk) {
  if (!c || !p || !vk || !leaf || !leaf_vk) return ZKSP_ERR_INVALID_ARG;
  return zksp_verify_with_leaves(c, p, vk, &leaf, &leaf_vk, 1);
}

int zksp_proof_public_tuples(const zksp_proof* p, uint32_t* n_tuples, uint32_t* digest8) {
  if (!p || !n_tuples || !digest8 || p->version != mach::kMachineVersion) return ZKSP_ERR_INVALID_ARG;
  *n_tuples = p->mhdr.pub_n;
  memcpy(digest8, p->mhdr.pub_digest, 32);
  return ZKSP_OK;
}

int zksp_vk_machine(const zksp_vk* vk, uint32_t* prep_root8, uint32_t* digest8) {
  if (!vk || !prep_root8 || !digest8) return ZKSP_ERR_INVALID_ARG;
  memcpy(prep_root8, vk->machine.prep_root, 32);
  memcpy(digest8, vk->machine.digest, 32);
  return ZKSP_OK;
}

}  // extern "C"
