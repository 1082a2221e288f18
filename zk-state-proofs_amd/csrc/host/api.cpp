// C ABI of include/zksp.h.  Nothing here throws across the boundary.
#include "../../../include/zksp.h"
#ifdef ZKSP_COMPONENT
#include "../../../include/zksp_component.h"
#endif

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <thread>

#include "context.hpp"
#ifdef ZKSP_COMPONENT
#include "prover.hpp"
#include "verifier.hpp"
#endif
#include "machine_defs.hpp"
#include "host_hash.hpp"

#include "api_types.hpp"

using namespace zksp;

namespace {

void compute_vk_digest(const ElfImage& elf, uint32_t out[8]) {
  // vk = keccak256(sha256(ELF) || chip/parameter tag), 8 words reduced into the field
  static const char tag[] = "zksp/vk/keccak-chip/w2633/c3182/blowup2/v1";
  std::vector<uint8_t> buf(elf.sha256.begin(), elf.sha256.end());
  buf.insert(buf.end(), tag, tag + sizeof(tag) - 1);
  uint8_t dg[32];
  keccak256(buf.data(), buf.size(), dg);
  for (int i = 0; i < 8; ++i) {
    uint32_t w;
    memcpy(&w, dg + 4 * i, 4);
    out[i] = w % kP;
  }
}

}  // namespace

extern "C" {

int zksp_client_new(const zksp_options* opts, zksp_client** out) {
  if (!out) return ZKSP_ERR_INVALID_ARG;
  *out = nullptr;
  zksp_client* c = new (std::nothrow) zksp_client();
  if (!c) return ZKSP_ERR_INVALID_ARG;
  Context& ctx = c->ctx;
  int dev = opts ? opts->device_ordinal : 0;
  if (opts) {
    if (opts->keccak_mode) ctx.params.keccak_mode = opts->keccak_mode;
    if (opts->num_queries) ctx.params.num_queries = opts->num_queries;
    if (opts->pow_bits != 0xffffffffu) ctx.params.pow_bits = opts->pow_bits;
    if (opts->max_batch) ctx.params.max_batch = opts->max_batch;
    if (opts->proof_mode) ctx.params.proof_mode = opts->proof_mode;
  }
  // Run-time backend selection, the counterpart of SP1_PROVER (reference .env.example:1-2) next to
  // the compile-time features of prover/Cargo.toml:32-35.  ZKSP_PROVER = "hip" | "local": prove on the
  // GPU (an ordinal of -1 becomes 0); "host": executor/verifier only, no GPU touched; "cpu", "mock",
  // "network": refused, this library has no CPU, mock or remote proving backend.
  if (const char* e = getenv("ZKSP_PROVER")) {
    const std::string v(e);
    if (v == "hip" || v == "local") { if (dev < 0) dev = 0; }
    else if (v == "host") dev = -1;
    else if (!v.empty()) { delete c; return ZKSP_ERR_UNSUPPORTED; }
  }
  if (ctx.params.keccak_mode != 1 && ctx.params.keccak_mode != 2) { delete c; return ZKSP_ERR_INVALID_ARG; }
  if (ctx.params.pow_bits > 30 || ctx.params.num_queries > 4096) { delete c; return ZKSP_ERR_INVALID_ARG; }
  if (ctx.params.proof_mode != ZKSP_PROOF_MACHINE && ctx.params.proof_mode != ZKSP_PROOF_KECCAK_CHIP) { delete c; return ZKSP_ERR_INVALID_ARG; }
#ifndef ZKSP_COMPONENT
  // (the round-1 keccak-chip component path is a build switch: `ZKSP_COMPONENT=1 python -m zk-state-proofs_amd.build` makes
  // libzksp_component.so, which has it; include/zksp_component.h)
  if (ctx.params.proof_mode == ZKSP_PROOF_KECCAK_CHIP) { delete c; return ZKSP_ERR_UNSUPPORTED; }
#endif
  if (dev >= 0) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || dev >= count) { delete c; return ZKSP_ERR_NO_DEVICE; }
    if (hipSetDevice(dev) != hipSuccess) { delete c; return ZKSP_ERR_NO_DEVICE; }
    ctx.device = dev;
    bool ok = hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc(&ctx.d_consts, sizeof(P2Consts)) == hipSuccess;
    ok = ok && hipMemcpy(ctx.d_consts, &host_p2_consts(), sizeof(P2Consts), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipEventCreate(&ctx.timer_a) == hipSuccess && hipEventCreate(&ctx.timer_b) == hipSuccess;
    ok = ok && lde_configure() == (int)hipSuccess;
    if (!ok) { delete c; return ZKSP_ERR_HIP; }
  }
  *out = c;
  return ZKSP_OK;
}

void zksp_client_free(zksp_client* c) { delete c; }

const char* zksp_last_error(const zksp_client* c) { return c ? c->ctx.error.c_str() : "null client"; }

int zksp_setup(zksp_client* c, const uint8_t* elf, size_t elf_len, zksp_pk** pk, zksp_vk** vk) {
  if (!c || !elf || !pk || !vk) return ZKSP_ERR_INVALID_ARG;
  zksp_pk* p = new (std::nothrow) zksp_pk();
  zksp_vk* v = new (std::nothrow) zksp_vk();
  if (!p || !v) { delete p; delete v; return ZKSP_ERR_INVALID_ARG; }
  std::string e = load_elf(elf, elf_len, &p->elf);
  if (!e.empty()) { delete p; delete v; return c->ctx.fail(ZKSP_ERR_ELF, "setup: " + e); }
  if (p->elf.keccakf_entries.empty() && c->ctx.params.proof_mode == ZKSP_PROOF_KECCAK_CHIP) {
    delete p; delete v;  // a machine proof of a guest without keccak is a keccak chip of padding rows only
    return c->ctx.fail(ZKSP_ERR_ELF, "setup: ELF has no keccakf symbol; the keccak chip has nothing to prove");
  }
  if (c->ctx.params.keccak_mode == (int)KeccakMode::kReplace) {
    // the precompile shape replaces calls of these addresses by the keccak chip: its key is only made for an ELF whose
    // functions of that name are keccak-f on the test states (executor.hpp check_keccakf_entries)
    const std::string ke = check_keccakf_entries(p->elf);
    if (!ke.empty()) { delete p; delete v; return c->ctx.fail(ZKSP_ERR_ELF, "setup: " + ke); }
  }
  {
    std::string me = build_machine_program(p->elf, (KeccakMode)c->ctx.params.keccak_mode, &p->mprog);
    if (!me.empty()) { delete p; delete v; return c->ctx.fail(ZKSP_ERR_ELF, "setup: " + me); }
  }
  compute_vk_digest(p->elf, p->vk_digest);
  memcpy(v->digest, p->vk_digest, 32);
  try {
    machine_host_setup(p->mprog, &p->mvk);  // commitment of the preprocessed Program / Image tables
  } catch (...) {
    delete p; delete v;
    return c->ctx.fail(ZKSP_ERR_ELF, "setup: out of memory while committing the preprocessed tables");
  }
  v->machine = p->mvk;
  *pk = p;
  *vk = v;
  return ZKSP_OK;
}
void zksp_pk_free(zksp_pk* pk) { delete pk; }
void zksp_vk_free(zksp_vk* vk) { delete vk; }
int zksp_vk_digest(const zksp_vk* vk, const uint8_t** ptr, size_t* len) {
  if (!vk || !ptr || !len) return ZKSP_ERR_INVALID_ARG;
  *ptr = reinterpret_cast<const uint8_t*>(vk->digest);
  *len = 32;
  return ZKSP_OK;
}

zksp_stdin* zksp_stdin_new(void) { return new (std::nothrow) zksp_stdin(); }
int zksp_stdin_write(zksp_stdin* s, const uint8_t* buf, size_t len) {
  if (!s || (!buf && len)) return ZKSP_ERR_INVALID_ARG;
  std::vector<uint8_t> e(8 + len);
  uint64_t l = len;
  memcpy(e.data(), &l, 8);  // bincode Vec<u8>: u64-LE length prefix
  if (len) memcpy(e.data() + 8, buf, len);
  s->entries.push_back(std::move(e));
  return ZKSP_OK;
}
void zksp_stdin_free(zksp_stdin* s) { delete s; }

int zksp_execute(zksp_client* c, const zksp_pk* pk, const zksp_stdin* stdin_, int keccak_mode, zksp_exec_report* report,
                 uint8_t* public_values, size_t pv_cap, char* stderr_buf, size_t stderr_cap) {
  if (!c || !pk || !stdin_ || !report || keccak_mode < 0 || keccak_mode > 2) return ZKSP_ERR_INVALID_ARG;
  ExecOptions o;
  o.keccak_mode = (KeccakMode)keccak_mode;
  o.want_hist = true;
  ExecutionRecord r = execute(pk->elf, stdin_->entries, o);
  memset(report, 0, sizeof *report);
  report->cycles = r.cycles;
  report->memory_ops = r.memory_ops;
  report->exit_code = r.exit_code;
  report->n_keccak = (uint32_t)r.keccak_events.size();
  report->pv_len = (uint32_t)r.public_values.size();
  memcpy(report->pv_digest, r.pv_digest.data(), 32);
  const int codes[6] = {0x00, 0x02, 0x10, 0x1a, 0xf0, 0xf1};
  for (int i = 0; i < 6; ++i) report->syscalls[i] = r.syscall_counts[codes[i]];
  for (size_t i = 0; i < r.opcode_hist.size() && i < 64; ++i) report->opcode_hist[i] = r.opcode_hist[i];
  if (public_values && pv_cap) memcpy(public_values, r.public_values.data(), std::min(pv_cap, r.public_values.size()));
  if (stderr_buf && stderr_cap) {
    size_t m = std::min(stderr_cap - 1, r.stderr_text.size());
    memcpy(stderr_buf, r.stderr_text.data(), m);
    stderr_buf[m] = 0;
  }
  if (!r.error.empty()) return c->ctx.fail(ZKSP_ERR_EXECUTOR, "executor: " + r.error);
  if (!r.halted) return c->ctx.fail(ZKSP_ERR_EXECUTOR, "executor: guest did not halt");
  return ZKSP_OK;
}
const char* zksp_opcode_name(int index) { return op_name(index); }

int zksp_execute_keccak(zksp_client* c, const zksp_pk* pk, const zksp_stdin* stdin_, uint64_t* states, size_t cap_perms,
                        size_t* n) {
  if (!c || !pk || !stdin_ || !n) return ZKSP_ERR_INVALID_ARG;
  ExecOptions o;
  o.keccak_mode = (KeccakMode)c->ctx.params.keccak_mode;
  ExecutionRecord r = execute(pk->elf, stdin_->entries, o);
  if (!r.error.empty() || !r.halted) return c->ctx.fail(ZKSP_ERR_EXECUTOR, "executor: " + r.error);
  *n = r.keccak_events.size();
  if (states)
    for (size_t i = 0; i < r.keccak_events.size() && i < cap_perms; ++i) memcpy(states + 25 * i, r.keccak_events[i].state_in, 200);
  return ZKSP_OK;
}

int zksp_get_params(const zksp_client* c, zksp_params* out) {
  if (!c || !out) return ZKSP_ERR_INVALID_ARG;
  out->trace_width = ka::kWidth;          // the keccak chip's own columns
  out->num_constraints = ka::kNumConstraints;
  out->num_queries = c->ctx.params.num_queries;
  out->pow_bits = c->ctx.params.pow_bits;
  out->max_batch = c->ctx.params.max_batch;
  return ZKSP_OK;
}
#ifdef ZKSP_COMPONENT
size_t zksp_proof_body_words(const zksp_client* c, int log_h) {
  if (!c || log_h < 1 || log_h > 26) return 0;
  return proof_body_words(log_h, c->ctx.params.num_queries);
}

// ---------------------------------------------------------------------------
// device-resident hot path
// ---------------------------------------------------------------------------
int zksp_hip_load_batch(zksp_client* c, int log_h, size_t n, size_t max_perms, const uint64_t* states,
                        const uint32_t* n_perms, const uint32_t* init_obs) {
  if (!c || !states || !n_perms || !init_obs || n == 0 || max_perms == 0) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  if (!ctx->has_device()) return ctx->fail(ZKSP_ERR_NO_DEVICE, "load_batch: client has no GPU");
  if (log_h < 5 || log_h > 14) return ctx->fail(ZKSP_ERR_UNSUPPORTED, "load_batch: log_h must be in [5, 14]");
  for (size_t i = 0; i < n; ++i)
    if ((size_t)n_perms[i] > max_perms || (size_t)n_perms[i] * 24 > ((size_t)1 << log_h))
      return ctx->fail(ZKSP_ERR_INVALID_ARG, "load_batch: n_perms does not fit the trace height");
  ZKSP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = workspace_ensure(ctx, log_h, (int)n, (int)max_perms);
  if (rc) return rc;
  Workspace* ws = ctx->ws.get();
  if (!ctx->domain(log_h)) return ZKSP_ERR_HIP;
  // states arrive [n][max_perms][25]; the workspace may have a larger perm stride
  if ((size_t)ws->max_perms == max_perms) {
    ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(ws->states, states, n * max_perms * 200, hipMemcpyHostToDevice, ctx->stream));
  } else {
    ZKSP_HIP_CHECK(ctx, hipMemcpy2DAsync(ws->states, (size_t)ws->max_perms * 200, states, max_perms * 200,
                                         max_perms * 200, n, hipMemcpyHostToDevice, ctx->stream));
  }
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(ws->n_perms, n_perms, n * 4, hipMemcpyHostToDevice, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(ws->init_obs, init_obs, n * kInitObs * 4, hipMemcpyHostToDevice, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ws->n = (int)n;
  return ZKSP_OK;
}

int zksp_hip_prove_resident(zksp_client* c) {
  if (!c) return ZKSP_ERR_INVALID_ARG;
  if (!c->ctx.has_device()) return c->ctx.fail(ZKSP_ERR_NO_DEVICE, "prove_resident: client has no GPU");
  ZKSP_HIP_CHECK(&c->ctx, hipSetDevice(c->ctx.device));
  return prove_resident(&c->ctx);
}

int zksp_hip_fetch_bodies(zksp_client* c, uint32_t* out, size_t cap_words) {
  if (!c || !out) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  Workspace* ws = ctx->ws.get();
  if (!ws || ws->n == 0) return ctx->fail(ZKSP_ERR_INVALID_ARG, "fetch_bodies: no batch");
  size_t words = (size_t)ws->n * ws->body_words;
  if (cap_words < words) return ctx->fail(ZKSP_ERR_INVALID_ARG, "fetch_bodies: buffer too small");
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(out, ws->body, words * 4, hipMemcpyDeviceToHost, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ZKSP_OK;
}

int zksp_hip_fetch_roots(zksp_client* c, uint32_t* out, size_t cap_words) {
  if (!c || !out) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  Workspace* ws = ctx->ws.get();
  if (!ws || ws->n == 0) return ctx->fail(ZKSP_ERR_INVALID_ARG, "fetch_roots: no batch");
  if (cap_words < (size_t)ws->n * 8) return ctx->fail(ZKSP_ERR_INVALID_ARG, "fetch_roots: buffer too small");
  // the trace commitment is the first 8 words of every proof body
  ZKSP_HIP_CHECK(ctx, hipMemcpy2DAsync(out, 32, ws->body, ws->body_words * 4, 32, (size_t)ws->n, hipMemcpyDeviceToHost,
                                       ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ZKSP_OK;
}

#endif  // ZKSP_COMPONENT

int zksp_hip_sync(zksp_client* c) {
  if (!c || !c->ctx.has_device()) return ZKSP_ERR_INVALID_ARG;
  ZKSP_HIP_CHECK(&c->ctx, hipStreamSynchronize(c->ctx.stream));
  return ZKSP_OK;
}
int zksp_hip_timer_start(zksp_client* c) {
  if (!c || !c->ctx.has_device()) return ZKSP_ERR_INVALID_ARG;
  ZKSP_HIP_CHECK(&c->ctx, hipEventRecord(c->ctx.timer_a, c->ctx.stream));
  return ZKSP_OK;
}
int zksp_hip_timer_stop(zksp_client* c, float* ms) {
  if (!c || !ms || !c->ctx.has_device()) return ZKSP_ERR_INVALID_ARG;
  ZKSP_HIP_CHECK(&c->ctx, hipEventRecord(c->ctx.timer_b, c->ctx.stream));
  ZKSP_HIP_CHECK(&c->ctx, hipEventSynchronize(c->ctx.timer_b));
  ZKSP_HIP_CHECK(&c->ctx, hipEventElapsedTime(ms, c->ctx.timer_a, c->ctx.timer_b));
  return ZKSP_OK;
}
int zksp_hip_profile_enable(zksp_client* c, int on) {
  if (!c) return ZKSP_ERR_INVALID_ARG;
  c->ctx.profile = on != 0;
  return ZKSP_OK;
}
int zksp_hip_profile_reset(zksp_client* c) {
  if (!c) return ZKSP_ERR_INVALID_ARG;
  c->ctx.spans.clear();
  c->ctx.event_used = 0;
  return ZKSP_OK;
}
int zksp_hip_profile_read(zksp_client* c, const char* kernel, double* total_ms, uint64_t* launches) {
  if (!c || !kernel || !total_ms || !launches || !c->ctx.has_device()) return ZKSP_ERR_INVALID_ARG;
  Context* ctx = &c->ctx;
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0;
  uint64_t cnt = 0;
  for (auto& sp : ctx->spans)
    if (sp.name == kernel) {
      float ms = 0;
      ZKSP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, sp.a, sp.b));
      tot += ms;
      ++cnt;
    }
  *total_ms = tot;
  *launches = cnt;
  return ZKSP_OK;
}

// ---------------------------------------------------------------------------
// prove / verify
// ---------------------------------------------------------------------------
int zksp_proof_public_values(const zksp_proof* p, const uint8_t** ptr, size_t* len) {
  if (!p || !ptr || !len) return ZKSP_ERR_INVALID_ARG;
  if (p->version == mach::kMachineVersion) {
    *ptr = p->bytes.data() + p->mhdr.pv_offset;
    *len = p->mhdr.pv_len;
    return ZKSP_OK;
  }
#ifdef ZKSP_COMPONENT
  *ptr = p->bytes.data() + p->hdr.pv_offset;
  *len = p->hdr.pv_len;
  return ZKSP_OK;
#else
  return ZKSP_ERR_PROOF_FORMAT;
#endif
}
int zksp_proof_serialize(const zksp_proof* p, const uint8_t** ptr, size_t* len) {
  if (!p || !ptr || !len) return ZKSP_ERR_INVALID_ARG;
  *ptr = p->bytes.data();
  *len = p->bytes.size();
  return ZKSP_OK;
}
int zksp_proof_deserialize(const uint8_t* buf, size_t len, zksp_proof** out) {
  if (!buf || !out) return ZKSP_ERR_INVALID_ARG;
  zksp_proof* p = new (std::nothrow) zksp_proof();
  if (!p) return ZKSP_ERR_INVALID_ARG;
  p->bytes.assign(buf, buf + len);
  std::string err;
  const bool machine = len >= 8 && reinterpret_cast<const uint32_t*>(p->bytes.data())[1] == mach::kMachineVersion;
#ifdef ZKSP_COMPONENT
  p->version = machine ? mach::kMachineVersion : kProofVersion;
  if (!(machine ? parse_machine_header(p->bytes.data(), p->bytes.size(), &p->mhdr, &err)
                : parse_proof_header(p->bytes.data(), p->bytes.size(), &p->hdr, &err))) {
    delete p;
    return ZKSP_ERR_PROOF_FORMAT;
  }
#else
  p->version = mach::kMachineVersion;
  if (!machine || !parse_machine_header(p->bytes.data(), p->bytes.size(), &p->mhdr, &err)) {
    delete p;
    return ZKSP_ERR_PROOF_FORMAT;
  }
#endif
  *out = p;
  return ZKSP_OK;
}
void zksp_proof_free(zksp_proof* p) { delete p; }

#ifdef ZKSP_COMPONENT
int zksp_proof_from_body(const uint32_t* body, size_t body_words, uint32_t log_h, const uint64_t* states, uint32_t n_perms,
                         uint32_t exit_code, const uint8_t* public_values, size_t pv_len, const uint32_t* pv_digest,
                         const uint32_t* deferred_digest, const uint32_t* vk_digest, zksp_proof** out) {
  if (!body || (!states && n_perms) || (!public_values && pv_len) || !pv_digest || !deferred_digest || !vk_digest || !out)
    return ZKSP_ERR_INVALID_ARG;
  if (pv_len > (1u << 24)) return ZKSP_ERR_INVALID_ARG;
  zksp_proof* p = new (std::nothrow) zksp_proof();
  if (!p) return ZKSP_ERR_INVALID_ARG;
  try {
    const size_t hwords = proof_header_words((uint32_t)pv_len, n_perms);
    p->bytes.assign((hwords + body_words) * 4, 0);
    uint32_t* w = reinterpret_cast<uint32_t*>(p->bytes.data());
    w[0] = kProofMagic; w[1] = kProofVersion; w[2] = log_h; w[3] = n_perms; w[4] = exit_code; w[5] = (uint32_t)pv_len;
    memcpy(w + 6, pv_digest, 32);
    memcpy(w + 14, deferred_digest, 32);
    memcpy(w + 22, vk_digest, 32);
    if (pv_len) memcpy(p->bytes.data() + 120, public_values, pv_len);
    uint8_t* io = p->bytes.data() + (30 + (pv_len + 3) / 4) * 4;
    for (uint32_t k = 0; k < n_perms; ++k) {  // the public I/O list: input state, keccak-f of it
      uint64_t st[25];
      memcpy(st, states + 25 * (size_t)k, 200);
      memcpy(io + (size_t)k * 400, st, 200);
      keccak_f1600(st);
      memcpy(io + (size_t)k * 400 + 200, st, 200);
    }
    memcpy(p->bytes.data() + hwords * 4, body, body_words * 4);
  } catch (...) {
    delete p;
    return ZKSP_ERR_INVALID_ARG;
  }
  p->version = kProofVersion;
  std::string err;
  if (!parse_proof_header(p->bytes.data(), p->bytes.size(), &p->hdr, &err)) {
    delete p;
    return ZKSP_ERR_PROOF_FORMAT;
  }
  *out = p;
  return ZKSP_OK;
}

#endif  // ZKSP_COMPONENT

int zksp_verify(zksp_client* c, const zksp_proof* p, const zksp_vk* vk) {
  if (!c || !p || !vk) return ZKSP_ERR_INVALID_ARG;
  std::string err;
  int rc;
  try {
    // The client's proof_mode decides which statement is being checked, never the proof's own version word: a
    // default (MACHINE) client must not accept a keccak-chip component proof, which says nothing about the guest's
    // execution (reference: `client.verify(&proof, &vk)`, prover/src/bin/main.rs:80, accepts proofs of execution only).
    const bool want_machine = c->ctx.params.proof_mode == ZKSP_PROOF_MACHINE;
    if (want_machine != (p->version == mach::kMachineVersion))
      return c->ctx.fail(ZKSP_ERR_VERIFY, want_machine
                                              ? "verify: not a machine proof (this client verifies proofs of execution only)"
                                              : "verify: not a keccak-chip component proof (client created with ZKSP_PROOF_KECCAK_CHIP)");
#ifdef ZKSP_COMPONENT
    rc = want_machine ? verify_machine_proof(p->bytes.data(), p->bytes.size(), vk->machine, c->ctx.params.num_queries,
                                             c->ctx.params.pow_bits, &err)
                      : verify_proof(p->bytes.data(), p->bytes.size(), vk->digest, c->ctx.params.num_queries,
                                     c->ctx.params.pow_bits, &err);
#else
    rc = verify_machine_proof(p->bytes.data(), p->bytes.size(), vk->machine, c->ctx.params.num_queries, c->ctx.params.pow_bits, &err);
#endif
  } catch (...) {
    return c->ctx.fail(ZKSP_ERR_VERIFY, "verify: out of memory");
  }
  if (rc) return c->ctx.fail(rc, "verify: " + err);
  return ZKSP_OK;
}

// ---------------------------------------------------------------------------
// kernel-level entry points
// ---------------------------------------------------------------------------
#define NEED_GPU(c)                                                                  \
  if (!(c)) return ZKSP_ERR_INVALID_ARG;                                             \
  if (!(c)->ctx.has_device()) return (c)->ctx.fail(ZKSP_ERR_NO_DEVICE, "no GPU bound to this client"); \
  ZKSP_HIP_CHECK(&(c)->ctx, hipSetDevice((c)->ctx.device))

int zksp_dev_malloc(zksp_client* c, size_t bytes, void** out) {
  NEED_GPU(c);
  if (!out) return ZKSP_ERR_INVALID_ARG;
  ZKSP_HIP_CHECK(&c->ctx, hipMalloc(out, bytes));
  return ZKSP_OK;
}
int zksp_dev_free(zksp_client* c, void* p) {
  NEED_GPU(c);
  ZKSP_HIP_CHECK(&c->ctx, hipStreamSynchronize(c->ctx.stream));
  ZKSP_HIP_CHECK(&c->ctx, hipFree(p));
  return ZKSP_OK;
}
int zksp_dev_upload(zksp_client* c, void* dst, const void* src, size_t bytes) {
  NEED_GPU(c);
  ZKSP_HIP_CHECK(&c->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->ctx.stream));
  ZKSP_HIP_CHECK(&c->ctx, hipStreamSynchronize(c->ctx.stream));
  return ZKSP_OK;
}
int zksp_dev_download(zksp_client* c, void* dst, const void* src, size_t bytes) {
  NEED_GPU(c);
  ZKSP_HIP_CHECK(&c->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->ctx.stream));
  ZKSP_HIP_CHECK(&c->ctx, hipStreamSynchronize(c->ctx.stream));
  return ZKSP_OK;
}
int zksp_dev_memset(zksp_client* c, void* dst, int value, size_t bytes) {
  NEED_GPU(c);
  ZKSP_HIP_CHECK(&c->ctx, hipMemsetAsync(dst, value, bytes, c->ctx.stream));
  return ZKSP_OK;
}

static inline uint32_t bitrev32(uint32_t v, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

int zksp_hip_lde(zksp_client* c, const uint32_t* d_in, int log_h, size_t ncols, uint32_t in_shift, uint32_t* d_coefs_br,
                 uint32_t* d_lde) {
  NEED_GPU(c);
  Context* ctx = &c->ctx;
  if (!d_in || !d_lde || in_shift == 0 || in_shift >= kP) return ZKSP_ERR_INVALID_ARG;
  const DeviceDomain* dom = ctx->domain(log_h);
  if (!dom) return ZKSP_ERR_UNSUPPORTED;
  const size_t h = (size_t)1 << log_h;
  const uint32_t* table = nullptr;
  uint32_t* tmp = nullptr;
  const Fp g = Fp::from_canonical(kGen);
  if (in_shift == 1) table = dom->in_scale_br;
  else if (in_shift == g.to_canonical()) table = dom->in_scale_br + h;
  else if (in_shift == (g * fp_root_of_unity(log_h + 1)).to_canonical()) table = dom->in_scale_br + 2 * h;
  else {
    std::vector<uint32_t> nat(h), br(h);
    const Fp si = Fp::from_canonical(in_shift).inv();
    Fp p = Fp::from_canonical((uint32_t)(h % kP)).inv();
    for (size_t k = 0; k < h; ++k) { nat[k] = p.v; p = p * si; }
    for (size_t pos = 0; pos < h; ++pos) br[pos] = nat[bitrev32((uint32_t)pos, log_h)];
    ZKSP_HIP_CHECK(ctx, hipMalloc(&tmp, h * 4));
    ZKSP_HIP_CHECK(ctx, hipMemcpy(tmp, br.data(), h * 4, hipMemcpyHostToDevice));
    table = tmp;
  }
  uint32_t* tmp_coefs = nullptr;
  if (log_h > 14 && !d_coefs_br) {  // the two-pass path stages coefficients in HBM
    ZKSP_HIP_CHECK(ctx, hipMalloc(&tmp_coefs, ncols * h * 4));
    d_coefs_br = tmp_coefs;
  }
  launch_lde(ctx->stream, d_in, d_coefs_br, d_lde, dom->twc_fwd, dom->twc_inv, table, 0, 0, dom->out_scale_br, log_h, ncols);
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (tmp) ZKSP_HIP_CHECK(ctx, hipFree(tmp));
  if (tmp_coefs) ZKSP_HIP_CHECK(ctx, hipFree(tmp_coefs));
  ZKSP_HIP_CHECK(ctx, hipGetLastError());
  return ZKSP_OK;
}

int zksp_hip_merkle_commit(zksp_client* c, const uint32_t* d_mat, int width, int log_n, uint32_t* d_tree) {
  NEED_GPU(c);
  if (!d_mat || !d_tree || width < 0 || log_n < 0 || log_n > 28) return ZKSP_ERR_INVALID_ARG;
  const size_t n = (size_t)1 << log_n;
  launch_merkle_commit(c->ctx.stream, d_mat, (size_t)width * n, width, log_n, d_tree, (2 * n - 1) * 8, 1, c->ctx.d_consts);
  ZKSP_HIP_CHECK(&c->ctx, hipStreamSynchronize(c->ctx.stream));
  ZKSP_HIP_CHECK(&c->ctx, hipGetLastError());
  return ZKSP_OK;
}

int zksp_hip_poseidon2_permute(zksp_client* c, uint32_t* d_states, size_t n) {
  NEED_GPU(c);
  if (!d_states) return ZKSP_ERR_INVALID_ARG;
  launch_poseidon2_permute(c->ctx.stream, d_states, n, c->ctx.d_consts);
  ZKSP_HIP_CHECK(&c->ctx, hipStreamSynchronize(c->ctx.stream));
  ZKSP_HIP_CHECK(&c->ctx, hipGetLastError());
  return ZKSP_OK;
}

int zksp_host_poseidon2_permute(uint32_t* states, size_t n, int impl) {
  if (!states || impl < 0 || impl > 4) return ZKSP_ERR_INVALID_ARG;
  if ((impl == 1 || impl == 2) && !p2avx2::usable()) return ZKSP_ERR_UNSUPPORTED;
  if (impl >= 3 && !p2avx512::usable()) return ZKSP_ERR_UNSUPPORTED;
  for (size_t i = 0; i < 16 * n; ++i)
    if (states[i] >= kP) return ZKSP_ERR_INVALID_ARG;
  const P2Consts* k = &host_p2_consts();
  const size_t group = impl == 4 ? 4 : impl == 2 ? 2 : 1;  // states in lockstep (the remainder one by one)
  for (size_t i = 0; i < n;) {
    const size_t cnt = i + group <= n ? group : 1;
    Fp st[4][16];
    for (size_t g = 0; g < cnt; ++g)
      for (int j = 0; j < 16; ++j) st[g][j] = Fp::from_canonical(states[16 * (i + g) + j]);
    if (cnt == 4) p2avx512::permute4(&st[0][0].v, &st[1][0].v, &st[2][0].v, &st[3][0].v, k->ext, k->internal, k->diag);
    else if (cnt == 2) p2avx2::permute2(&st[0][0].v, &st[1][0].v, k->ext, k->internal, k->diag);
    else if (impl >= 3) p2avx512::permute1(&st[0][0].v, k->ext, k->internal, k->diag);
    else if (impl >= 1) p2avx2::permute(&st[0][0].v, k->ext, k->internal, k->diag);
    else p2_permute(st[0], k);
    for (size_t g = 0; g < cnt; ++g)
      for (int j = 0; j < 16; ++j) states[16 * (i + g) + j] = st[g][j].to_canonical();
    i += cnt;
  }
  return ZKSP_OK;
}

#ifdef ZKSP_COMPONENT  // parity entry points of the component path's keccak-only kernels
int zksp_hip_keccak_trace(zksp_client* c, const uint64_t* d_states, uint32_t n_perms, int log_h, uint32_t* d_trace) {
  NEED_GPU(c);
  Context* ctx = &c->ctx;
  if (!d_states || !d_trace || log_h < 1 || log_h > 26 || (size_t)n_perms * 24 > ((size_t)1 << log_h)) return ZKSP_ERR_INVALID_ARG;
  uint32_t* d_np = nullptr;
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d_np, 4));
  ZKSP_HIP_CHECK(ctx, hipMemcpy(d_np, &n_perms, 4, hipMemcpyHostToDevice));
  launch_keccak_trace(ctx->stream, d_states, (int)std::max<uint32_t>(n_perms, 1), d_np, d_trace, log_h, 1);
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipFree(d_np));
  ZKSP_HIP_CHECK(ctx, hipGetLastError());
  return ZKSP_OK;
}

int zksp_hip_keccak_quotient(zksp_client* c, const uint32_t* d_lde, const uint32_t* d_lde_p, int log_h,
                             const uint32_t* challenges, uint32_t* d_quot) {
  NEED_GPU(c);
  Context* ctx = &c->ctx;
  if (!d_lde || !d_lde_p || !challenges || !d_quot) return ZKSP_ERR_INVALID_ARG;
  if (log_h > 14) return ctx->fail(ZKSP_ERR_UNSUPPORTED, "keccak_quotient: log_h must be <= 14");
  const DeviceDomain* dom = ctx->domain(log_h);
  if (!dom) return ZKSP_ERR_UNSUPPORTED;
  const size_t n = (size_t)2 << log_h;
  // challenges: alpha, gamma, beta, cumulative sum (4 canonical words each)
  uint32_t cm[16];
  for (int i = 0; i < 16; ++i) {
    if (challenges[i] >= kP) return ZKSP_ERR_INVALID_ARG;
    cm[i] = Fp::from_canonical(challenges[i]).v;
  }
  uint32_t *d_ch = nullptr, *d_pows = nullptr, *d_bpows = nullptr, *d_partial = nullptr;
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d_ch, 64));
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d_pows, (size_t)kNumAllConstraints * 16));
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d_bpows, (size_t)kBusTuple * 16));
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d_partial, (size_t)13 * n * 16));
  ZKSP_HIP_CHECK(ctx, hipMemcpy(d_ch, cm, 64, hipMemcpyHostToDevice));
  launch_ext_powers(ctx->stream, d_ch, 4, kR1, d_pows, (size_t)kNumAllConstraints * 4, kNumAllConstraints, 0, 1);
  launch_ext_powers(ctx->stream, d_ch + 8, 4, kR1, d_bpows, (size_t)kBusTuple * 4, kBusTuple, 0, 1);
  QuotientArgs qa;
  qa.lde = d_lde;
  qa.lde_p = d_lde_p;
  qa.alpha_pows = d_pows;
  qa.bus_ch = d_ch + 4;  // gamma, beta
  qa.beta_pows = d_bpows;
  qa.cum_sum = d_ch + 12;
  qa.sel_first = dom->sel_first;
  qa.sel_trans = dom->sel_trans;
  qa.sel_last = dom->sel_last;
  qa.zh_inv = dom->zh_inv;
  qa.partial = d_partial;
  qa.quot = d_quot;
  qa.logh = log_h;
  qa.batch = 1;
  launch_keccak_quotient(ctx->stream, qa);
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipFree(d_ch));
  ZKSP_HIP_CHECK(ctx, hipFree(d_pows));
  ZKSP_HIP_CHECK(ctx, hipFree(d_bpows));
  ZKSP_HIP_CHECK(ctx, hipFree(d_partial));
  ZKSP_HIP_CHECK(ctx, hipGetLastError());
  return ZKSP_OK;
}

// trace [2633][H], gamma/beta (4 canonical words each) -> phi [4][H], cumulative sum (4 Montgomery words on device)
int zksp_hip_bus_perm_trace(zksp_client* c, const uint32_t* d_trace, int log_h, const uint32_t* gamma_beta, uint32_t* d_phi,
                            uint32_t* d_cum_sum) {
  NEED_GPU(c);
  Context* ctx = &c->ctx;
  if (!d_trace || !gamma_beta || !d_phi || !d_cum_sum || log_h < 1 || log_h > 20) return ZKSP_ERR_INVALID_ARG;
  uint32_t cm[8];
  for (int i = 0; i < 8; ++i) {
    if (gamma_beta[i] >= kP) return ZKSP_ERR_INVALID_ARG;
    cm[i] = Fp::from_canonical(gamma_beta[i]).v;
  }
  uint32_t *d_ch = nullptr, *d_bpows = nullptr, *d_terms = nullptr;
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d_ch, 32));
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d_bpows, (size_t)kBusTuple * 16));
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d_terms, ((size_t)16) << log_h));
  ZKSP_HIP_CHECK(ctx, hipMemcpy(d_ch, cm, 32, hipMemcpyHostToDevice));
  launch_ext_powers(ctx->stream, d_ch + 4, 4, kR1, d_bpows, (size_t)kBusTuple * 4, kBusTuple, 0, 1);
  launch_bus_perm_trace(ctx->stream, d_trace, d_ch, d_bpows, d_terms, d_phi, d_cum_sum, log_h, 1);
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipFree(d_ch));
  ZKSP_HIP_CHECK(ctx, hipFree(d_bpows));
  ZKSP_HIP_CHECK(ctx, hipFree(d_terms));
  ZKSP_HIP_CHECK(ctx, hipGetLastError());
  return ZKSP_OK;
}

#endif  // ZKSP_COMPONENT

int zksp_hip_fri_fold(zksp_client* c, const uint32_t* d_in, int log_hk, uint32_t shift_k, const uint32_t* beta,
                      uint32_t* d_out) {
  NEED_GPU(c);
  Context* ctx = &c->ctx;
  if (!d_in || !beta || !d_out || shift_k == 0 || shift_k >= kP) return ZKSP_ERR_INVALID_ARG;
  const DeviceDomain* dom = ctx->domain(log_hk);
  if (!dom) return ZKSP_ERR_UNSUPPORTED;
  uint32_t bm[4];
  for (int i = 0; i < 4; ++i) {
    if (beta[i] >= kP) return ZKSP_ERR_INVALID_ARG;
    bm[i] = Fp::from_canonical(beta[i]).v;
  }
  uint32_t* d_beta = nullptr;
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d_beta, 16));
  ZKSP_HIP_CHECK(ctx, hipMemcpy(d_beta, bm, 16, hipMemcpyHostToDevice));
  const Fp s = Fp::from_canonical(shift_k);
  const uint32_t x0 = s.inv().v, x1 = (s * fp_root_of_unity(log_hk + 1)).inv().v;
  launch_fri_fold(ctx->stream, d_in, 0, d_out, 0, d_beta, 0, dom->tw_inv, 0, x0, x1, log_hk, 1);
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipFree(d_beta));
  ZKSP_HIP_CHECK(ctx, hipGetLastError());
  return ZKSP_OK;
}

int zksp_hip_microbench(zksp_client* c, int which, double* gops) {
  NEED_GPU(c);
  Context* ctx = &c->ctx;
  if (!gops || which < 0 || (which > 5 + 64 && which != 100)) return ZKSP_ERR_INVALID_ARG;
  uint32_t* d = nullptr;
  ZKSP_HIP_CHECK(ctx, hipMalloc(&d, 64));
  if (which == 100) which = -6;  // the 64-bit shift-add chain: rate_kernel<6>
  if (which > 5) {
    // 6 + k: Poseidon2 permutations per second with k+1 workgroups of 256 per CU (result in G perms/s)
    const int per_cu = which - 5, blocks_p = 256 * per_cu, iters_p = 64;
    launch_perm_rate_kernel(ctx->stream, d, blocks_p, 2, ctx->d_consts);
    ZKSP_HIP_CHECK(ctx, hipEventRecord(ctx->timer_a, ctx->stream));
    launch_perm_rate_kernel(ctx->stream, d, blocks_p, iters_p, ctx->d_consts);
    ZKSP_HIP_CHECK(ctx, hipEventRecord(ctx->timer_b, ctx->stream));
    ZKSP_HIP_CHECK(ctx, hipEventSynchronize(ctx->timer_b));
    float msp = 0;
    ZKSP_HIP_CHECK(ctx, hipEventElapsedTime(&msp, ctx->timer_a, ctx->timer_b));
    ZKSP_HIP_CHECK(ctx, hipFree(d));
    *gops = (double)blocks_p * 256.0 * iters_p / (msp * 1e-3) / 1e9;
    return ZKSP_OK;
  }
  const int blocks = 256 * 16, iters = 2000;
  if (which < 0) which = -which;
  launch_rate_kernel(ctx->stream, which, d, blocks, 10);  // warm-up
  ZKSP_HIP_CHECK(ctx, hipEventRecord(ctx->timer_a, ctx->stream));
  launch_rate_kernel(ctx->stream, which, d, blocks, iters);
  ZKSP_HIP_CHECK(ctx, hipEventRecord(ctx->timer_b, ctx->stream));
  ZKSP_HIP_CHECK(ctx, hipEventSynchronize(ctx->timer_b));
  float ms = 0;
  ZKSP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->timer_a, ctx->timer_b));
  ZKSP_HIP_CHECK(ctx, hipFree(d));
  // 4 independent chains x 16 unrolled steps per iteration, one op per step per chain
  const double ops = (double)blocks * 256.0 * iters * 16.0 * 4.0;
  *gops = ops / (ms * 1e-3) / 1e9;
  return ZKSP_OK;
}

}  // extern "C"
