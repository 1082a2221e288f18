// C ABI of the machine-proof side (include/zksp.h, "machine proof" section): traced execution
// records for tests and the CPU oracle.  Nothing here throws across the boundary.
#include "../../../include/zksp.h"

#include <cstring>
#include <new>

#include "api_types.hpp"

using namespace zksp;

extern "C" {

int zksp_machine_trace(zksp_client* c, const zksp_pk* pk, const zksp_stdin* stdin_, zksp_mtrace** out) {
  if (!c || !pk || !stdin_ || !out) return ZKSP_ERR_INVALID_ARG;
  *out = nullptr;
  zksp_mtrace* t = new (std::nothrow) zksp_mtrace();
  if (!t) return ZKSP_ERR_INVALID_ARG;
  t->prog = &pk->mprog;
  try {
    trace_execute(pk->elf, pk->mprog, stdin_->entries, (uint64_t)1 << 21, &t->t);
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
    case ZKSP_MT_IMAGE_USED: *ptr = m.image_used.data(); *bytes = m.image_used.size() * 4; break;
    case ZKSP_MT_PROGRAM: *ptr = t->prog->rows.data(); *bytes = t->prog->rows.size() * sizeof(ProgramRow); break;
    case ZKSP_MT_IMAGE: *ptr = t->prog->image.data(); *bytes = t->prog->image.size() * sizeof(ImageRow); break;
    case ZKSP_MT_PUBLIC_VALUES: *ptr = m.rec.public_values.data(); *bytes = m.rec.public_values.size(); break;
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
  return ZKSP_OK;
}

int zksp_vk_machine(const zksp_vk* vk, uint32_t* prep_root8, uint32_t* digest8) {
  if (!vk || !prep_root8 || !digest8) return ZKSP_ERR_INVALID_ARG;
  memcpy(prep_root8, vk->machine.prep_root, 32);
  memcpy(digest8, vk->machine.digest, 32);
  return ZKSP_OK;
}

}  // extern "C"
