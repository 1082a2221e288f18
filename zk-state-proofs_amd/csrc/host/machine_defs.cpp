// See machine_defs.hpp.  Bus protocol (DESIGN.md "Machine proof"):
//   MEM    (addr, lo, hi, time)   every access consumes its predecessor's tuple and produces its own
//   PROG   (pc, op, wr, use2, rd, rs1, rs2, imm_lo, imm_hi, tgt)   instruction fetch
//   KCALL  (time, ptr_lo, ptr_hi)   CPU -> keccak-memory: one precompile call
//   KIO    (time, word index, in_lo, in_hi, out_lo, out_hi)   keccak-f chip -> keccak-memory
//   MUL    (hi, a_lo, a_hi, b_lo, b_hi, c_lo, c_hi)   CPU -> multiplier
//   PUBC   (kind, index, lo, hi), PUBH (exit_lo, exit_hi)   CPU -> verifier
//   RANGE  (value)   a 12-bit limb of an access-time difference, looked up in the range table
#include "machine_defs.hpp"

#include <cstdlib>
#include <mutex>

namespace zksp {
namespace mach {

namespace {

uint32_t mont(uint64_t canonical) { return Fp::from_canonical((uint32_t)(canonical % kP)).v; }

LinForm lf_zero() {
  LinForm f{};
  return f;
}
void lf_add(LinForm& f, int col, uint64_t coef) {
  if (f.n >= kLfMax) abort();
  f.col[f.n] = col;
  f.coef[f.n] = mont(coef);
  f.n++;
}
LinForm lf_col(int col) {
  LinForm f = lf_zero();
  lf_add(f, col, 1);
  return f;
}
LinForm lf_const(uint32_t c) {
  LinForm f = lf_zero();
  f.c0 = mont(c);
  return f;
}
LinForm lf_limb(int bits, int limb) {
  LinForm f = lf_zero();
  for (int i = 0; i < 16; ++i) lf_add(f, bits + 16 * limb + i, (uint64_t)1 << i);
  return f;
}
LinForm lf_plus(LinForm f, uint32_t c) {
  f.c0 = (Fp::raw(f.c0) + Fp::raw(mont(c))).v;
  return f;
}
Interaction mem_inter(int sign, const LinForm& mult, const LinForm& addr, const LinForm& lo, const LinForm& hi, const LinForm& ts) {
  Interaction it{};
  it.bus = BUS_MEM; it.sign = sign; it.mult = mult; it.n_el = 4;
  it.el[0] = addr; it.el[1] = lo; it.el[2] = hi; it.el[3] = ts;
  return it;
}

Interaction range_inter(int sign, const LinForm& mult, const LinForm& value) {
  Interaction it{};
  it.bus = BUS_RANGE; it.sign = sign; it.mult = mult; it.n_el = 1;
  it.el[0] = value;
  return it;
}

Interaction g_cpu[21], g_keccak[50], g_kmem[6], g_memfinal[2], g_image[1], g_program[1], g_mul[2], g_range[1];
ChipDef g_chips[kNumChips];

void build() {
  const LinForm is_real = lf_col(C_IS_REAL), ts = lf_col(C_TS);
  const LinForm a_lo = lf_col(C_A), a_hi = lf_col(C_A + 1), b_lo = lf_limb(C_B, 0), b_hi = lf_limb(C_B, 1),
                c_lo = lf_limb(C_C, 0), c_hi = lf_limb(C_C, 1), m_lo = lf_limb(C_M, 0), m_hi = lf_limb(C_M, 1);
  {
    Interaction& it = g_cpu[0];
    it = Interaction{};
    it.bus = BUS_PROG; it.sign = -1; it.mult = is_real; it.n_el = 10;
    it.el[0] = lf_col(C_PC);
    it.el[1] = lf_zero();
    for (int k = 1; k <= kNumOps; ++k) lf_add(it.el[1], C_OP + k - 1, (uint64_t)k);
    it.el[2] = lf_col(C_WR); it.el[3] = lf_col(C_USE2); it.el[4] = lf_col(C_RD); it.el[5] = lf_col(C_RS1);
    it.el[6] = lf_col(C_RS2); it.el[7] = lf_col(C_IMM_LO); it.el[8] = lf_col(C_IMM_HI); it.el[9] = lf_col(C_TGT);
  }
  g_cpu[1] = mem_inter(-1, is_real, lf_col(C_RS1), b_lo, b_hi, lf_col(C_R1_PTS));
  g_cpu[2] = mem_inter(+1, is_real, lf_col(C_RS1), b_lo, b_hi, ts);
  g_cpu[3] = mem_inter(-1, lf_col(C_USE2), lf_col(C_RS2), c_lo, c_hi, lf_col(C_R2_PTS));
  g_cpu[4] = mem_inter(+1, lf_col(C_USE2), lf_col(C_RS2), c_lo, c_hi, lf_plus(ts, 1));
  {
    LinForm memq = lf_zero(), maddr = lf_zero();
    for (int k = LB; k <= SW; ++k) lf_add(memq, C_OP + k - 1, 1);
    lf_add(memq, C_OP + ECALL - 1, 1);
    // word address = X as a value (mod p: guest addresses stay below 0x78000000 < p) - byte offset
    for (int i = 0; i < 32; ++i) lf_add(maddr, C_X + i, (uint64_t)1 << i);
    lf_add(maddr, C_O1, kP - 1); lf_add(maddr, C_O2, kP - 2); lf_add(maddr, C_O3, kP - 3);
    g_cpu[5] = mem_inter(-1, memq, maddr, m_lo, m_hi, lf_col(C_M_PTS));
    g_cpu[6] = mem_inter(+1, memq, maddr, lf_col(C_MV_LO), lf_col(C_MV_HI), lf_plus(ts, 2));
    // the limbs of the four access-time differences are looked up when their access is live
    for (int j = 0; j < kTsLimbs; ++j) {
      g_cpu[13 + j] = range_inter(-1, is_real, lf_col(C_R1_D + j));
      g_cpu[15 + j] = range_inter(-1, lf_col(C_USE2), lf_col(C_R2_D + j));
      g_cpu[17 + j] = range_inter(-1, memq, lf_col(C_M_D + j));
      g_cpu[19 + j] = range_inter(-1, lf_col(C_WR), lf_col(C_W_D + j));
    }
  }
  g_cpu[7] = mem_inter(-1, lf_col(C_WR), lf_col(C_RD), lf_col(C_W_PLO), lf_col(C_W_PHI), lf_col(C_W_PTS));
  g_cpu[8] = mem_inter(+1, lf_col(C_WR), lf_col(C_RD), a_lo, a_hi, lf_plus(ts, 3));
  {
    Interaction& kc = g_cpu[9];
    kc = Interaction{};
    kc.bus = BUS_KCALL; kc.sign = +1; kc.mult = lf_col(C_OP + KECCAK - 1); kc.n_el = 3;
    kc.el[0] = ts; kc.el[1] = c_lo; kc.el[2] = c_hi;
    Interaction& mu = g_cpu[10];
    mu = Interaction{};
    mu.bus = BUS_MUL; mu.sign = +1; mu.n_el = 7;
    mu.mult = lf_zero(); lf_add(mu.mult, C_OP + MUL - 1, 1); lf_add(mu.mult, C_OP + MULHU - 1, 1);
    mu.el[0] = lf_col(C_OP + MULHU - 1);
    mu.el[1] = a_lo; mu.el[2] = a_hi; mu.el[3] = b_lo; mu.el[4] = b_hi; mu.el[5] = c_lo; mu.el[6] = c_hi;
    Interaction& pc = g_cpu[11];
    pc = Interaction{};
    pc.bus = BUS_PUBC; pc.sign = +1; pc.n_el = 4;
    pc.mult = lf_zero(); lf_add(pc.mult, C_SC + SC_COMMIT, 1); lf_add(pc.mult, C_SC + SC_DEFER, 1);
    pc.el[0] = lf_zero(); lf_add(pc.el[0], C_SC + SC_COMMIT, 1); lf_add(pc.el[0], C_SC + SC_DEFER, 2);
    pc.el[1] = c_lo; pc.el[2] = m_lo; pc.el[3] = m_hi;
    Interaction& ph = g_cpu[12];
    ph = Interaction{};
    ph.bus = BUS_PUBH; ph.sign = +1; ph.mult = lf_col(C_SC + SC_HALT); ph.n_el = 2;
    ph.el[0] = c_lo; ph.el[1] = c_hi;
  }
  for (int i = 0; i < 50; ++i) {
    Interaction& it = g_keccak[i];
    it = Interaction{};
    const int lane = i >> 1, half = i & 1;
    const int out = lane == 0 ? ka::kAppp00 + 2 * half : ka::kApp + 4 * lane + 2 * half;
    it.bus = BUS_KIO; it.sign = +1; it.mult = lf_col(ka::kExport); it.n_el = 6;
    it.el[0] = lf_col(KC_TS); it.el[1] = lf_const((uint32_t)i);
    it.el[2] = lf_col(ka::kPreimage + 4 * lane + 2 * half); it.el[3] = lf_col(ka::kPreimage + 4 * lane + 2 * half + 1);
    it.el[4] = lf_col(out); it.el[5] = lf_col(out + 1);
  }
  {
    Interaction& kc = g_kmem[0];
    kc = Interaction{};
    kc.bus = BUS_KCALL; kc.sign = -1; kc.mult = lf_col(KM_CALL); kc.n_el = 3;
    kc.el[0] = lf_col(KM_TS); kc.el[1] = lf_col(KM_PTR_LO); kc.el[2] = lf_col(KM_PTR_HI);
    Interaction& io = g_kmem[1];
    io = Interaction{};
    io.bus = BUS_KIO; io.sign = -1; io.mult = lf_col(KM_IS_REAL); io.n_el = 6;
    io.el[0] = lf_col(KM_TS); io.el[1] = lf_col(KM_IDX); io.el[2] = lf_col(KM_OLD_LO); io.el[3] = lf_col(KM_OLD_HI);
    io.el[4] = lf_col(KM_NEW_LO); io.el[5] = lf_col(KM_NEW_HI);
    g_kmem[2] = mem_inter(-1, lf_col(KM_IS_REAL), lf_col(KM_ADDR), lf_col(KM_OLD_LO), lf_col(KM_OLD_HI), lf_col(KM_PTS));
    g_kmem[3] = mem_inter(+1, lf_col(KM_IS_REAL), lf_col(KM_ADDR), lf_col(KM_NEW_LO), lf_col(KM_NEW_HI), lf_plus(lf_col(KM_TS), 2));
    for (int j = 0; j < kTsLimbs; ++j) g_kmem[4 + j] = range_inter(-1, lf_col(KM_IS_REAL), lf_col(KM_D + j));
  }
  g_memfinal[0] = mem_inter(-1, lf_col(MF_IS_REAL), lf_col(MF_ADDR), lf_col(MF_FIN_LO), lf_col(MF_FIN_HI), lf_col(MF_FIN_TS));
  g_memfinal[1] = mem_inter(+1, lf_col(MF_IS_INIT), lf_col(MF_ADDR), lf_limb(MF_INIT, 0), lf_limb(MF_INIT, 1), lf_const(0));
  g_image[0] = mem_inter(+1, lf_col(kImagePrepWidth + 0), lf_col(IMG_P_ADDR), lf_col(IMG_P_LO), lf_col(IMG_P_HI), lf_const(0));
  {
    Interaction& it = g_program[0];
    it = Interaction{};
    it.bus = BUS_PROG; it.sign = +1; it.mult = lf_col(kProgramPrepWidth + 0); it.n_el = 10;
    for (int j = 0; j < 10; ++j) it.el[j] = lf_col(j);
  }
  for (int hi = 0; hi < 2; ++hi) {
    Interaction& it = g_mul[hi];
    it = Interaction{};
    it.bus = BUS_MUL; it.sign = -1; it.n_el = 7;
    if (hi) it.mult = lf_col(MU_HI);
    else { it.mult = lf_zero(); lf_add(it.mult, MU_IS_REAL, 1); lf_add(it.mult, MU_HI, kP - 1); }
    it.el[0] = lf_const((uint32_t)hi);
    it.el[1] = lf_limb(MU_P, 2 * hi); it.el[2] = lf_limb(MU_P, 2 * hi + 1);
    it.el[3] = lf_limb(MU_B, 0); it.el[4] = lf_limb(MU_B, 1); it.el[5] = lf_limb(MU_C, 0); it.el[6] = lf_limb(MU_C, 1);
  }
  g_range[0] = range_inter(+1, lf_col(kRangePrepWidth + 0), lf_col(0));
  g_chips[kRange] = {"range", kRangePrepWidth, kRangeWidth, 1, g_range, 0};
  g_chips[kCpu] = {"cpu", 0, kCpuWidth, 21, g_cpu, kCpuConstraints};
  g_chips[kCpu2] = {"cpu2", 0, kCpuWidth, 21, g_cpu, kCpuConstraints};
  g_chips[kKeccak] = {"keccak", 0, kKeccakWidth, 50, g_keccak, kKeccakConstraints};
  g_chips[kKmem] = {"keccak-mem", 0, kKmemWidth, 6, g_kmem, kKmemConstraints};
  g_chips[kMemFinal] = {"mem-final", 0, kMemFinalWidth, 2, g_memfinal, kMemFinalConstraints};
  g_chips[kImage] = {"image", kImagePrepWidth, kImageWidth, 1, g_image, 1};
  g_chips[kProgram] = {"program", kProgramPrepWidth, kProgramWidth, 1, g_program, 0};
  g_chips[kMul] = {"mul", 0, kMulWidth, 2, g_mul, kMulConstraints};
}

}  // namespace

const ChipDef& chip_def(int chip) {
  static std::once_flag once;
  std::call_once(once, build);
  return g_chips[chip];
}

}  // namespace mach
}  // namespace zksp
