// Device side of the machine proof (kernels_machine.h): what
// sp1-core-machine / sp1-stark do on the host for the CPU, memory, program and ALU chips beneath
// the reference's `client.prove(&pk, stdin).run()` (prover/src/bin/main.rs:71-74; Cargo.lock:7130,
// :7485), here as HIP kernels over a batch of proofs that share one program.
//
// Data layout: every matrix is column-major, so in trace expansion (lane = row), leaf hashing
// (lane = LDE row), quotient evaluation and reduced openings (lane = LDE point) a wave reads or
// writes 64 consecutive words of one column at every step.  Tree positions are bit-reversed
// relative to the LDE index (position >> d addresses the matching row of a 2^d times shorter
// matrix); the leaf kernel keeps lanes on consecutive LDE rows and scatters only its 32-byte
// digest.
#include "kernels_machine.h"
#include "poseidon2_coop.hpp"

namespace zksp {

using namespace mach;

constexpr int kMT = 256;

__device__ __forceinline__ uint32_t mont(uint32_t canonical) { return Fp::from_canonical(canonical).v; }
__device__ __forceinline__ Fp4 m_load_fp4(const uint32_t* p) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  Fp4 r;
  r.c[0] = Fp::raw(v.x); r.c[1] = Fp::raw(v.y); r.c[2] = Fp::raw(v.z); r.c[3] = Fp::raw(v.w);
  return r;
}
__device__ __forceinline__ void m_store_fp4(uint32_t* p, const Fp4& a) {
  *reinterpret_cast<uint4*>(p) = make_uint4(a.c[0].v, a.c[1].v, a.c[2].v, a.c[3].v);
}

// ===========================================================================================
// trace expansion: one lane per row, records in, Montgomery columns out
// ===========================================================================================
struct Col {
  uint32_t* t;
  size_t cs;
  __device__ __forceinline__ void put(int col, uint32_t monty) const { t[(size_t)col * cs] = monty; }
  __device__ __forceinline__ void val(int col, uint32_t canonical) const { t[(size_t)col * cs] = mont(canonical); }
  __device__ __forceinline__ void flag(int col, bool on) const { t[(size_t)col * cs] = on ? kR1 : 0u; }
  __device__ __forceinline__ void bits(int col, uint32_t v, int n) const {
    for (int i = 0; i < n; ++i) t[(size_t)(col + i) * cs] = ((v >> i) & 1u) ? kR1 : 0u;
  }
  __device__ __forceinline__ void zero(int col, int n) const {
    for (int i = 0; i < n; ++i) t[(size_t)(col + i) * cs] = 0u;
  }
  // a 32-bit word as its two 16-bit limbs
  __device__ __forceinline__ void limbs(int col, uint32_t v) const {
    val(col, v & 0xffff);
    val(col + 1, v >> 16);
  }
};

__device__ __forceinline__ uint32_t less_than(uint32_t code, uint32_t b, uint32_t c) {
  return code == SLT ? (uint32_t)((int32_t)b < (int32_t)c) : (uint32_t)(b < c);
}

// One CPU instance: row r is cycle row0 + r (row0 = 0 for the first instance, its height for the second); rows past
// the last cycle execute the padding instruction (jal x0, 0 at the padding pc: reads x0, jumps to itself).
__global__ __launch_bounds__(kMT) void cpu_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh, uint32_t row0) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kCpuWidth * h + r, h};
  const size_t cyc = (size_t)row0 + r;
  const uint32_t ts = 4 * ((uint32_t)cyc + 1);
  const uint32_t n_cycles = rec.counts[kCountWords * b];
  const uint32_t pad_pc = rec.text_base + 4 * (rec.n_program - 1);
  uint32_t gap[3] = {0, 0, 0};
  if (cyc >= n_cycles) {
    const uint32_t pts = cyc == n_cycles ? rec.counts[kCountWords * b + 6] : ts - 4;
    o.zero(0, kCpuWidth);
    o.val(C_PC, pad_pc);
    o.val(C_TS, ts);
    o.val(C_NEXT_PC, pad_pc);
    o.put(selc(CL_JAL), kR1);
    o.limbs(C_TGT_LO, pad_pc);
    o.limbs(C_GAP, ts - pts - 1);
    return;
  }
  const uint32_t* cy = rec.cycles + ((size_t)b * rec.cap_cycles + cyc) * 12;
  const uint32_t pc = cy[0], bb = cy[2], wprev = cy[6];
  uint32_t a = cy[1], c = cy[3], m = cy[4], mv = cy[5];
  const uint32_t* p = rec.program + 9 * (size_t)((pc - rec.text_base) >> 2);
  const uint32_t op = p[1], wr = p[2], rd = p[4], rs1 = p[5], imm = p[7], tgt = p[8];
  uint32_t use2 = p[3], rs2 = p[6];
  const int cls = class_of(op);
  if (cls == CL_ECALL) { use2 = 0; rs2 = 0; c = 0; m = 0; mv = 0; }  // a0 and a1 are read by the ecall chip
  const uint32_t code = code_of(op);
  o.val(C_PC, pc);
  o.val(C_TS, ts);
  for (int k = 1; k <= kNumCls; ++k) o.flag(selc(k), k == cls);
  const bool uc = ucmp_of(op);
  o.val(C_CODE, code);
  o.flag(C_UC, uc);
  o.flag(C_WR, wr != 0);
  o.flag(C_USE2, use2 != 0);
  o.val(C_RD, rd); o.val(C_RS1, rs1); o.val(C_RS2, rs2);
  o.limbs(C_IMM_LO, imm);
  o.limbs(C_TGT_LO, tgt);
  uint32_t x = 0, next = pc + 4, k0 = 0, k1 = 0, off = 4;
  uint32_t x_lo_m = 0, x_hi_m = 0;  // beq / bne: the two limbs of X are inverses (Montgomery words)
  const uint32_t blo = bb & 0xffff, bhi = bb >> 16, clo = c & 0xffff, chi = c >> 16;
  switch (cls) {
    case CL_ADD: x = a; k0 = (blo + clo) >> 16; k1 = (bhi + chi + k0) >> 16; break;
    case CL_SUB: x = a; k0 = ((a & 0xffff) + clo) >> 16; k1 = ((a >> 16) + chi + k0) >> 16; break;
    case CL_JAL: next = tgt; break;
    case CL_JALR:
      x = bb + c; k0 = (blo + clo) >> 16; k1 = (bhi + chi + k0) >> 16;
      off = x & 1; next = x & ~1u;
      break;
    case CL_LW: case CL_LDS: case CL_SW: case CL_STS:  // the address is rs1 + immediate
      x = bb + imm; k0 = (blo + (imm & 0xffff)) >> 16; k1 = (bhi + (imm >> 16) + k0) >> 16;
      off = x & 3;
      break;
    case CL_BEQ: case CL_BNE:
      k0 = blo == clo; k1 = bhi == chi;
      x_lo_m = k0 ? 0u : (Fp::from_canonical(blo) - Fp::from_canonical(clo)).inv().v;
      x_hi_m = k1 ? 0u : (Fp::from_canonical(bhi) - Fp::from_canonical(chi)).inv().v;
      a = k0 & k1;
      if ((cls == CL_BEQ) == (a != 0)) next = tgt;
      break;
    case CL_BLT: case CL_BGE:
      a = less_than(code, bb, c);
      if ((cls == CL_BLT) == (a != 0)) next = tgt;
      break;
    case CL_ECALL:
      x = a;
      if (bb == 0x00) next = pad_pc;
      break;
    case CL_KECCAK: x = bb; next = bb; break;
    default: break;
  }
  if (uc) {  // the row's own unsigned comparison: X = B - C + 2^32 [B < C], limb by limb with borrows K0, K1
    k0 = blo < clo;
    k1 = bb < c;
    x = bb - c;
  }
  if (cls == CL_BEQ || cls == CL_BNE) { o.put(C_X, x_lo_m); o.put(C_X + 1, x_hi_m); }
  else o.limbs(C_X, x);
  // second access: rs2, or the word a load reads (which then sits in C); written location: rd, or the word a store
  // leaves behind (in A, its old value in W_P)
  const bool load = cls == CL_LW || cls == CL_LDS, store = cls == CL_SW || cls == CL_STS;
  o.limbs(C_A, store ? mv : a); o.limbs(C_B, bb); o.limbs(C_C, load ? m : c);
  o.flag(C_K0, k0 != 0); o.flag(C_K1, k1 != 0);
  for (uint32_t i = 1; i < 4; ++i) o.flag(C_O1 + i - 1, i == off);
  o.val(C_NEXT_PC, next);
  o.val(C_ADDR2, use2 ? rs2 : load ? (x & ~3u) : 0u);
  o.val(C_ADDR3, wr ? rd : store ? (x & ~3u) : 0u);
  gap[0] = ts - cy[7] - 1;
  if (use2) gap[1] = ts - cy[8];
  if (load) gap[1] = ts - cy[9];
  uint32_t wp = 0;
  if (wr) { gap[2] = ts + 1 - cy[10]; wp = wprev; }
  if (store) { gap[2] = ts + 1 - cy[9]; wp = m; }
  o.limbs(C_W_PLO, wp);
  for (int q = 0; q < 3; ++q) o.limbs(C_GAP + 2 * q, gap[q]);
}

// The ecall chip: row r is the r-th ecall of the run
__global__ __launch_bounds__(kMT) void ecall_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kEcallWidth * h + r, h};
  if (r >= rec.counts[kCountWords * b + 9]) { o.zero(0, kEcallWidth); return; }
  const uint32_t cyc = rec.ecall_idx[(size_t)b * rec.cap_ecall + r], ts = 4 * (cyc + 1);
  const uint32_t* cy = rec.cycles + ((size_t)b * rec.cap_cycles + cyc) * 12;
  const uint32_t pc = cy[0], a = cy[1], bb = cy[2], c = cy[3], m = cy[4];
  const uint32_t pad_pc = rec.text_base + 4 * (rec.n_program - 1);
  const uint32_t sc = bb == 0x00 ? 0 : bb == 0x02 ? 1 : bb == 0x10 ? 2 : bb == 0x1a ? 3 : bb == 0xf0 ? 4 : bb == 0xf1 ? 5 : 6;
  o.put(EC_IS_REAL, kR1);
  for (uint32_t i = 0; i < 6; ++i) o.flag(EC_SC + i, i == sc);
  o.val(EC_TS, ts);
  o.val(EC_PC, pc);
  o.val(EC_NP, bb == 0x00 ? pad_pc : pc + 4);
  o.val(EC_B_LO, bb & 0xffff);
  o.limbs(EC_A_LO, a);
  o.limbs(EC_C_LO, c);
  o.limbs(EC_M_LO, m);
  o.limbs(EC_GAP, ts - cy[8]);
  o.limbs(EC_GAP + 2, ts + 1 - cy[9]);
  // a HINT_READ of m bytes covers NW = ceil(m / 4) words: 4 NW = m + P1 + 2 P2
  const uint32_t nw = sc == 5 ? (m + 3) / 4 : 0u, padb = sc == 5 ? 4 * nw - m : 0u;
  o.val(EC_NW, nw);
  o.flag(EC_P1, (padb & 1u) != 0);
  o.flag(EC_P2, (padb & 2u) != 0);
}

// The hint chip: row r is the r-th word the run's HINT_READs cover, in the order of the reads: its address, how many words of its
// read are left (itself included), and - where the run touches it (the memory boundary list has the address) - its initial value
__global__ __launch_bounds__(kMT) void hint_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kHintWidth * h + r, h};
  if (r >= rec.counts[kCountWords * b + 13]) { o.zero(0, kHintWidth); return; }
  // which read: the HINT_READ ecalls in execution order (a run has one or two)
  const uint32_t n_ecall = rec.counts[kCountWords * b + 9];
  uint32_t before = 0, ptr = 0, nw = 0;
  for (uint32_t e = 0; e < n_ecall; ++e) {
    const uint32_t* cy = rec.cycles + ((size_t)b * rec.cap_cycles + rec.ecall_idx[(size_t)b * rec.cap_ecall + e]) * 12;
    if (cy[2] != 0xf1) continue;
    const uint32_t w = (cy[4] + 3) / 4;
    if (r < before + w) { ptr = cy[3]; nw = w; break; }
    before += w;
  }
  const uint32_t j = (uint32_t)r - before, addr = ptr + 4 * j;
  // is the address on the memory boundary list (sorted by address)?
  const uint32_t n = rec.counts[kCountWords * b + 2];
  const uint32_t* mf = rec.memfinal + (size_t)b * rec.cap_memfinal * 5;
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (mf[5 * (size_t)mid] < addr) lo = mid + 1;
    else hi = mid;
  }
  const bool used = lo < n && mf[5 * (size_t)lo] == addr;
  o.put(HN_IS_REAL, kR1);
  o.flag(HN_FIRST, j == 0);
  o.flag(HN_LAST, j + 1 == nw);
  o.val(HN_ADDR, addr);
  o.val(HN_CNT, nw - j);
  o.limbs(HN_LO, used ? mf[5 * (size_t)lo + 1] : 0u);
  o.flag(HN_USED, used);
}

// One ALU-chip instance: row r is event row0 + r of the list alu_idx (cycle indices)
__global__ __launch_bounds__(kMT) void alu_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh, uint32_t row0) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kAluWidth * h + r, h};
  const size_t ev = (size_t)row0 + r;
  if (ev >= rec.counts[kCountWords * b + 4]) { o.zero(0, kAluWidth); return; }
  const uint32_t cyc = rec.alu_idx[(size_t)b * rec.cap_alu + ev];
  const uint32_t* cy = rec.cycles + ((size_t)b * rec.cap_cycles + cyc) * 12;
  const uint32_t code = code_of(rec.program[9 * (size_t)((cy[0] - rec.text_base) >> 2) + 1]), bb = cy[2], c = cy[3];
  uint32_t a = cy[1], x = 0, k0 = 0, k1 = 0;
  if (code == SLL || code == SRL || code == SRA) x = 1u << (c & 31);
  if (code == SLT) {
    const uint32_t blo = bb & 0xffff, bhi = bb >> 16, clo = c & 0xffff, chi = c >> 16;
    k0 = blo < clo;
    k1 = less_than(code, bb, c);
    const uint32_t dlo = blo - clo + 65536 * k0;
    const int32_t dhi = (int32_t)bhi - (int32_t)chi - (int32_t)k0 + 65536 * (int32_t)k1 +
                        65536 * ((int32_t)(c >> 31) - (int32_t)(bb >> 31));
    x = dlo | ((uint32_t)dhi << 16);
    a = k1;
  }
  o.put(AL_IS_REAL, kR1);
  for (uint32_t k = 0; k < 4; ++k) o.flag(AL_SEL + k, SLL + k == code);
  o.limbs(AL_A, a);
  o.bits(AL_B, bb, 32); o.bits(AL_C, c, 32); o.bits(AL_X, x, 32);
  o.flag(AL_K0, k0 != 0); o.flag(AL_K1, k1 != 0);
}

// One bitwise-chip instance: row r is event row0 + r of the list bw_idx
__global__ __launch_bounds__(kMT) void bw_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh, uint32_t row0) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kBwWidth * h + r, h};
  const size_t ev = (size_t)row0 + r;
  if (ev >= rec.counts[kCountWords * b + 7]) { o.zero(0, kBwWidth); return; }
  const uint32_t cyc = rec.bw_idx[(size_t)b * rec.cap_bw + ev];
  const uint32_t* cy = rec.cycles + ((size_t)b * rec.cap_cycles + cyc) * 12;
  const uint32_t code = rec.program[9 * (size_t)((cy[0] - rec.text_base) >> 2) + 1], a = cy[1], bb = cy[2], c = cy[3];
  o.put(BW_IS_REAL, kR1);
  for (uint32_t k = 0; k < 3; ++k) o.flag(BW_SEL + k, XOR + k == code);
  for (int i = 0; i < 4; ++i) {
    o.val(BW_A + i, (a >> (8 * i)) & 0xff);
    o.val(BW_B + i, (bb >> (8 * i)) & 0xff);
    o.val(BW_C + i, (c >> (8 * i)) & 0xff);
  }
}

// Poseidon2 chip: row r is record r (air_machine.hpp: a heap node of an aggregation payload, or a sponge / path / injection
// row of a leaf-proof check): its flags, labels and input state, and the cubes and seventh powers of every S-box
__global__ __launch_bounds__(64) void p2_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kP2Width * h + r, h};
  const P2Consts* kc = rec.consts;
  const bool real = r < rec.counts[kCountWords * b + 8];
  const uint32_t* row = rec.agg_heap + ((size_t)b * rec.cap_agg + (real ? r : 0)) * kP2RecWords;
  const uint32_t flags = real ? row[0] : 0u, kind = flags & 15u;
  Fp st[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    st[i] = real ? Fp::from_canonical(row[4 + i]) : Fp::zero();
    o.put(P2_IN + i, st[i].v);
  }
  o.flag(P2_IS_REAL, real);
  o.val(P2_KL, real ? row[2] & 0xffff : 0u);
  o.val(P2_KH, real ? row[2] >> 16 : 0u);
  o.val(P2_T, real ? row[1] : 0u);
  o.val(P2_M, real ? row[3] : 0u);
  o.flag(P2_FN, kind == P2K_NODE); o.flag(P2_SZ, kind == P2K_SZ); o.flag(P2_SC, kind == P2K_SC); o.flag(P2_PL, kind == P2K_PL);
  o.flag(P2_PR, kind == P2K_PR); o.flag(P2_FJ, kind == P2K_J);
  o.flag(P2_NEW, (flags & kP2FlagNew) != 0); o.flag(P2_SND, (flags & kP2FlagSnd) != 0); o.flag(P2_FR, (flags & kP2FlagFri) != 0);
  // format v16: the ends of runs and of matrix-row hashes, the Horner sum after this block, alpha_f^1 .. alpha_f^8
  o.flag(P2_RE, (flags & kP2FlagRe) != 0); o.flag(P2_SE, (flags & kP2FlagSe) != 0);
  o.val(P2_RID, real ? row[kP2RecRid] : 0u);
  {
    Fp4 ap;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o.val(P2_SO + i, real ? row[kP2RecSo + i] : 0u);
      ap.c[i] = real ? Fp::from_canonical(row[kP2RecAlpha + i]) : Fp::zero();
    }
    Fp4 pw = ap;
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o.put(P2_AP + 4 * j + i, pw.c[i].v);
      pw = pw * ap;
    }
  }
  p2air_external_linear(st);
  for (int rd = 0; rd < 8; ++rd) {
    if (rd == 4) {
      for (int ir = 0; ir < 13; ++ir) {
        const Fp x = st[0] + Fp::raw(kc->internal[ir]), x3 = x * x * x, y = x3 * x3 * x;
        o.put(P2_INT + 2 * ir, x3.v);
        o.put(P2_INT + 2 * ir + 1, y.v);
        st[0] = y;
        Fp sum = st[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) sum = sum + st[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = st[i] * Fp::raw(kc->diag[i]) + sum;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const Fp x = st[i] + Fp::raw(kc->ext[rd][i]), x3 = x * x * x, y = x3 * x3 * x;
      o.put(P2_EXT + 32 * rd + i, x3.v);
      o.put(P2_EXT + 32 * rd + 16 + i, y.v);
      st[i] = y;
    }
    p2air_external_linear(st);
  }
}

// Query chip: row r is record r - the row itself, as the host verifier computed it while it checked the query (canonical words)
__global__ __launch_bounds__(kMT) void qr_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kQrWidth * h + r, h};
  if (r >= rec.counts[kCountWords * b + 10]) { o.zero(0, kQrWidth); return; }
  const uint32_t* row = rec.fold_rows + ((size_t)b * rec.cap_fold + r) * kQrRecWords;
  for (int c = 0; c < kQrWidth; ++c) o.val(c, row[c]);
}

// Transcript chip: row r is record r (one duplex of a checked leaf's transcript): flags, labels, uses and the input state, and
// the cubes and seventh powers of every S-box of its permutation
__global__ __launch_bounds__(64) void tr_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kTrWidth * h + r, h};
  const P2Consts* kc = rec.consts;
  const bool real = r < rec.counts[kCountWords * b + 12];
  const uint32_t* row = rec.tr_rows + ((size_t)b * rec.cap_tr + (real ? r : 0)) * kTrRecWords;
  const uint32_t flags = real ? row[0] : 0u;
  Fp st[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    st[i] = real ? Fp::from_canonical(row[10 + i]) : Fp::zero();
    o.put(TR_IN + i, st[i].v);
  }
  o.flag(TR_IS_REAL, real);
  o.val(TR_LEAF, real ? row[1] : 0u);
  o.val(TR_STEP, real ? row[2] : 0u);
  o.flag(TR_FIRST, (flags & kTrRecFirst) != 0); o.flag(TR_ABS, (flags & kTrRecAbs) != 0);
  for (int k = 0; k < 7; ++k) o.flag(TR_UROOT + k, ((flags >> k) & 1u) != 0);
  for (int k = 0; k < 8; ++k) o.flag(TR_QM + k, ((flags >> (7 + k)) & 1u) != 0);
  o.val(TR_RIDK, real ? row[3] : 0u);
  o.val(TR_QBASE, real ? row[4] : 0u);
  for (int k = 0; k < 5; ++k) o.val(TR_MROOT + k, real ? row[5 + k] : 0u);
  p2air_external_linear(st);
  for (int rd = 0; rd < 8; ++rd) {
    if (rd == 4) {
      for (int ir = 0; ir < 13; ++ir) {
        const Fp x = st[0] + Fp::raw(kc->internal[ir]), x3 = x * x * x, y = x3 * x3 * x;
        o.put(TR_INT + 2 * ir, x3.v);
        o.put(TR_INT + 2 * ir + 1, y.v);
        st[0] = y;
        Fp sum = st[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) sum = sum + st[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = st[i] * Fp::raw(kc->diag[i]) + sum;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const Fp x = st[i] + Fp::raw(kc->ext[rd][i]), x3 = x * x * x, y = x3 * x3 * x;
      o.put(TR_EXT + 32 * rd + i, x3.v);
      o.put(TR_EXT + 32 * rd + 16 + i, y.v);
      st[i] = y;
    }
    p2air_external_linear(st);
  }
}

// One sub-word-chip instance: row r is event row0 + r of the list sub_idx
__global__ __launch_bounds__(kMT) void sub_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh, uint32_t row0) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kSubWidth * h + r, h};
  const size_t ev = (size_t)row0 + r;
  if (ev >= rec.counts[kCountWords * b + 5]) { o.zero(0, kSubWidth); return; }
  const uint32_t cyc = rec.sub_idx[(size_t)b * rec.cap_sub + ev];
  const uint32_t* cy = rec.cycles + ((size_t)b * rec.cap_cycles + cyc) * 12;
  const uint32_t* p = rec.program + 9 * (size_t)((cy[0] - rec.text_base) >> 2);
  const uint32_t code = p[1], a = cy[1], bb = cy[2], c = cy[3], m = cy[4], mv = cy[5];
  const bool store = code == SB || code == SH;
  const uint32_t off = (bb + p[7]) & 3;
  const uint32_t sel = code == LB ? 0 : code == LH ? 1 : code == LBU ? 2 : code == LHU ? 3 : code == SB ? 4 : 5;
  o.put(SW_IS_REAL, kR1);
  for (uint32_t k = 0; k < 6; ++k) o.flag(SW_SEL + k, k == sel);
  for (uint32_t k = 0; k < 4; ++k) o.flag(SW_O + k, k == off);
  o.limbs(SW_A, store ? mv : a);  // loads: the value loaded; stores: the word left behind
  for (uint32_t k = 0; k < 4; ++k) o.put(SW_MB + k, mont((m >> (8 * k)) & 0xff));
  o.put(SW_CB, store ? mont(c & 0xff) : 0u);
  o.put(SW_CB + 1, store ? mont((c >> 8) & 0xff) : 0u);
  const bool sl = code == LB || code == LH;
  const uint32_t sb = sl ? (m >> (8 * (code == LB ? off : (off | 1)))) & 0xff : 0u;
  o.flag(SW_S, (sb >> 7) != 0);
  o.put(SW_SELB, mont(sb));
}

__device__ __forceinline__ uint64_t m_rol64(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }
__device__ void m_keccak_f(uint64_t* a) {
  const ka::Tables& T = ka::tables();
  for (int r = 0; r < 24; ++r) {
    uint64_t c[5], d[5], bb[25];
#pragma unroll
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ m_rol64(c[(x + 1) % 5], 1);
#pragma unroll
    for (int j = 0; j < 25; ++j) a[j] ^= d[j % 5];
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
      for (int y = 0; y < 5; ++y) bb[y + 5 * ((2 * x + 3 * y) % 5)] = m_rol64(a[x + 5 * y], T.rot[x][y]);
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = bb[x + 5 * y] ^ (~bb[(x + 1) % 5 + 5 * y] & bb[(x + 2) % 5 + 5 * y]);
    a[0] ^= T.rc[r];
  }
}

struct KCall {
  uint32_t ts, ptr;
  uint64_t in[25];
  uint32_t pts[50];
};
static_assert(sizeof(KCall) == 408, "keccak call record layout");

__global__ __launch_bounds__(64) void kmem_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kKmemWidth * h + r, h};
  const uint32_t p = (uint32_t)(r / 50), i = (uint32_t)(r % 50);
  o.zero(0, kKmemWidth);
  o.val(KM_IDX, i);
  o.put(KM_ISF, i == 0 ? kR1 : 0u);
  o.put(KM_ISL, i == 49 ? kR1 : 0u);
  if (p >= rec.counts[kCountWords * b + 1]) return;
  const KCall* k = reinterpret_cast<const KCall*>(rec.kcalls + ((size_t)b * rec.cap_keccak + p) * 408);
  uint64_t st[25];
#pragma unroll
  for (int j = 0; j < 25; ++j) st[j] = k->in[j];
  const uint32_t wi = (uint32_t)(k->in[i >> 1] >> (32 * (i & 1)));
  m_keccak_f(st);
  uint64_t lane = 0;
#pragma unroll
  for (int j = 0; j < 25; ++j) lane = (uint32_t)j == (i >> 1) ? st[j] : lane;
  const uint32_t wo = (uint32_t)(lane >> (32 * (i & 1)));
  o.put(KM_IS_REAL, kR1);
  o.val(KM_TS, k->ts);
  o.val(KM_PTR_LO, k->ptr & 0xffff); o.val(KM_PTR_HI, k->ptr >> 16);
  o.put(KM_CALL, i == 0 ? kR1 : 0u);
  o.val(KM_ADDR, k->ptr + 4 * i);
  o.val(KM_OLD_LO, wi & 0xffff); o.val(KM_OLD_HI, wi >> 16);
  o.val(KM_NEW_LO, wo & 0xffff); o.val(KM_NEW_HI, wo >> 16);
  o.val(KM_PTS, k->pts[i]);
  o.limbs(KM_GL, k->ts + 1 - k->pts[i]);
}

__global__ __launch_bounds__(kMT) void memfinal_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kMemFinalWidth * h + r, h};
  const uint32_t n = rec.counts[kCountWords * b + 2];
  if (r >= n) { o.zero(0, kMemFinalWidth); return; }
  const uint32_t* f = rec.memfinal + ((size_t)b * rec.cap_memfinal + r) * 5;
  o.put(MF_IS_REAL, kR1);
  o.limbs(MF_LO, f[0]);
  o.flag(MF_IS_INIT, f[4] == 1);
  o.flag(MF_IS_ZERO, f[4] == 2);
  o.limbs(MF_INIT_LO, f[1]);
  o.limbs(MF_FIN_LO, f[2]);
  // x0 (row 0) is read once more by every CPU row after the last cycle: its last access is the last row's
  o.val(MF_FIN_TS, r == 0 && rec.cpu_rows > rec.counts[kCountWords * b] ? 4 * rec.cpu_rows : f[3]);
  uint32_t d_lo = 0, d_hi = 0, bw = 0;
  if (r + 1 < n) {  // next address - address - 1, limb-wise with a borrow
    const uint32_t nx = f[5];
    bw = (nx & 0xffff) < (f[0] & 0xffff) + 1;
    d_lo = (nx & 0xffff) + 65536 * bw - (f[0] & 0xffff) - 1;
    d_hi = (nx >> 16) - (f[0] >> 16) - bw;
  }
  o.val(MF_D_LO, d_lo);
  o.val(MF_D_HI, d_hi);
  o.flag(MF_BW, bw != 0);
}

__global__ __launch_bounds__(64) void mul_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kMulWidth * h + r, h};
  if (r >= rec.counts[kCountWords * b + 3]) { o.zero(0, kMulWidth); return; }
  const uint32_t* mu = rec.muls + ((size_t)b * rec.cap_muls + r) * 3;
  const uint32_t bb = mu[1], c = mu[2];
  const uint64_t prod = (uint64_t)bb * c;
  o.put(MU_IS_REAL, kR1);
  o.put(MU_HI, mu[0] == 1 ? kR1 : 0u);
  o.flag(MU_SH, mu[0] == 2); o.flag(MU_SHU, mu[0] == 3);
  {
    // mulh / mulhsu: R + b31 * C + [mulh] c31 * B = P_hi + 2^32 k, limb by limb
    const bool sgd = mu[0] >= 2;
    const uint32_t b31 = bb >> 31, c31 = mu[0] == 2 ? c >> 31 : 0u, phi = (uint32_t)(prod >> 32);
    const uint32_t rr = sgd ? phi - (b31 ? c : 0u) - (c31 ? bb : 0u) : 0u;
    const uint32_t lo = (rr & 0xffff) + (b31 ? (c & 0xffff) : 0u) + (c31 ? (bb & 0xffff) : 0u);
    const uint32_t k0 = sgd ? (lo - (phi & 0xffff)) >> 16 : 0u;
    const uint32_t hi = (rr >> 16) + (b31 ? (c >> 16) : 0u) + (c31 ? (bb >> 16) : 0u) + k0;
    const uint32_t k1 = sgd ? (hi - (phi >> 16)) >> 16 : 0u;
    o.limbs(MU_R, rr);
    o.flag(MU_K0, k0 >= 1); o.flag(MU_K0 + 1, k0 >= 2); o.flag(MU_K1, k1 >= 1); o.flag(MU_K1 + 1, k1 >= 2);
  }
  o.bits(MU_B, bb, 32); o.bits(MU_C, c, 32);
  o.bits(MU_P, (uint32_t)prod, 32); o.bits(MU_P + 32, (uint32_t)(prod >> 32), 32);
  uint64_t s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s[i + j] += (uint64_t)((bb >> (8 * i)) & 0xff) * ((c >> (8 * j)) & 0xff);
  const uint64_t q0 = (s[0] + 256 * s[1]) >> 16, q1 = (s[2] + 256 * s[3] + q0) >> 16, q2 = (s[4] + 256 * s[5] + q1) >> 16;
  o.bits(MU_Q0, (uint32_t)q0, 10); o.bits(MU_Q1, (uint32_t)q1, 11); o.bits(MU_Q2, (uint32_t)q2, 10);
}

// Divider chip: row r is event r of the list div_idx (div divu rem remu): operands and result from the cycle record, the
// absolute values, signs, carries, the product |q| |d| and the zero tests' inverses computed here (air_machine.hpp eval_div)
__global__ __launch_bounds__(kMT) void div_trace_kernel(MachineRecords rec, uint32_t* __restrict__ trace, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const Col o{trace + (size_t)b * kDivWidth * h + r, h};
  if (r >= rec.counts[kCountWords * b + 11]) { o.zero(0, kDivWidth); return; }
  const uint32_t cyc = rec.div_idx[(size_t)b * rec.cap_div + r];
  const uint32_t* cy = rec.cycles + ((size_t)b * rec.cap_cycles + cyc) * 12;
  const uint32_t code = rec.program[9 * (size_t)((cy[0] - rec.text_base) >> 2) + 1], n = cy[2], d = cy[3];
  const bool sg = code == DIV || code == REM, ovf = sg && n == 0x80000000u && d == 0xffffffffu;
  uint32_t q, rm;
  if (d == 0) { q = 0xffffffffu; rm = n; }
  else if (ovf) { q = n; rm = 0; }
  else if (sg) { q = (uint32_t)((int32_t)n / (int32_t)d); rm = (uint32_t)((int32_t)n % (int32_t)d); }
  else { q = n / d; rm = n % d; }
  const uint32_t sn = sg ? n >> 31 : 0u, sd = sg ? d >> 31 : 0u;
  const uint32_t an = sn ? 0u - n : n, ad = sd ? 0u - d : d;
  uint32_t sq, sr, aq, ar;
  if (d == 0) {  // q = 0xffffffff as "minus one", r = n with n's sign
    sq = 1; aq = 1; sr = sn; ar = an;
  } else {
    aq = an / ad; ar = an % ad;  // (|n| = 2^31, |d| = 1 included: |q| = 2^31, shown with the sign flag clear)
    sq = (sn ^ sd) & (aq != 0 ? 1u : 0u);
    sr = sn & (ar != 0 ? 1u : 0u);
  }
  o.put(DV_IS_REAL, kR1);
  o.flag(DV_F + 0, code == DIV); o.flag(DV_F + 1, code == DIVU); o.flag(DV_F + 2, code == REM); o.flag(DV_F + 3, code == REMU);
  o.limbs(DV_N, n); o.limbs(DV_D, d);
  o.limbs(DV_A, (code == DIV || code == DIVU) ? q : rm);
  o.flag(DV_SN, sn != 0); o.flag(DV_SD, sd != 0);
  o.val(DV_NH, (n >> 16) - 32768u * sn); o.val(DV_DH, (d >> 16) - 32768u * sd);
  o.limbs(DV_AN, an); o.limbs(DV_AD, ad); o.limbs(DV_AQ, aq); o.limbs(DV_AR, ar);
  // X + AX = 2^32 where the sign is set: the carry out of the low limbs
  o.flag(DV_CN, sn && (n & 0xffff) != 0); o.flag(DV_CD, sd && (d & 0xffff) != 0);
  o.flag(DV_CQ, sq && (q & 0xffff) != 0); o.flag(DV_CR, sr && (rm & 0xffff) != 0);
  o.limbs(DV_Q, q); o.limbs(DV_R, rm);
  o.flag(DV_SQ, sq != 0); o.flag(DV_SR, sr != 0); o.flag(DV_XS, (sn ^ sd) != 0);
  const uint32_t pl = d == 0 ? 0u : aq * ad;
  o.limbs(DV_PL, pl);
  o.flag(DV_K, d != 0 && (pl & 0xffff) + (ar & 0xffff) > 0xffff);
  const uint32_t e = d == 0 ? 0u : ad - ar - 1;
  o.limbs(DV_E, e);
  o.flag(DV_BE, d != 0 && (ad & 0xffff) < (ar & 0xffff) + 1);
  auto inv_of = [](uint32_t lo, uint32_t hi) { const uint32_t sm = lo + hi; return sm ? Fp::from_canonical(sm).inv().v : 0u; };
  o.flag(DV_NZD, d != 0); o.put(DV_INVD, inv_of(d & 0xffff, d >> 16));
  o.flag(DV_NZQ, aq != 0); o.put(DV_INVQ, inv_of(aq & 0xffff, aq >> 16));
  o.flag(DV_NZR, ar != 0); o.put(DV_INVR, inv_of(ar & 0xffff, ar >> 16));
}

// one main column: multiplicities of the Program table / use flags of the Image table
__global__ __launch_bounds__(kMT) void count_column_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ trace,
                                                          size_t n) {
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= n) return;
  trace[(size_t)blockIdx.y * n + r] = mont(src[(size_t)blockIdx.y * n + r]);
}

// program chip: multiplicities; the padding instruction (last row) is fetched by every CPU row after the last cycle
__global__ __launch_bounds__(kMT) void prog_mult_kernel(MachineRecords rec, uint32_t* __restrict__ trace, size_t n) {
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= n) return;
  const int b = blockIdx.y;
  uint32_t v = rec.prog_mult[(size_t)b * n + r];
  if (r == rec.n_program - 1) v = rec.cpu_rows - rec.counts[kCountWords * b];
  trace[(size_t)b * n + r] = mont(v);
}

// image chip: every word of the image is sent once (the main column repeats the preprocessed is-real flag)
__global__ __launch_bounds__(kMT) void image_used_kernel(uint32_t* __restrict__ trace, size_t n, uint32_t n_image) {
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= n) return;
  trace[(size_t)blockIdx.y * n + r] = r < n_image ? kR1 : 0u;
}

void launch_machine_trace(hipStream_t stream, int chip, const MachineRecords& rec, uint32_t* trace, int logh, int batch) {
  const size_t h = (size_t)1 << logh;
  const dim3 grid((unsigned)((h + kMT - 1) / kMT), batch), block(kMT);
  switch (chip) {
    case kCpu:
    case kCpu2: case kCpu3: case kCpu4: case kCpu5: case kCpu6: case kCpu7: case kCpu8: hipLaunchKernelGGL(cpu_trace_kernel, grid, block, 0, stream, rec, trace, logh, rec.row0[chip]); break;
    case kAlu:
    case kAlu2: hipLaunchKernelGGL(alu_trace_kernel, grid, block, 0, stream, rec, trace, logh, rec.row0[chip]); break;
    case kSub:
    case kSub2: hipLaunchKernelGGL(sub_trace_kernel, grid, block, 0, stream, rec, trace, logh, rec.row0[chip]); break;
    case kBw:
    case kBw2: hipLaunchKernelGGL(bw_trace_kernel, grid, block, 0, stream, rec, trace, logh, rec.row0[chip]); break;
    case kKmem:
      hipLaunchKernelGGL(kmem_trace_kernel, dim3((unsigned)((h + 63) / 64), batch), dim3(64), 0, stream, rec, trace, logh);
      break;
    case kP2:
      hipLaunchKernelGGL(p2_trace_kernel, dim3((unsigned)((h + 63) / 64), batch), dim3(64), 0, stream, rec, trace, logh);
      break;
    case kEcall: hipLaunchKernelGGL(ecall_trace_kernel, grid, block, 0, stream, rec, trace, logh); break;
    case kQr: hipLaunchKernelGGL(qr_trace_kernel, grid, block, 0, stream, rec, trace, logh); break;
    case kHint: hipLaunchKernelGGL(hint_trace_kernel, grid, block, 0, stream, rec, trace, logh); break;
    case kTr:
      hipLaunchKernelGGL(tr_trace_kernel, dim3((unsigned)((h + 63) / 64), batch), dim3(64), 0, stream, rec, trace, logh);
      break;
    case kDiv: hipLaunchKernelGGL(div_trace_kernel, grid, block, 0, stream, rec, trace, logh); break;
    case kMemFinal: hipLaunchKernelGGL(memfinal_trace_kernel, grid, block, 0, stream, rec, trace, logh); break;
    case kMul:
      hipLaunchKernelGGL(mul_trace_kernel, dim3((unsigned)((h + 63) / 64), batch), dim3(64), 0, stream, rec, trace, logh);
      break;
    case kImage: hipLaunchKernelGGL(image_used_kernel, grid, block, 0, stream, trace, h, rec.n_image); break;
    case kProgram: hipLaunchKernelGGL(prog_mult_kernel, grid, block, 0, stream, rec, trace, h); break;
    default: break;
  }
}

__global__ __launch_bounds__(kMT) void keccak_ts_kernel(MachineRecords rec, uint32_t* __restrict__ trace, size_t bstride, int logh) {
  const size_t h = (size_t)1 << logh;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= h) return;
  const int b = blockIdx.y;
  const uint32_t p = (uint32_t)(r / 24);
  uint32_t v = 0;
  if (p < rec.counts[kCountWords * b + 1]) v = mont(reinterpret_cast<const KCall*>(rec.kcalls + ((size_t)b * rec.cap_keccak + p) * 408)->ts);
  trace[(size_t)b * bstride + (size_t)KC_TS * h + r] = v;
}
void launch_keccak_ts(hipStream_t stream, const MachineRecords& rec, uint32_t* trace, size_t trace_bstride, int logh, int batch) {
  const size_t h = (size_t)1 << logh;
  hipLaunchKernelGGL(keccak_ts_kernel, dim3((unsigned)((h + kMT - 1) / kMT), batch), dim3(kMT), 0, stream, rec, trace, trace_bstride, logh);
}

// ===========================================================================================
// mixed-height Merkle commitment
// ===========================================================================================
struct LeafArgs {
  Seg seg[kMaxSegs];
  int start[kMaxSegs + 1];  // first virtual column of each segment; start[nseg] = total width
  int nseg, logh;
  uint32_t* out;
  size_t out_bstride;
};

__global__ __launch_bounds__(kMT) void mmcs_leaf_kernel(LeafArgs a, const P2Consts* __restrict__ consts) {
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  const size_t r = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (r >= n) return;
  const int b = blockIdx.y;
  int32_t s[16];  // the sponge state stays in signed lazy form between permutations (poseidon2.hpp)
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = 0;
  const int total = a.start[a.nseg];
  int sg = 0;
  const uint32_t* base = a.seg[0].p + (size_t)b * a.seg[0].bstride + r;
  for (int c0 = 0; c0 < total; c0 += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int vc = c0 + i;
      if (vc < total) {
        while (vc >= a.start[sg + 1]) {  // uniform across the grid
          ++sg;
          base = a.seg[sg].p + (size_t)b * a.seg[sg].bstride + r;
        }
        s[i] = (int32_t)base[(size_t)(vc - a.start[sg]) * n];
      } else {
        s[i] = 0;  // the last block is zero-filled
      }
    }
    p2_permute_signed(s, consts);
  }
  const size_t c = r >> a.logh, m = r & (h - 1);
  const size_t pos = c * h + (a.logh ? (size_t)(__brev((uint32_t)m) >> (32 - a.logh)) : 0);
  uint4* d = reinterpret_cast<uint4*>(a.out + (size_t)b * a.out_bstride + pos * 8);
  d[0] = make_uint4(fps_canon(s[0]), fps_canon(s[1]), fps_canon(s[2]), fps_canon(s[3]));
  d[1] = make_uint4(fps_canon(s[4]), fps_canon(s[5]), fps_canon(s[6]), fps_canon(s[7]));
}

// Latency form for groups with few rows (the 2 634-column keccak chip of a single proof has 4 096 of them and
// 330 dependent permutations per row): 16 lanes per row, one sponge word each (poseidon2_coop.hpp), absorbed
// words fetched four steps ahead.  About twice the arithmetic of the lane-per-row kernel, so it is used only
// while that kernel could not fill the chip.
__global__ __launch_bounds__(kMT) void mmcs_leaf_coop_kernel(LeafArgs a, const P2Consts* __restrict__ consts) {
  // a chain of dependent permutations that shares the CUs with the throughput kernels of other lanes (a small batch hashes
  // its height groups side by side): its waves issue first, the others fill the gaps its dependencies leave
  __builtin_amdgcn_s_setprio(3);
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  const int e = threadIdx.x & 15;
  const size_t row = (size_t)blockIdx.x * (kMT / 16) + (threadIdx.x >> 4);
  const bool act = row < n;
  const size_t r = act ? row : 0;
  const int b = blockIdx.y;
  const CoopConsts cc = coop_load_consts(consts, e);
  const int total = a.start[a.nseg];
  auto fetch = [&](int c) -> uint32_t {
    const int vc = c + e;
    const uint32_t* p = nullptr;  // constant indices into the kernel arguments: no copy of them to scratch
#pragma unroll
    for (int sg = 0; sg < kMaxSegs; ++sg)
      if (e < 8 && vc >= a.start[sg] && vc < a.start[sg + 1])
        p = a.seg[sg].p + (size_t)b * a.seg[sg].bstride + (size_t)(vc - a.start[sg]) * n + r;
    return p ? *p : 0u;
  };
  constexpr int kPrefetch = 4;
  uint32_t ahead[kPrefetch];
#pragma unroll
  for (int j = 0; j < kPrefetch; ++j) ahead[j] = fetch(8 * j);
  int32_t x = 0;
  for (int c0 = 0; c0 < total; c0 += 8 * kPrefetch) {
    uint32_t cur[kPrefetch];
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) cur[j] = ahead[j];
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) ahead[j] = fetch(c0 + 8 * (kPrefetch + j));
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) {
      const int c = c0 + 8 * j;
      if (c < total) {  // uniform
        if (e < 8) x = (int32_t)cur[j];  // (fetch() returns zero beyond the width: the last block is zero-filled)
        x = p2_permute_coop_signed(x, cc, consts);
      }
    }
  }
  if (act && e < 8) {
    const size_t c = row >> a.logh, m = row & (h - 1);
    const size_t pos = c * h + (a.logh ? (size_t)(__brev((uint32_t)m) >> (32 - a.logh)) : 0);
    a.out[(size_t)b * a.out_bstride + pos * 8 + e] = fps_canon(x);
  }
}

void launch_mmcs_leaves(hipStream_t stream, const Seg* segs, int nseg, int logh, uint32_t* digests, size_t out_bstride,
                        int batch, const P2Consts* consts) {
  LeafArgs a;
  a.nseg = 0;
  a.start[0] = 0;
  for (int i = 0; i < nseg && a.nseg < kMaxSegs; ++i) {
    if (segs[i].width == 0) continue;
    a.seg[a.nseg] = segs[i];
    a.start[a.nseg + 1] = a.start[a.nseg] + segs[i].width;
    a.nseg++;
  }
  for (int i = a.nseg; i < kMaxSegs; ++i) { a.seg[i] = Seg{nullptr, 0, 0}; a.start[i + 1] = a.start[a.nseg]; }
  a.logh = logh;
  a.out = digests;
  a.out_bstride = out_bstride;
  const size_t n = (size_t)2 << logh;
  // fewer than 64 K rows batch-wide (1 024 waves, one per SIMD) and more than a few permutations each: latency form
  if (n * (size_t)batch < 65536 && a.start[a.nseg] > 32) {
    constexpr int rows_per_block = kMT / 16;
    hipLaunchKernelGGL(mmcs_leaf_coop_kernel, dim3((unsigned)((n + rows_per_block - 1) / rows_per_block), batch), dim3(kMT), 0,
                       stream, a, consts);
    return;
  }
  hipLaunchKernelGGL(mmcs_leaf_kernel, dim3((unsigned)((n + kMT - 1) / kMT), batch), dim3(kMT), 0, stream, a, consts);
}

__device__ __forceinline__ void m_compress(const uint4* l, const uint4* r, uint4* out, const P2Consts* __restrict__ consts) {
  const uint4 a = l[0], b = l[1], c = r[0], d = r[1];
  Fp s[16] = {Fp::raw(a.x), Fp::raw(a.y), Fp::raw(a.z), Fp::raw(a.w), Fp::raw(b.x), Fp::raw(b.y), Fp::raw(b.z), Fp::raw(b.w),
              Fp::raw(c.x), Fp::raw(c.y), Fp::raw(c.z), Fp::raw(c.w), Fp::raw(d.x), Fp::raw(d.y), Fp::raw(d.z), Fp::raw(d.w)};
  p2_permute(s, consts);
  out[0] = make_uint4(s[0].v, s[1].v, s[2].v, s[3].v);
  out[1] = make_uint4(s[4].v, s[5].v, s[6].v, s[7].v);
}

__global__ __launch_bounds__(kMT) void mmcs_level_kernel(const uint32_t* __restrict__ in, size_t in_bstride,
                                                        uint32_t* __restrict__ out, size_t out_bstride,
                                                        const uint32_t* __restrict__ inject, size_t inject_bstride, size_t count,
                                                        const P2Consts* __restrict__ consts) {
  const size_t i = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (i >= count) return;
  const int b = blockIdx.y;
  const uint4* src = reinterpret_cast<const uint4*>(in + (size_t)b * in_bstride + 16 * i);
  uint4 d[2];
  m_compress(src, src + 2, d, consts);
  if (inject) {
    const uint4* g = reinterpret_cast<const uint4*>(inject + (size_t)b * inject_bstride + 8 * i);
    uint4 e[2];
    m_compress(d, g, e, consts);
    d[0] = e[0];
    d[1] = e[1];
  }
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)b * out_bstride + 8 * i);
  dst[0] = d[0];
  dst[1] = d[1];
}

void launch_mmcs_level(hipStream_t stream, const uint32_t* in, size_t in_bstride, uint32_t* out, size_t out_bstride,
                       const uint32_t* inject, size_t inject_bstride, size_t count, int batch, const P2Consts* consts) {
  hipLaunchKernelGGL(mmcs_level_kernel, dim3((unsigned)((count + kMT - 1) / kMT), batch), dim3(kMT), 0, stream, in, in_bstride,
                     out, out_bstride, inject, inject_bstride, count, consts);
}

// Several levels in ONE launch.  gridDim.x = 1: the levels of at most kMT nodes, one workgroup per proof (nine launches of a
// few microseconds' work each were a fifth of a single proof's commitment).  gridDim.x = S: the levels above those, S
// subtrees side by side - workgroup s climbs nodes s * count / S .. of every level, which only depend on its own nodes of
// the level below.  (Stores to the tree are visible to the workgroup after the barrier's fence.)
__global__ __launch_bounds__(kMT) void mmcs_top_kernel(MmcsTopArgs a, const P2Consts* __restrict__ consts) {
  const int sub = blockIdx.x, b = blockIdx.y;
  uint32_t* tree = a.tree + (size_t)b * a.tree_bstride;
  for (int lv = 0; lv < a.n_levels; ++lv) {
    const int per = a.count[lv] / (int)gridDim.x;
    for (int k = threadIdx.x; k < per; k += kMT) {
      const size_t i = (size_t)sub * per + k;
      const uint4* src = reinterpret_cast<const uint4*>(tree + a.in_off[lv] + 16 * i);
      uint4 d[2];
      m_compress(src, src + 2, d, consts);
      if (a.inject[lv]) {
        const uint4* g = reinterpret_cast<const uint4*>(a.inject[lv] + (size_t)b * a.inj_bstride[lv] + 8 * i);
        uint4 e[2];
        m_compress(d, g, e, consts);
        d[0] = e[0];
        d[1] = e[1];
      }
      uint4* dst = reinterpret_cast<uint4*>(tree + a.out_off[lv] + 8 * i);
      dst[0] = d[0];
      dst[1] = d[1];
    }
    __threadfence_block();
    __syncthreads();
  }
}
void launch_mmcs_top(hipStream_t stream, const MmcsTopArgs& a, int subtrees, int batch, const P2Consts* consts) {
  hipLaunchKernelGGL(mmcs_top_kernel, dim3(subtrees, batch), dim3(kMT), 0, stream, a, consts);
}

// ===========================================================================================
// LogUp: fingerprints and the permutation trace
// ===========================================================================================
struct RowView {
  const uint32_t* prep;
  const uint32_t* main_;
  int prep_w;
  size_t cs;
  __device__ __forceinline__ Fp at(int col) const {
    return Fp::raw(col < prep_w ? prep[(size_t)col * cs] : main_[(size_t)(col - prep_w) * cs]);
  }
};
__device__ __forceinline__ Fp m_lf_eval(const LinForm& f, const RowView& rv) {
  Fp v = Fp::raw(f.c0);
  for (int i = 0; i < f.n; ++i) v += Fp::raw(f.coef[i]) * rv.at(f.col[i]);
  return v;
}
__device__ __forceinline__ Fp4 m_fingerprint(const Interaction& it, const RowView& rv, const Fp4& gamma, const uint32_t* bpow) {
  Fp4 f = gamma;
  f.c[0] += Fp::from_canonical((uint32_t)it.bus);
  for (int j = 0; j < it.n_el; ++j) f += m_load_fp4(bpow + 4 * (size_t)(j + 1)) * m_lf_eval(it.el[j], rv);
  return f;
}

// ===========================================================================================
// table chip multiplicities: the RANGE / BYTES receives of a chip's rows, counted per table row
// ===========================================================================================
constexpr int kTableRowsPerBlock = 4096;
constexpr uint32_t kTableLdsBins = 4096;  // the low range16 values, the small high address limbs, the aligned low limbs below
                                          // 2^14 and the byte pairs with a small second byte are hot: counted in LDS, flushed
                                          // once per workgroup
constexpr size_t kTableRows = (size_t)1 << kTableLogH;
// bins[idx] += m for the active lanes of a wave.  Lanes that share a bin (in these histograms most of the wave: a zero high
// byte, a gap of 3, the one stack page a run's loads hit) are added with ONE atomic: up to four leaders in turn collect the
// lanes that agree with them, whoever is left adds for itself.  Same-address atomics serialise in the memory system
// (about 130 ns each at device scope: 4 096 keccak-state rows that look up the same pointer limb took 530 us).
__device__ __forceinline__ void wave_hist_add(uint32_t* bins, uint32_t idx, uint32_t m, bool active) {
  const int lane = threadIdx.x & 63;
  unsigned long long act = __ballot(active);
#pragma unroll 1
  for (int round = 0; round < 4 && act; ++round) {
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t k0 = __shfl(idx, leader, 64), m0 = __shfl(m, leader, 64);
    const bool same = active && idx == k0 && m == m0;
    const unsigned long long sm = __ballot(same);
    if (lane == leader) atomicAdd(&bins[k0], m0 * (uint32_t)__popcll(sm));
    active = active && !same;
    act &= ~sm;
  }
  if (active) atomicAdd(&bins[idx], m);
}
__global__ __launch_bounds__(kMT) void table_count_kernel(const Interaction* __restrict__ inter, int n_inter,
                                                         const uint32_t* __restrict__ trace, int width, int logh,
                                                         uint32_t* __restrict__ hist, int rows_per_block) {
  __shared__ uint32_t lds[4 * kTableLdsBins];  // range16, high address limb, byte pair, aligned low limb / 4
  const size_t h = (size_t)1 << logh;
  const int b = blockIdx.y;
  for (uint32_t i = threadIdx.x; i < 4 * kTableLdsBins; i += kMT) lds[i] = 0;
  __syncthreads();
  uint32_t* hb = hist + (size_t)b * kTableWidth * kTableRows;
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  for (size_t rr = threadIdx.x; rr < (size_t)rows_per_block; rr += kMT) {  // every lane of a wave takes every trip: the
    const size_t r = r0 + rr;                                            // wave-level aggregation needs converged lanes
    const bool in_range = r < h;
    const RowView rv{nullptr, trace + (size_t)b * width * h + (in_range ? r : 0), 0, h};
    for (int k = 0; k < n_inter; ++k) {
      const Interaction& it = inter[k];
      if (it.sign > 0 || (it.bus != BUS_RANGE && it.bus != BUS_BYTES && it.bus != BUS_BYTEOP)) continue;
      const uint32_t m = in_range ? m_lf_eval(it.mult, rv).to_canonical() : 0u;
      const uint32_t v1 = m_lf_eval(it.el[0], rv).to_canonical(), v2 = m_lf_eval(it.el[1], rv).to_canonical();
      if (it.bus == BUS_BYTEOP) {  // (kind, x, y, z): counted only if z is the table's answer; byte pairs are spread out
        const uint32_t y = m_lf_eval(it.el[2], rv).to_canonical(), z = m_lf_eval(it.el[3], rv).to_canonical();
        const uint32_t want = v1 == 1 ? (v2 ^ y) : v1 == 2 ? (v2 | y) : (v2 & y);
        const bool ok = m != 0 && v1 >= 1 && v1 <= 3 && v2 <= 255 && y <= 255 && z == want;
        wave_hist_add(hb + (size_t)TB_M_XOR * kTableRows, (v1 - 1) * (uint32_t)kTableRows + v2 + 256 * y, m, ok);
        continue;
      }
      // a value without a table row is not counted: the buses of such a (dishonest or unprovable) run do not balance
      if (it.bus == BUS_RANGE) {
        const bool ok = m != 0 && v2 < kTableRows && v1 <= 2 && !(v1 == 1 && (v2 & 3)) && !(v1 == 2 && (v2 == 0 || v2 > kAddrHiMax));
        const bool hot = v1 == 1 ? v2 < 4 * kTableLdsBins : v2 < kTableLdsBins;
        wave_hist_add(lds, v1 == 1 ? 3 * kTableLdsBins + (v2 >> 2) : (v1 == 2 ? kTableLdsBins : 0) + v2, m, ok && hot);  // (the kind differs between lanes)
        wave_hist_add(hb, (size_t)(v1 == 0 ? TB_M_R16 : v1 == 1 ? TB_M_AL : TB_M_TOP) * kTableRows + v2, m, ok && !hot);
      } else {
        const bool ok = m != 0 && v1 <= 255 && v2 <= 255;
        const uint32_t idx = v1 + 256 * v2;
        const bool hot = idx < kTableLdsBins;
        wave_hist_add(lds, 2 * kTableLdsBins + idx, m, ok && hot);
        wave_hist_add(hb, (uint32_t)(TB_M_BY * kTableRows) + idx, m, ok && !hot);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kTableLdsBins; i += kMT) {
    if (lds[i]) atomicAdd(&hb[(size_t)TB_M_R16 * kTableRows + i], lds[i]);
    if (lds[kTableLdsBins + i]) atomicAdd(&hb[(size_t)TB_M_TOP * kTableRows + i], lds[kTableLdsBins + i]);
    if (lds[2 * kTableLdsBins + i]) atomicAdd(&hb[(size_t)TB_M_BY * kTableRows + i], lds[2 * kTableLdsBins + i]);
    if (lds[3 * kTableLdsBins + i]) atomicAdd(&hb[(size_t)TB_M_AL * kTableRows + 4 * i], lds[3 * kTableLdsBins + i]);
  }
}
// The CPU chip's seven lookups (machine_defs.cpp g_cpu[7..13]) read from the columns that hold them, instead of through
// the generic linear forms: three range16 gaps, two pairs of high bytes, the adder output's high limb (kind 2 where it
// is an address) and its low limb less the byte offset (kind 1 where aligned), the last two on the rows that check X.
__global__ __launch_bounds__(kMT) void cpu_table_count_kernel(const uint32_t* __restrict__ trace, int logh, uint32_t* __restrict__ hist,
                                                              int rows_per_block) {
  __shared__ uint32_t lds[4 * kTableLdsBins];
  const size_t h = (size_t)1 << logh;
  const int b = blockIdx.y;
  for (uint32_t i = threadIdx.x; i < 4 * kTableLdsBins; i += kMT) lds[i] = 0;
  __syncthreads();
  uint32_t* hb = hist + (size_t)b * kTableWidth * kTableRows;
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  auto range = [&](uint32_t kind, uint32_t v, uint32_t m) {
    const bool ok = m != 0 && v < kTableRows && kind <= 2 && !(kind == 1 && (v & 3)) && !(kind == 2 && (v == 0 || v > kAddrHiMax));
    const bool hot = kind == 1 ? v < 4 * kTableLdsBins : v < kTableLdsBins;
    wave_hist_add(lds, kind == 1 ? 3 * kTableLdsBins + (v >> 2) : (kind == 2 ? kTableLdsBins : 0) + v, m, ok && hot);
    wave_hist_add(hb, (uint32_t)((kind == 0 ? TB_M_R16 : kind == 1 ? TB_M_AL : TB_M_TOP) * kTableRows) + v, m, ok && !hot);
  };
  auto bytes = [&](uint32_t v1, uint32_t v2, bool in_range) {
    const bool ok = in_range && v1 <= 255 && v2 <= 255;
    const uint32_t idx = v1 + 256 * v2;
    const bool hot = idx < kTableLdsBins;
    wave_hist_add(lds, 2 * kTableLdsBins + idx, 1u, ok && hot);
    wave_hist_add(hb, (uint32_t)(TB_M_BY * kTableRows) + idx, 1u, ok && !hot);
  };
  for (size_t rr = threadIdx.x; rr < (size_t)rows_per_block; rr += kMT) {  // (converged lanes, as in table_count_kernel)
    const size_t r = r0 + rr;
    const bool in_range = r < h;
    const uint32_t* row = trace + (size_t)b * kCpuWidth * h + (in_range ? r : 0);
    auto col = [&](int c) { return Fp::raw(row[(size_t)c * h]); };
    auto can = [&](int c) { return col(c).to_canonical(); };
    uint32_t g[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = can(C_GAP + i);
    const uint32_t one = in_range ? 1u : 0u;
    range(0, g[0], one); range(0, g[2], one); range(0, g[4], one);
    bytes(g[1], g[3], in_range); bytes(g[5], 0, in_range);
    const Fp al = col(selc(CL_JALR)) + col(selc(CL_LW)) + col(selc(CL_SW)) + col(selc(CL_LDS)) + col(selc(CL_STS));
    const Fp top = al + col(selc(CL_KECCAK));
    const Fp chk = top + col(selc(CL_ADD)) + col(selc(CL_SUB)) + col(selc(CL_ECALL)) + col(C_UC);
    const Fp off = col(C_O1) + col(C_O2).dbl() + Fp::raw(cmonty(3)) * col(C_O3);
    const uint32_t m = in_range ? chk.to_canonical() : 0u;
    range(top.dbl().to_canonical(), can(C_X + 1), m);
    range(al.to_canonical(), (col(C_X) - off).to_canonical(), m);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kTableLdsBins; i += kMT) {
    if (lds[i]) atomicAdd(&hb[(size_t)TB_M_R16 * kTableRows + i], lds[i]);
    if (lds[kTableLdsBins + i]) atomicAdd(&hb[(size_t)TB_M_TOP * kTableRows + i], lds[kTableLdsBins + i]);
    if (lds[2 * kTableLdsBins + i]) atomicAdd(&hb[(size_t)TB_M_BY * kTableRows + i], lds[2 * kTableLdsBins + i]);
    if (lds[3 * kTableLdsBins + i]) atomicAdd(&hb[(size_t)TB_M_AL * kTableRows + 4 * i], lds[3 * kTableLdsBins + i]);
  }
}
// rows a workgroup counts before it adds its LDS histograms to the table's: 4 096, or 512 where a small batch would
// otherwise run sixteen workgroups on the whole GPU
static int table_rows_per_block(int batch) { return batch < 8 ? 512 : kTableRowsPerBlock; }
void launch_cpu_table_count(hipStream_t stream, const uint32_t* trace, int logh, const MachineRecords& rec, int batch) {
  const size_t h = (size_t)1 << logh;
  const int rpb = table_rows_per_block(batch);
  hipLaunchKernelGGL(cpu_table_count_kernel, dim3((unsigned)((h + rpb - 1) / rpb), batch), dim3(kMT), 0, stream, trace, logh,
                     rec.table_hist, rpb);
}
void launch_table_clear(hipStream_t stream, const MachineRecords& rec, int batch) {
  (void)hipMemsetAsync(rec.table_hist, 0, (size_t)batch * kTableWidth * kTableRows * 4, stream);
}
void launch_table_count(hipStream_t stream, const Interaction* inter, int n_inter, const uint32_t* trace, int width, int logh,
                        const MachineRecords& rec, int batch) {
  const size_t h = (size_t)1 << logh;
  const int rpb = table_rows_per_block(batch);
  hipLaunchKernelGGL(table_count_kernel, dim3((unsigned)((h + rpb - 1) / rpb), batch), dim3(kMT), 0, stream, inter, n_inter, trace,
                     width, logh, rec.table_hist, rpb);
}
void launch_table_trace(hipStream_t stream, const MachineRecords& rec, uint32_t* trace, int batch) {
  const size_t n = kTableWidth * kTableRows;
  hipLaunchKernelGGL(count_column_kernel, dim3((unsigned)((n + kMT - 1) / kMT), batch), dim3(kMT), 0, stream, rec.table_hist, trace, n);
}

// The CPU chip's 19 bus interactions evaluated from values a row holds once (the class id, the selector sums, the
// three previous access times), instead of through the generic linear forms (which reload and rescale every column for
// every tuple): the same field elements, a fraction of the work.  visit(j, ma, fa, mb, fb) is called for the LogUp
// slots j = 0..7 in order (machine_defs.hpp "LogUp layout") with the slot's value ma / fa + mb / fb: slots 0..6 are
// pairs, slot 7 the five merged sends, M / F with M = sum m_k and F = sum m_k f_k + 1 - M.  Multiplicities are signed.
// Must restate machine_defs.cpp's g_cpu[] exactly: the whole-proof parity tests compare against the oracle's generic
// evaluation.
// J0, J1: only the slots J0 <= j < J1 are visited (the loads the others need are dead code): the kernels below
// evaluate the slots in kCpuBusGroups launches, because all the extension-field fingerprints at once do not fit the
// register file (256 VGPRs and one wave per SIMD when they are evaluated together).
constexpr int kCpuSlots = 8, kCpuHelpers = kCpuSlots - 1, kCpuBusGroups = 2;
__host__ __device__ constexpr int cpu_group_lo(int g) { return g == 0 ? 0 : 4; }
__host__ __device__ constexpr int cpu_group_hi(int g) { return g == 0 ? 4 : kCpuSlots; }
template <int J0, int J1, class V>
__device__ __forceinline__ void cpu_bus_pairs(const uint32_t* __restrict__ row, size_t cs, const Fp4& gamma,
                                              const uint32_t* __restrict__ bpow, V&& visit_all) {
  auto visit = [&](int j, Fp ma, const Fp4& fa, Fp mb, const Fp4& fb) {
    if (j >= J0 && j < J1) visit_all(j, ma, fa, mb, fb);
  };
  auto col = [&](int c) { return Fp::raw(row[(size_t)c * cs]); };
  const Fp one = Fp::one(), three = Fp::raw(cmonty(3)), k65536 = Fp::raw(cmonty(65536));
  Fp sel[kNumCls + 1];
  Fp clsid = Fp::zero(), kf = Fp::zero();
#pragma unroll
  for (int k = 1; k <= kNumCls; ++k) {
    kf = kf + one;
    sel[k] = col(selc(k));
    clsid = clsid + kf * sel[k];
  }
  const Fp memw = sel[CL_LW] + sel[CL_SW] + sel[CL_LDS] + sel[CL_STS];
  const Fp al = memw + sel[CL_JALR], top = al + sel[CL_KECCAK];
  const Fp uc = col(C_UC);
  const Fp chk = top + sel[CL_ADD] + sel[CL_SUB] + sel[CL_ECALL] + uc;
  const Fp alu = sel[CL_ALU] + sel[CL_BLT] + sel[CL_BGE] - uc;
  const Fp ts = col(C_TS), wr = col(C_WR), use2 = col(C_USE2), rd = col(C_RD), rs1 = col(C_RS1), rs2 = col(C_RS2), code = col(C_CODE);
  const Fp a_lo = col(C_A), a_hi = col(C_A + 1), b_lo = col(C_B), b_hi = col(C_B + 1), c_lo = col(C_C), c_hi = col(C_C + 1),
           x_lo = col(C_X), x_hi = col(C_X + 1);
  const Fp o1 = col(C_O1), o2 = col(C_O2), o3 = col(C_O3), off = o1 + o2.dbl() + three * o3;
  Fp g[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) g[i] = col(C_GAP + i);
  // A fingerprint gamma + bus + sum_j beta^(j+1) t_j is a dot product of uniform extension elements with the row's base-field
  // values: four signed 64-bit accumulators, one multiply-add per term and coordinate (the powers of beta and the
  // values centred into (-p/2, p/2], so up to nine terms fit), folded once - instead of a canonical extension-by-base
  // product and an extension addition per term (32 instructions against 4).
  struct Beta { int32_t c[4]; };
  auto beta = [&](int j) {  // beta^j, centred (uniform: scalar loads and scalar arithmetic)
    Beta r;
#pragma unroll
    for (int t = 0; t < 4; ++t) r.c[t] = fps_centre(bpow[4 * j + t]);
    return r;
  };
  struct Acc {
    int64_t a[4] = {0, 0, 0, 0};
    __device__ __forceinline__ void mad(const Beta& bt, Fp x) {
      const int32_t xc = fps_centre(x.v);
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] += (int64_t)bt.c[t] * (int64_t)xc;
    }
    __device__ __forceinline__ Fp4 plus(const Fp4& g) const {
      Fp4 r;
#pragma unroll
      for (int t = 0; t < 4; ++t) r.c[t] = Fp::raw(fps_canon(fps_fold(a[t]))) + g.c[t];
      return r;
    }
  };
  const Beta b1 = beta(1), b2 = beta(2), b3 = beta(3), b4 = beta(4);
  auto busc = [&](int bus) {
    Fp4 f = gamma;
    f.c[0] += Fp::raw(cmonty((uint32_t)bus));
    return f;
  };
  const Fp4 gmem = busc(BUS_MEM), grng = busc(BUS_RANGE), gbyt = busc(BUS_BYTES);
  auto mem = [&](Fp addr, Fp lo, Fp hi, Fp t) {
    Acc f;
    f.mad(b1, addr); f.mad(b2, lo); f.mad(b3, hi); f.mad(b4, t);
    return f.plus(gmem);
  };
  auto pair = [&](const Fp4& g0, Fp v1, Fp v2) {  // (kind, value) or (x, y)
    Acc f;
    f.mad(b1, v1); f.mad(b2, v2);
    return f.plus(g0);
  };
  // previous access time of slot q (accessed at ts + q): ts + q - 1 - (gap_lo + 2^16 gap_hi)
  auto pts = [&](int q) { return ts - (g[2 * q] + k65536 * g[2 * q + 1]) + (q == 0 ? -one : q == 1 ? Fp::zero() : one); };
  // the second access (rs2, or a load's word, whose value sits in C) and the written location (rd, or a store's word:
  // new value in A, old value in W_P)
  const Fp m2 = use2 + sel[CL_LW] + sel[CL_LDS], m3 = wr + sel[CL_SW] + sel[CL_STS];
  const Fp ad2 = col(C_ADDR2), ad3 = col(C_ADDR3);
  if (J0 <= 0 && 0 < J1) {  // instruction fetch (receive), rs1 consume
    Acc f0, f1;
    f0.mad(b1, col(C_PC)); f0.mad(b2, clsid); f0.mad(b3, code); f0.mad(b4, uc);
    f0.mad(beta(5), wr); f0.mad(beta(6), use2); f0.mad(beta(7), rd);
    f1.mad(beta(8), rs1); f1.mad(beta(9), rs2); f1.mad(beta(10), col(C_IMM_LO)); f1.mad(beta(11), col(C_IMM_HI));
    f1.mad(beta(12), col(C_TGT_LO)); f1.mad(beta(13), col(C_TGT_HI));
    visit(0, -one, f0.plus(f1.plus(busc(BUS_PROG))), -one, mem(rs1, b_lo, b_hi, pts(0)));
  }
  if (J0 <= 1 && 1 < J1) visit(1, one, mem(rs1, b_lo, b_hi, ts), -m2, mem(ad2, c_lo, c_hi, pts(1)));
  if (J0 <= 2 && 2 < J1) visit(2, m2, mem(ad2, c_lo, c_hi, ts + one), -m3, mem(ad3, col(C_W_PLO), col(C_W_PHI), pts(2)));
  // the low limbs of the three access-time differences (range16), their high bytes
  if (J0 <= 3 && 3 < J1) visit(3, m3, mem(ad3, a_lo, a_hi, ts + one.dbl()), -one, pair(grng, Fp::zero(), g[0]));
  if (J0 <= 4 && 4 < J1) visit(4, -one, pair(grng, Fp::zero(), g[2]), -one, pair(grng, Fp::zero(), g[4]));
  if (J0 <= 5 && 5 < J1) visit(5, -one, pair(gbyt, g[1], g[3]), -one, pair(gbyt, g[5], Fp::zero()));
  // the adder output: high limb (kind 2 where it is an address), low limb less the byte offset (kind 1 where aligned)
  if (J0 <= 6 && 6 < J1) visit(6, -chk, pair(grng, top.dbl(), x_hi), -chk, pair(grng, al, x_lo - off));
  if (J0 <= 7 && 7 < J1) {
    // one instruction class each: ALU-chip sends, sub-word loads and stores, the keccak call, the ecall hand-over.
    // F = sum_k m_k f_k + 1 - M with f_k = gamma + bus_k + sum_j beta^(j+1) t_kj: the tuples are blended in the base
    // field first (T_j = sum_k m_k t_kj, the m_k boolean and exclusive), then ONE fingerprint is accumulated.
    //   ALU   (op, a_lo, a_hi, b_lo, b_hi, c_lo, c_hi)
    //   SUB   (op, offset, a_lo, a_hi, w_lo, w_hi, c_lo of a store): a load has the word in C, a store in W_P
    //   KCALL (ts, c_lo, c_hi)      ECALL (ts, pc, next pc, b_lo, a_lo, a_hi)
    const Fp lds = sel[CL_LDS], sts = sel[CL_STS], sub = lds + sts, kec = sel[CL_KECCAK], ecl = sel[CL_ECALL];
    const Fp w_lo = lds * c_lo + sts * col(C_W_PLO), w_hi = lds * c_hi + sts * col(C_W_PHI);
    const Fp msum = alu + sub + kec + ecl;
    Acc f;
    f.mad(b1, (alu + sub) * code + (kec + ecl) * ts);
    f.mad(b2, alu * a_lo + sub * off + kec * c_lo + ecl * col(C_PC));
    f.mad(b3, (alu * a_hi + sub * a_lo) + (kec * c_hi + ecl * col(C_NEXT_PC)));
    f.mad(b4, (alu + ecl) * b_lo + sub * a_hi);
    f.mad(beta(5), alu * b_hi + w_lo + ecl * a_lo);
    f.mad(beta(6), alu * c_lo + w_hi + ecl * a_hi);
    f.mad(beta(7), (alu * c_hi) + (sts * c_lo));
    // sum_k m_k (gamma + bus_k) + 1 - M
    Fp4 g0 = gamma * msum;
    g0.c[0] += alu * Fp::raw(cmonty((uint32_t)BUS_ALU)) + sub * Fp::raw(cmonty((uint32_t)BUS_SUB)) +
               kec * Fp::raw(cmonty((uint32_t)BUS_KCALL)) + ecl * Fp::raw(cmonty((uint32_t)BUS_ECALL)) + one - msum;
    visit(7, msum, f.plus(g0), Fp::zero(), Fp4::one());
  }
}

// inverses of n extension elements with one inversion (Montgomery's trick); the elements are fingerprints, non-zero
// except with negligible probability
template <int N>
__device__ __forceinline__ void batch_inverse(Fp4* f) {
  Fp4 pre[N];
  pre[0] = f[0];
#pragma unroll
  for (int i = 1; i < N; ++i) pre[i] = pre[i - 1] * f[i];
  Fp4 inv = pre[N - 1].inv();
#pragma unroll
  for (int i = N - 1; i >= 1; --i) {
    const Fp4 fi = f[i];
    f[i] = inv * pre[i - 1];
    inv = inv * fi;
  }
  f[0] = inv;
}

// slots of group G (cpu_group_lo .. cpu_group_hi): their helper columns (the last slot has none) and their share of the
// row sum, which accumulates over the three launches
template <int G>
__device__ __forceinline__ void perm_terms_cpu_body(const PermArgs& a, int bx, int b) {
  constexpr int J0 = cpu_group_lo(G), J1 = cpu_group_hi(G), NF = 2 * (J1 - J0);
  const size_t h = (size_t)1 << a.logh;
  const size_t r = (size_t)bx * kMT + threadIdx.x;
  if (r >= h) return;
  const Fp4 gamma = m_load_fp4(a.bus_ch + (size_t)b * 8);
  const uint32_t* bpow = a.bpow + (size_t)b * (kInterMaxElems + 1) * 4;
  uint32_t* p = a.perm + (size_t)b * a.perm_bstride + r;
  Fp4 f[NF];
  Fp m[NF];
  cpu_bus_pairs<J0, J1>(a.main_.p + (size_t)b * a.main_.bstride + r, h, gamma, bpow,
                        [&](int j, Fp ma, const Fp4& fa, Fp mb, const Fp4& fb) {
                          f[2 * (j - J0)] = fa; f[2 * (j - J0) + 1] = fb;
                          m[2 * (j - J0)] = ma; m[2 * (j - J0) + 1] = mb;
                        });
  batch_inverse<NF>(f);  // one inversion for the group's fractions
  uint32_t* rs = a.rowsum + ((size_t)b * h + r) * 4;
  Fp4 tot = G == 0 ? Fp4::zero() : m_load_fp4(rs);
#pragma unroll
  for (int j = J0; j < J1; ++j) {
    const Fp4 hj = f[2 * (j - J0)] * m[2 * (j - J0)] + f[2 * (j - J0) + 1] * m[2 * (j - J0) + 1];
    if (j < kCpuHelpers) {
#pragma unroll
      for (int t = 0; t < 4; ++t) p[(size_t)(4 * j + t) * h] = hj.c[t].v;
    }
    tot += hj;
  }
  m_store_fp4(rs, tot);
}

__device__ __forceinline__ void perm_terms_body(const PermArgs& a, int bx, int b) {
  const size_t h = (size_t)1 << a.logh;
  const size_t r = (size_t)bx * kMT + threadIdx.x;
  if (r >= h) return;
  RowView rv{a.prep.width ? a.prep.p + (size_t)b * a.prep.bstride + r : nullptr, a.main_.p + (size_t)b * a.main_.bstride + r,
             a.prep.width, h};
  const Fp4 gamma = m_load_fp4(a.bus_ch + (size_t)b * 8);
  const uint32_t* bpow = a.bpow + (size_t)b * (kInterMaxElems + 1) * 4;
  const int ns = (a.n_inter + 1) / 2;  // slots (no merged interactions outside the CPU chip); the last has no column
  uint32_t* p = a.perm + (size_t)b * a.perm_bstride + r;
  Fp4 tot = Fp4::zero();
  for (int j = 0; j < ns; ++j) {
    Fp4 hj = Fp4::zero();
    for (int k = 2 * j; k < 2 * j + 2 && k < a.n_inter; ++k) {
      const Interaction& it = a.inter[k];
      Fp m = m_lf_eval(it.mult, rv);
      if (m.v == 0) continue;
      if (it.sign < 0) m = -m;
      hj += m_fingerprint(it, rv, gamma, bpow).inv() * m;
    }
    if (j < ns - 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) p[(size_t)(4 * j + t) * h] = hj.c[t].v;
    }
    tot += hj;
  }
  m_store_fp4(a.rowsum + ((size_t)b * h + r) * 4, tot);
}

// The same over (row, group of slots): 32 rows x 8 groups per workgroup, group g takes the slots g, g + 8, ...; the row
// sum is added up through LDS (exact additions, the order does not matter).  For a chip of many interactions and few rows
// in a small batch - the keccak chip's 25 slots over 2 048 rows ran 450 us on eight workgroups.
__global__ __launch_bounds__(kMT) void perm_terms_split_kernel(PermArgs a) {
  constexpr int kRows = 32, kGroups = kMT / kRows;
  __shared__ uint32_t part[kGroups][kRows][4];
  const size_t h = (size_t)1 << a.logh;
  const int lr = threadIdx.x % kRows, g = threadIdx.x / kRows;
  const size_t r = (size_t)blockIdx.x * kRows + lr;
  const bool in_range = r < h;
  const size_t rr = in_range ? r : 0;
  const int b = blockIdx.y;
  RowView rv{a.prep.width ? a.prep.p + (size_t)b * a.prep.bstride + rr : nullptr, a.main_.p + (size_t)b * a.main_.bstride + rr,
             a.prep.width, h};
  const Fp4 gamma = m_load_fp4(a.bus_ch + (size_t)b * 8);
  const uint32_t* bpow = a.bpow + (size_t)b * (kInterMaxElems + 1) * 4;
  const int ns = (a.n_inter + 1) / 2;
  uint32_t* p = a.perm + (size_t)b * a.perm_bstride + rr;
  Fp4 tot = Fp4::zero();
  for (int j = g; j < ns && in_range; j += kGroups) {
    Fp4 hj = Fp4::zero();
    for (int k = 2 * j; k < 2 * j + 2 && k < a.n_inter; ++k) {
      const Interaction& it = a.inter[k];
      Fp m = m_lf_eval(it.mult, rv);
      if (m.v == 0) continue;
      if (it.sign < 0) m = -m;
      hj += m_fingerprint(it, rv, gamma, bpow).inv() * m;
    }
    if (j < ns - 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) p[(size_t)(4 * j + t) * h] = hj.c[t].v;
    }
    tot += hj;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) part[g][lr][t] = tot.c[t].v;
  __syncthreads();
  if (g == 0 && in_range) {
    for (int o = 1; o < kGroups; ++o) {
      Fp4 v;
#pragma unroll
      for (int t = 0; t < 4; ++t) v.c[t] = Fp::raw(part[o][lr][t]);
      tot += v;
    }
    m_store_fp4(a.rowsum + ((size_t)b * h + r) * 4, tot);
  }
}

// one workgroup per proof: total of the row sums -> cum; phi_0 = 0, phi_{r+1} = phi_r + rowsum_r - cum / H
__device__ __forceinline__ void perm_scan_body(const PermArgs& a, int b) {
  __shared__ Fp4 part[kMT];
  const size_t h = (size_t)1 << a.logh;
  const int tid = threadIdx.x;
  const size_t chunk = (h + kMT - 1) / kMT;
  const size_t r0 = (size_t)tid * chunk, r1 = r0 + chunk < h ? r0 + chunk : h;
  const uint32_t* tm = a.rowsum + (size_t)b * h * 4;
  Fp4 local = Fp4::zero();
  for (size_t r = r0; r < r1; ++r) local += m_load_fp4(tm + r * 4);
  part[tid] = local;
  __syncthreads();
  for (int off = 1; off < kMT; off <<= 1) {
    Fp4 v = part[tid];
    if (tid >= off) v += part[tid - off];
    __syncthreads();
    part[tid] = v;
    __syncthreads();
  }
  const Fp4 step = part[kMT - 1] * Fp::raw(a.h_inv);
  Fp4 acc = (tid ? part[tid - 1] : Fp4::zero()) - step * Fp::from_canonical((uint32_t)r0);
  uint32_t* ph = a.perm + (size_t)b * a.perm_bstride + (size_t)(a.perm_width - 4) * h;
  for (size_t r = r0; r < r1; ++r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ph[(size_t)j * h + r] = acc.c[j].v;
    acc += m_load_fp4(tm + r * 4) - step;
  }
  if (tid == kMT - 1) m_store_fp4(a.cum + (size_t)b * a.cum_bstride, part[kMT - 1]);
}

__device__ __forceinline__ Fp4 m_block_sum_fwd(Fp4 v, Fp4* red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Fp4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.c[j] = Fp::raw(__shfl_down(v.c[j].v, off, 64));
    v += o;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  Fp4 r = red[0];
  for (int w = 1; w < kMT / 64; ++w) r += red[w];
  return r;
}

// Tall chips: the running sum in slices.  Pass 1 sums every slice of kScanSlice rows (one workgroup each);
// pass 2 gives a workgroup the total of the slices before its own (and of all, for the step cum / H) and scans the slice.  Field addition is
// exact and associative, so the columns equal the single-workgroup scan's.
constexpr int kScanSlice = 4096;
__device__ __forceinline__ void perm_slice_sum_body(const PermArgs& a, uint32_t* __restrict__ slice_sums, int nslices, int g, int b) {
  __shared__ Fp4 red[kMT / 64];
  const size_t h = (size_t)1 << a.logh;
  const uint32_t* tm = a.rowsum + ((size_t)b * h + (size_t)g * kScanSlice) * 4;
  Fp4 local = Fp4::zero();
  for (int r = threadIdx.x; r < kScanSlice; r += kMT) local += m_load_fp4(tm + (size_t)r * 4);
  const Fp4 tot = m_block_sum_fwd(local, red);
  if (threadIdx.x == 0) m_store_fp4(slice_sums + ((size_t)b * nslices + g) * 4, tot);
}
__device__ __forceinline__ void perm_slice_scan_body(const PermArgs& a, const uint32_t* __restrict__ slice_sums, int nslices, int g, int b) {
  __shared__ Fp4 part[kMT];
  const size_t h = (size_t)1 << a.logh;
  const int tid = threadIdx.x;
  // offset = total of the slices before this one, total = of all of them (at most 512: one strided pass + a block sum each)
  Fp4 before = Fp4::zero(), all = Fp4::zero();
  for (int i = tid; i < nslices; i += kMT) {
    const Fp4 v = m_load_fp4(slice_sums + ((size_t)b * nslices + i) * 4);
    all += v;
    if (i < g) before += v;
  }
  part[tid] = before;
  __syncthreads();
  for (int off = kMT / 2; off >= 1; off >>= 1) {
    if (tid < off) part[tid] += part[tid + off];
    __syncthreads();
  }
  const Fp4 offset = part[0];
  __syncthreads();
  part[tid] = all;
  __syncthreads();
  for (int off = kMT / 2; off >= 1; off >>= 1) {
    if (tid < off) part[tid] += part[tid + off];
    __syncthreads();
  }
  const Fp4 total = part[0], step = total * Fp::raw(a.h_inv);
  __syncthreads();
  // phi[r] = sum over r' < r of (rowsum[r'] - step).  The slice is walked in rounds of kMT consecutive rows, lane = row,
  // so that the row sums are read and the four phi columns written as consecutive words (a lane that owns sixteen
  // consecutive rows writes single words 64 bytes apart); inside a round: an inclusive scan per wave with lane shuffles,
  // the waves' totals through LDS.
  __shared__ Fp4 wtot[kMT / 64];
  const int lane = tid & 63, wave = tid >> 6;
  const size_t r_base = (size_t)g * kScanSlice;
  const uint32_t* tm = a.rowsum + (size_t)b * h * 4;
  uint32_t* ph = a.perm + (size_t)b * a.perm_bstride + (size_t)(a.perm_width - 4) * h;
  Fp4 carry = offset - step * Fp::from_canonical((uint32_t)r_base) - step * Fp::from_canonical((uint32_t)tid);
  const Fp4 round_step = step * Fp::from_canonical((uint32_t)kMT);
  for (int i = 0; i < kScanSlice / kMT; ++i) {
    const size_t r = r_base + (size_t)i * kMT + tid;
    const Fp4 v = m_load_fp4(tm + r * 4);
    Fp4 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      Fp4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o.c[j] = Fp::raw((uint32_t)__shfl_up((int)incl.c[j].v, off, 64));
      if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    Fp4 before = Fp4::zero(), blk = Fp4::zero();
#pragma unroll
    for (int w = 0; w < kMT / 64; ++w) {
      const Fp4 t = wtot[w];
      if (w < wave) before += t;
      blk += t;
    }
    __syncthreads();
    const Fp4 phi = carry + before + (incl - v);
#pragma unroll
    for (int j = 0; j < 4; ++j) ph[(size_t)j * h + r] = phi.c[j].v;
    carry += blk - round_step;
  }
  if (g == 0 && tid == 0) m_store_fp4(a.cum + (size_t)b * a.cum_bstride, total);
}

template <int G>
__global__ __launch_bounds__(kMT) void perm_terms_cpu_kernel(PermArgs a) { perm_terms_cpu_body<G>(a, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(kMT) void perm_terms_kernel(PermArgs a) { perm_terms_body(a, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(kMT) void perm_scan_kernel(PermArgs a) { perm_scan_body(a, blockIdx.x); }
__global__ __launch_bounds__(kMT) void perm_slice_sum_kernel(PermArgs a, uint32_t* __restrict__ slice_sums, int nslices) {
  perm_slice_sum_body(a, slice_sums, nslices, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(kMT) void perm_slice_scan_kernel(PermArgs a, const uint32_t* __restrict__ slice_sums, int nslices) {
  perm_slice_scan_body(a, slice_sums, nslices, blockIdx.x, blockIdx.y);
}

// ---- a small batch's LogUp stage: one launch per kind of kernel over device tables of PermArgs (PermArgs::blk0 = the
// task's first workgroup in that launch), instead of three launches per chip on three lanes (90 launches, 0.57 ms of a
// single proof).  The bodies are the single-chip kernels', so the columns are the same words. ----
__device__ __forceinline__ const PermArgs& perm_task_of(const PermArgs* __restrict__ tasks, int n, int blk) {
  int t = 0;
  while (t + 1 < n && blk >= tasks[t + 1].blk0) ++t;
  return tasks[t];
}
template <int G>
__global__ __launch_bounds__(kMT) void perm_terms_cpu_multi_kernel(const PermArgs* __restrict__ tasks, int n) {
  const PermArgs& a = perm_task_of(tasks, n, blockIdx.x);
  const int local = blockIdx.x - a.blk0, nb = (int)((((size_t)1 << a.logh) + kMT - 1) / kMT);
  perm_terms_cpu_body<G>(a, local % nb, local / nb);
}
__global__ __launch_bounds__(kMT) void perm_terms_multi_kernel(const PermArgs* __restrict__ tasks, int n) {
  const PermArgs& a = perm_task_of(tasks, n, blockIdx.x);
  const int local = blockIdx.x - a.blk0, nb = (int)((((size_t)1 << a.logh) + kMT - 1) / kMT);
  perm_terms_body(a, local % nb, local / nb);
}
__global__ __launch_bounds__(kMT) void perm_scan_multi_kernel(const PermArgs* __restrict__ tasks, int n) {
  const PermArgs& a = perm_task_of(tasks, n, blockIdx.x);
  perm_scan_body(a, blockIdx.x - a.blk0);
}
__global__ __launch_bounds__(kMT) void perm_slice_sum_multi_kernel(const PermArgs* __restrict__ tasks, int n) {
  const PermArgs& a = perm_task_of(tasks, n, blockIdx.x);
  const int local = blockIdx.x - a.blk0, nslices = (int)(((size_t)1 << a.logh) / kScanSlice);
  perm_slice_sum_body(a, a.slice_sums, nslices, local % nslices, local / nslices);
}
__global__ __launch_bounds__(kMT) void perm_slice_scan_multi_kernel(const PermArgs* __restrict__ tasks, int n) {
  const PermArgs& a = perm_task_of(tasks, n, blockIdx.x);
  const int local = blockIdx.x - a.blk0, nslices = (int)(((size_t)1 << a.logh) / kScanSlice);
  perm_slice_scan_body(a, a.slice_sums, nslices, local % nslices, local / nslices);
}
// which launches a chip's LogUp trace takes (launch_perm_trace's choices): bit 0 the CPU kernels, bit 1 the generic terms,
// bit 2 the split terms (its own launch), bit 3 the sliced scan, bit 4 the single-workgroup scan
int perm_task_kinds(const PermArgs& a) {
  const size_t h = (size_t)1 << a.logh;
  int k = is_cpu_chip(a.chip) ? 1 : (a.n_inter >= 16 && h * (size_t)a.batch <= 32768) ? 4 : 2;
  k |= (h >= (size_t)4 * kScanSlice && a.slice_sums) ? 8 : 16;
  return k;
}
int perm_task_blocks(const PermArgs& a, int kind_bit) {
  const size_t h = (size_t)1 << a.logh;
  if (kind_bit == 1 || kind_bit == 2) return (int)((h + kMT - 1) / kMT) * a.batch;
  if (kind_bit == 8) return (int)(h / kScanSlice) * a.batch;
  return a.batch;
}
// (the three kinds of term kernels are independent of one another - the caller may put them on three streams - and the
// scans follow all of them)
void launch_perm_multi_cpu_terms(hipStream_t stream, const PermMulti& m) {
  if (!m.n[0]) return;
  hipLaunchKernelGGL(perm_terms_cpu_multi_kernel<0>, dim3(m.blocks[0]), dim3(kMT), 0, stream, m.tasks[0], m.n[0]);
  hipLaunchKernelGGL(perm_terms_cpu_multi_kernel<1>, dim3(m.blocks[0]), dim3(kMT), 0, stream, m.tasks[0], m.n[0]);
}
void launch_perm_multi_terms(hipStream_t stream, const PermMulti& m) {
  if (m.n[1]) hipLaunchKernelGGL(perm_terms_multi_kernel, dim3(m.blocks[1]), dim3(kMT), 0, stream, m.tasks[1], m.n[1]);
}
void launch_perm_multi_scans(hipStream_t stream, const PermMulti& m) {
  if (m.n[2]) {
    hipLaunchKernelGGL(perm_slice_sum_multi_kernel, dim3(m.blocks[2]), dim3(kMT), 0, stream, m.tasks[2], m.n[2]);
    hipLaunchKernelGGL(perm_slice_scan_multi_kernel, dim3(m.blocks[2]), dim3(kMT), 0, stream, m.tasks[2], m.n[2]);
  }
  if (m.n[3]) hipLaunchKernelGGL(perm_scan_multi_kernel, dim3(m.blocks[3]), dim3(kMT), 0, stream, m.tasks[3], m.n[3]);
}
void launch_perm_terms_split(hipStream_t stream, const PermArgs& a) {
  const size_t h = (size_t)1 << a.logh;
  hipLaunchKernelGGL(perm_terms_split_kernel, dim3((unsigned)((h + 31) / 32), a.batch), dim3(kMT), 0, stream, a);
}

void launch_perm_trace(hipStream_t stream, const PermArgs& a) {
  const size_t h = (size_t)1 << a.logh;
  if (is_cpu_chip(a.chip)) {
    const dim3 grid((unsigned)((h + kMT - 1) / kMT), a.batch);
    hipLaunchKernelGGL(perm_terms_cpu_kernel<0>, grid, dim3(kMT), 0, stream, a);
    hipLaunchKernelGGL(perm_terms_cpu_kernel<1>, grid, dim3(kMT), 0, stream, a);
    static_assert(kCpuBusGroups == 2, "one launch per group of slots");
  } else if (a.n_inter >= 16 && h * (size_t)a.batch <= 32768)
    hipLaunchKernelGGL(perm_terms_split_kernel, dim3((unsigned)((h + 31) / 32), a.batch), dim3(kMT), 0, stream, a);
  else
    hipLaunchKernelGGL(perm_terms_kernel, dim3((unsigned)((h + kMT - 1) / kMT), a.batch), dim3(kMT), 0, stream, a);
  if (h >= (size_t)4 * kScanSlice && a.slice_sums) {
    const int nslices = (int)(h / kScanSlice);
    hipLaunchKernelGGL(perm_slice_sum_kernel, dim3(nslices, a.batch), dim3(kMT), 0, stream, a, a.slice_sums, nslices);
    hipLaunchKernelGGL(perm_slice_scan_kernel, dim3(nslices, a.batch), dim3(kMT), 0, stream, a, a.slice_sums, nslices);
  } else {
    hipLaunchKernelGGL(perm_scan_kernel, dim3(a.batch), dim3(kMT), 0, stream, a);
  }
}

__global__ __launch_bounds__(64) void public_bus_kernel(const uint32_t* __restrict__ pub, const uint32_t* __restrict__ bus_ch,
                                                       const uint32_t* __restrict__ bpow_all, uint32_t* __restrict__ out,
                                                       size_t out_bstride, int batch) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= batch) return;
  const Fp4 gamma = m_load_fp4(bus_ch + (size_t)b * 8);
  const uint32_t* bp = bpow_all + (size_t)b * (kInterMaxElems + 1) * 4;
  const Fp4 b1 = m_load_fp4(bp + 4), b2 = m_load_fp4(bp + 8), b3 = m_load_fp4(bp + 12), b4 = m_load_fp4(bp + 16);
  const uint32_t* w = pub + (size_t)b * kPubWords;
  Fp4 total = Fp4::zero();
  for (uint32_t kind = 1; kind <= 2; ++kind)
    for (uint32_t i = 0; i < 8; ++i) {
      const uint32_t v = w[(kind - 1) * 8 + i];
      Fp4 f = gamma;
      f.c[0] += Fp::from_canonical((uint32_t)BUS_PUBC);
      f += b1 * Fp::from_canonical(kind) + b2 * Fp::from_canonical(i) + b3 * Fp::from_canonical(v & 0xffff) +
           b4 * Fp::from_canonical(v >> 16);
      total -= f.inv();
    }
  Fp4 fh = gamma;
  fh.c[0] += Fp::from_canonical((uint32_t)BUS_PUBH);
  fh += b1 * Fp::from_canonical(w[16] & 0xffff) + b2 * Fp::from_canonical(w[16] >> 16);
  total -= fh.inv();
  m_store_fp4(out + (size_t)b * out_bstride, total);
}
void launch_public_bus(hipStream_t stream, const uint32_t* pub_words, const uint32_t* bus_ch, const uint32_t* bpow, uint32_t* out,
                       size_t out_bstride, int batch) {
  hipLaunchKernelGGL(public_bus_kernel, dim3((batch + 63) / 64), dim3(64), 0, stream, pub_words, bus_ch, bpow, out, out_bstride, batch);
}

// ===========================================================================================
// quotient evaluation
// ===========================================================================================
struct MQCtx {
  using F = Fp;
  const uint32_t* loc;
  const uint32_t* nxt;
  const uint32_t* prp;  // this point of the preprocessed LDE (same column stride), or null
  size_t cs;
  Fp first, trans, last;
  uint32_t pub_[kNumCpuPub];
  const uint32_t* ap;
  int k_;
  Fp4 acc;
  int64_t lazy[4];
  int pending;
  const P2Consts* p2_;  // Poseidon2 chip: the permutation's constants
  __device__ __forceinline__ const P2Consts* p2() const { return p2_; }
  uint32_t* stash_;    // ALU task 1: this lane's column of a [32][kMT] LDS array
  __device__ __forceinline__ void stash(int i, F v) const { stash_[i * kMT] = v.v; }
  __device__ __forceinline__ F stashed(int i) const { return Fp::raw(stash_[i * kMT]); }
  __device__ __forceinline__ F local(int col) const { return Fp::raw(loc[(size_t)col * cs]); }
  __device__ __forceinline__ F prep(int col) const { return Fp::raw(prp[(size_t)col * cs]); }
  __device__ __forceinline__ F next(int col) const { return Fp::raw(nxt[(size_t)col * cs]); }
  __device__ __forceinline__ F is_first() const { return first; }
  __device__ __forceinline__ F is_trans() const { return trans; }
  __device__ __forceinline__ F is_last() const { return last; }
  __device__ __forceinline__ F pub(int which) const { return Fp::raw(pub_[which]); }
  __device__ __forceinline__ F one() const { return Fp::one(); }
  __device__ __forceinline__ F k(uint32_t monty) const { return Fp::raw(monty); }
  // acc += alpha^k * v through signed 64-bit lazy sums (field.hpp): both factors centred, so a product is one
  // v_mad_i64_i32 and eight of them (< 2^62.8) fit between two range reductions
  __device__ __forceinline__ void emit_at(int idx, F v) {
    const uint32_t* p = ap + 4 * (size_t)idx;
    const int32_t vc = fps_centre(v.v);
#pragma unroll
    for (int i = 0; i < 4; ++i) lazy[i] += (int64_t)fps_centre(p[i]) * (int64_t)vc;
    if (++pending == 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) lazy[i] = (int64_t)fps_fold(lazy[i]) * (int64_t)kRModP;
      pending = 0;
    }
  }
  __device__ __forceinline__ void emit(F v) { emit_at(k_++, v); }
  __device__ __forceinline__ void set_count(int n) { k_ = n; }
  __device__ __forceinline__ void flush() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc.c[i] = acc.c[i] + Fp::raw(fps_canon(fps_fold(lazy[i])));
      lazy[i] = 0;
    }
    pending = 0;
  }
};

struct PointInfo {
  int b, c;
  size_t m, mn, pt;
  Fp first, trans, last;
};
// domain point (coset c, index m) of an LDE of height H and the three selectors there
__device__ __forceinline__ void point_selectors(const MQuotArgs& a, size_t pt, PointInfo* pi) {
  const size_t h = (size_t)1 << a.logh;
  pi->pt = pt;
  pi->c = pt >= h ? 1 : 0;
  pi->m = pt - (size_t)pi->c * h;
  pi->mn = (pi->m + 1) & (h - 1);
  const size_t half = h >> 1;
  const Fp wm = pi->m < half ? Fp::raw(a.tw_fwd[pi->m]) : -Fp::raw(a.tw_fwd[pi->m - half]);
  const Fp x = Fp::raw(pi->c ? a.shift[1] : a.shift[0]) * wm;  // (no dynamic index into kernel arguments: that would spill them)
  const Fp zh = Fp::raw(pi->c ? a.zh_inv[1] : a.zh_inv[0]).inv();
  const Fp whi = Fp::raw(a.wh_inv);
  pi->trans = x - whi;
  pi->first = zh * (x - Fp::one()).inv();
  pi->last = zh * pi->trans.inv();
}

// the LogUp constraints of a chip at one point (one per slot: machine_defs.hpp "LogUp layout"), folded with their powers
// of alpha into `acc`.  No merged interactions here: the CPU chip has its own kernels.
__device__ __forceinline__ void logup_constraints(const MQuotArgs& a, const PointInfo& pi, Fp4* acc) {
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  const int b = pi.b;
  RowView rv{a.prep.width ? a.prep.p + (size_t)b * a.prep.bstride + pi.pt : nullptr,
             a.main_.p + (size_t)b * a.main_.bstride + pi.pt, a.prep.width, n};
  const Fp4 gamma = m_load_fp4(a.bus_ch + (size_t)b * 8);
  const uint32_t* bpow = a.bpow + (size_t)b * (kInterMaxElems + 1) * 4;
  const uint32_t* ap = a.alpha_pows + (size_t)b * a.alpha_bstride;
  const int ns = (a.n_inter + 1) / 2, nh = ns - 1;
  const uint32_t* pl = a.perm.p + (size_t)b * a.perm.bstride + (size_t)pi.c * h;
  Fp4 hsum = Fp4::zero();
  for (int j = 0; j < ns; ++j) {
    Fp4 hj;
    if (j < nh) {
#pragma unroll
      for (int t = 0; t < 4; ++t) hj.c[t] = Fp::raw(pl[(size_t)(4 * j + t) * n + pi.m]);
      hsum += hj;
    } else {  // the last slot's value: phi_next - phi + cum / H - the helper columns
      Fp4 phi, phin;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        phi.c[t] = Fp::raw(pl[(size_t)(4 * nh + t) * n + pi.m]);
        phin.c[t] = Fp::raw(pl[(size_t)(4 * nh + t) * n + pi.mn]);
      }
      hj = phin - phi + m_load_fp4(a.cum + (size_t)b * a.cum_bstride) * Fp::raw(a.h_inv) - hsum;
    }
    const Interaction& ia = a.inter[2 * j];
    Fp ma = m_lf_eval(ia.mult, rv);
    if (ia.sign < 0) ma = -ma;
    const Fp4 fa = m_fingerprint(ia, rv, gamma, bpow);
    Fp4 v;
    if (2 * j + 1 < a.n_inter) {
      const Interaction& ib = a.inter[2 * j + 1];
      Fp mb = m_lf_eval(ib.mult, rv);
      if (ib.sign < 0) mb = -mb;
      const Fp4 fb = m_fingerprint(ib, rv, gamma, bpow);
      v = hj * fa * fb - (fb * ma + fa * mb);
    } else {
      v = hj * fa;
      v.c[0] -= ma;
    }
    *acc += m_load_fp4(ap + 4 * (size_t)(a.n_base + j)) * v;
  }
}

__device__ __forceinline__ void init_ctx(const MQuotArgs& a, const PointInfo& pi, MQCtx* ctx) {
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  const uint32_t* base = a.main_.p + (size_t)pi.b * a.main_.bstride + (size_t)pi.c * h;
  ctx->loc = base + pi.m;
  ctx->nxt = base + pi.mn;
  ctx->cs = n;
  ctx->first = pi.first;
  ctx->trans = pi.trans;
  ctx->last = pi.last;
  ctx->prp = a.prep.width ? a.prep.p + (size_t)pi.b * a.prep.bstride + (size_t)pi.c * h + pi.m : nullptr;
#pragma unroll
  for (int i = 0; i < kNumCpuPub; ++i) ctx->pub_[i] = a.pubs ? a.pubs[(size_t)pi.b * a.pubs_bstride + i] : 0u;
  ctx->ap = a.alpha_pows + (size_t)pi.b * a.alpha_bstride;
  ctx->k_ = 0;
  ctx->acc = Fp4::zero();
  ctx->lazy[0] = ctx->lazy[1] = ctx->lazy[2] = ctx->lazy[3] = 0;
  ctx->pending = 0;
  ctx->stash_ = nullptr;
  ctx->p2_ = a.consts;
}

template <int CHIP>
__global__ __launch_bounds__(kMT) void machine_quotient_kernel(MQuotArgs a) {
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  const size_t pt = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (pt >= n) return;
  PointInfo pi;
  pi.b = blockIdx.y;
  point_selectors(a, pt, &pi);
  MQCtx ctx;
  init_ctx(a, pi, &ctx);
  __shared__ uint32_t stash[is_alu_chip(CHIP) ? 32 * kMT : 1];  // the ALU chip parks B's 32 bits per lane (MQCtx::stash)
  ctx.stash_ = stash + (is_alu_chip(CHIP) ? threadIdx.x : 0);
  if constexpr (CHIP == kKmem) eval_kmem(ctx);
  else if constexpr (CHIP == kMemFinal) eval_memfinal(ctx);
  else if constexpr (CHIP == kImage) eval_image(ctx);
  else if constexpr (CHIP == kMul) eval_mul(ctx);
  else if constexpr (CHIP == kTable) eval_table(ctx);
  else if constexpr (is_alu_chip(CHIP)) eval_alu(ctx);
  else if constexpr (is_sub_chip(CHIP)) eval_sub(ctx);
  else if constexpr (is_bw_chip(CHIP)) eval_bw(ctx);
  else if constexpr (CHIP == kP2) eval_p2(ctx);
  else if constexpr (CHIP == kEcall) eval_ecall(ctx);
  else if constexpr (CHIP == kQr) eval_qr(ctx);
  else if constexpr (CHIP == kTr) eval_tr(ctx);
  else if constexpr (CHIP == kHint) eval_hint(ctx);
  else if constexpr (CHIP == kDiv) eval_div(ctx);
  ctx.flush();
  logup_constraints(a, pi, &ctx.acc);
  const Fp4 q = ctx.acc * Fp::raw(pi.c ? a.zh_inv[1] : a.zh_inv[0]);
  uint32_t* dst = a.quot + (size_t)pi.b * 8 * h + pi.m;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t* qd = dst + (size_t)(4 * pi.c + j) * h;
    *qd = a.accumulate ? (q.c[j] + Fp::raw(*qd)).v : q.c[j].v;
  }
}

// CPU chip: the base constraints (task 0) and the LogUp constraints with the fingerprints of cpu_bus_pairs, one launch
// per group of slots (tasks 1..2), each with its own register budget.  A point's partial sum travels through
// a.partial ([B][2H] Fp4): task 0 writes it, the others add, the last one divides by the vanishing polynomial.
constexpr int kCpuQuotSplit = kCpuSlots;  // slots [0, split) in task 1, [split, 8) in task 2 (none: task 1 is the last)
template <int TASK>
__global__ __launch_bounds__(kMT) __attribute__((amdgpu_waves_per_eu(4))) void cpu_quotient_task_kernel(MQuotArgs a) {
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  const size_t pt = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (pt >= n) return;
  PointInfo pi;
  pi.b = blockIdx.y;
  point_selectors(a, pt, &pi);
  uint32_t* part = a.partial + ((size_t)pi.b * n + pt) * 4;
  // TASK 3: both in one launch - the base constraints' sum stays in registers and the row the LogUp terms read again is
  // still in the cache (the two phases' registers do not add up: only the sum lives across them)
  Fp4 base_acc = Fp4::zero();
  if constexpr (TASK == 0 || TASK == 3) {
    MQCtx ctx;
    init_ctx(a, pi, &ctx);
    eval_cpu(ctx);
    ctx.flush();
    if constexpr (TASK == 0) m_store_fp4(part, ctx.acc);
    base_acc = ctx.acc;
  }
  if constexpr (TASK != 0) {
    constexpr int J0 = TASK == 2 ? kCpuQuotSplit : 0, J1 = TASK == 1 ? kCpuQuotSplit : kCpuSlots, nh = kCpuHelpers;
    const int b = pi.b;
    const Fp4 gamma = m_load_fp4(a.bus_ch + (size_t)b * 8);
    const uint32_t* bpow = a.bpow + (size_t)b * (kInterMaxElems + 1) * 4;
    const uint32_t* ap = a.alpha_pows + (size_t)b * a.alpha_bstride;
    const uint32_t* pl = a.perm.p + (size_t)b * a.perm.bstride + (size_t)pi.c * h;
    Fp4 acc = TASK == 3 ? base_acc : m_load_fp4(part);
    cpu_bus_pairs<J0, J1>(a.main_.p + (size_t)b * a.main_.bstride + pt, n, gamma, bpow,
                          [&](int j, Fp ma, const Fp4& fa, Fp mb, const Fp4& fb) {
                            Fp4 hj;
                            if (j < nh) {
#pragma unroll
                              for (int t = 0; t < 4; ++t) hj.c[t] = Fp::raw(pl[(size_t)(4 * j + t) * n + pi.m]);
                            } else {
                              // the last slot has no column: its value is phi_next - phi + cum / H - the helper columns
#pragma unroll
                              for (int t = 0; t < 4; ++t) {
                                Fp sacc = Fp::zero();
                                for (int i = 0; i < nh; ++i) sacc += Fp::raw(pl[(size_t)(4 * i + t) * n + pi.m]);
                                hj.c[t] = Fp::raw(pl[(size_t)(4 * nh + t) * n + pi.mn]) - Fp::raw(pl[(size_t)(4 * nh + t) * n + pi.m]) - sacc;
                              }
                              hj += m_load_fp4(a.cum + (size_t)b * a.cum_bstride) * Fp::raw(a.h_inv);
                            }
                            acc += m_load_fp4(ap + 4 * (size_t)(a.n_base + j)) * (hj * fa * fb - (fb * ma + fa * mb));
                          });
    if constexpr (J1 == kCpuSlots) {
      const Fp4 q = acc * Fp::raw(pi.c ? a.zh_inv[1] : a.zh_inv[0]);
      uint32_t* dst = a.quot + (size_t)b * 8 * h + pi.m;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
    uint32_t* qd = dst + (size_t)(4 * pi.c + j) * h;
    *qd = a.accumulate ? (q.c[j] + Fp::raw(*qd)).v : q.c[j].v;
  }
    } else {
      m_store_fp4(part, acc);
    }
  }
}

// keccak chip: p3-keccak-air's 12 evaluation tasks (air_keccak.hpp) plus one task for the call-time
// constraint and the LogUp constraints; XCD-aware tile order as in keccak_quotient_kernel
__global__ __launch_bounds__(kMT) void keccak_machine_quotient_kernel(MQuotArgs a, int tiles_per_proof, int total_tiles) {
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  int tile, g;
  if ((total_tiles & 7) == 0) {
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    tile = (seq / ka::kNumTasks) * 8 + xcd;
    g = seq % ka::kNumTasks;
  } else {
    tile = blockIdx.x / ka::kNumTasks;
    g = blockIdx.x % ka::kNumTasks;
  }
  const int b = tile / tiles_per_proof;
  const size_t pt = (size_t)(tile - b * tiles_per_proof) * kMT + threadIdx.x;
  if (pt >= n) return;
  PointInfo pi;
  pi.b = b;
  point_selectors(a, pt, &pi);
  MQCtx ctx;
  init_ctx(a, pi, &ctx);
  if (g == ka::kBusTask) {
    ctx.k_ = ka::kNumConstraints;
    eval_keccak_ts(ctx);
    ctx.flush();
    logup_constraints(a, pi, &ctx.acc);
  } else {
    ka::eval_task(g, ctx);
    ctx.flush();
  }
  m_store_fp4(a.partial + (((size_t)b * ka::kNumTasks + g) * n + pt) * 4, ctx.acc);
}
__global__ __launch_bounds__(kMT) void keccak_machine_combine_kernel(MQuotArgs a) {
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  const size_t pt = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (pt >= n) return;
  const int b = blockIdx.y;
  const int c = pt >= h ? 1 : 0;
  const size_t m = pt - (size_t)c * h;
  Fp4 acc = Fp4::zero();
  for (int g = 0; g < ka::kNumTasks; ++g) acc += m_load_fp4(a.partial + (((size_t)b * ka::kNumTasks + g) * n + pt) * 4);
  acc = acc * Fp::raw(c ? a.zh_inv[1] : a.zh_inv[0]);
  uint32_t* q = a.quot + (size_t)b * 8 * h + m;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t* qd = q + (size_t)(4 * c + j) * h;
    *qd = a.accumulate ? (acc.c[j] + Fp::raw(*qd)).v : acc.c[j].v;
  }
}

void launch_machine_quotient(hipStream_t stream, const MQuotArgs& a) {
  const size_t n = (size_t)2 << a.logh;
  const dim3 grid((unsigned)((n + kMT - 1) / kMT), a.batch), block(kMT);
  switch (a.chip) {
    case kCpu:
    case kCpu2: case kCpu3: case kCpu4: case kCpu5: case kCpu6: case kCpu7: case kCpu8:
    {
      static const bool fused = getenv("ZKSP_CPU_QUOT_FUSED") != nullptr;  // measurement switch
      if (fused && kCpuQuotSplit == kCpuSlots) {
        hipLaunchKernelGGL(cpu_quotient_task_kernel<3>, grid, block, 0, stream, a);
        break;
      }
      hipLaunchKernelGGL(cpu_quotient_task_kernel<0>, grid, block, 0, stream, a);
      hipLaunchKernelGGL(cpu_quotient_task_kernel<1>, grid, block, 0, stream, a);
      if (kCpuQuotSplit < kCpuSlots) hipLaunchKernelGGL(cpu_quotient_task_kernel<2>, grid, block, 0, stream, a);
      break;
    }
    case kAlu:
    case kAlu2: hipLaunchKernelGGL(machine_quotient_kernel<kAlu>, grid, block, 0, stream, a); break;
    case kSub:
    case kSub2: hipLaunchKernelGGL(machine_quotient_kernel<kSub>, grid, block, 0, stream, a); break;
    case kBw:
    case kBw2: hipLaunchKernelGGL(machine_quotient_kernel<kBw>, grid, block, 0, stream, a); break;
    case kP2: hipLaunchKernelGGL(machine_quotient_kernel<kP2>, grid, block, 0, stream, a); break;
    case kEcall: hipLaunchKernelGGL(machine_quotient_kernel<kEcall>, grid, block, 0, stream, a); break;
    case kQr: hipLaunchKernelGGL(machine_quotient_kernel<kQr>, grid, block, 0, stream, a); break;
    case kTr: hipLaunchKernelGGL(machine_quotient_kernel<kTr>, grid, block, 0, stream, a); break;
    case kHint: hipLaunchKernelGGL(machine_quotient_kernel<kHint>, grid, block, 0, stream, a); break;
    case kDiv: hipLaunchKernelGGL(machine_quotient_kernel<kDiv>, grid, block, 0, stream, a); break;
    case kKmem: hipLaunchKernelGGL(machine_quotient_kernel<kKmem>, grid, block, 0, stream, a); break;
    case kMemFinal: hipLaunchKernelGGL(machine_quotient_kernel<kMemFinal>, grid, block, 0, stream, a); break;
    case kImage: hipLaunchKernelGGL(machine_quotient_kernel<kImage>, grid, block, 0, stream, a); break;
    case kProgram: hipLaunchKernelGGL(machine_quotient_kernel<kProgram>, grid, block, 0, stream, a); break;
    case kMul: hipLaunchKernelGGL(machine_quotient_kernel<kMul>, grid, block, 0, stream, a); break;
    case kTable: hipLaunchKernelGGL(machine_quotient_kernel<kTable>, grid, block, 0, stream, a); break;
    case kKeccak: {
      const int blocks = (int)((n + kMT - 1) / kMT), total_tiles = blocks * a.batch;
      hipLaunchKernelGGL(keccak_machine_quotient_kernel, dim3((unsigned)total_tiles * ka::kNumTasks), block, 0, stream, a, blocks,
                         total_tiles);
      hipLaunchKernelGGL(keccak_machine_combine_kernel, grid, block, 0, stream, a);
      break;
    }
  }
}

// ===========================================================================================
// reduced openings of one chip
// ===========================================================================================
constexpr int kMReduceChunk = 128;
int mreduce_nchunks(int total_width) { return (total_width + kMReduceChunk - 1) / kMReduceChunk; }

__device__ __forceinline__ Fp4 m_block_sum(Fp4 v, Fp4* red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Fp4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.c[j] = Fp::raw(__shfl_down(v.c[j].v, off, 64));
    v += o;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  Fp4 r = red[0];
  for (int w = 1; w < kMT / 64; ++w) r += red[w];
  return r;
}

__global__ __launch_bounds__(kMT) void mreduce_bsum_kernel(MReduceArgs a, int n1, int n2) {
  __shared__ Fp4 red[kMT / 64];
  const int b = blockIdx.x;
  const uint32_t* ap = a.af_pows + (size_t)b * a.af_bstride + a.pow_off * 4;
  const uint32_t* op = a.opened + (size_t)b * a.opened_bstride + a.open_off * 4;
  Fp4 s1 = Fp4::zero(), s2 = Fp4::zero();
  for (int i = threadIdx.x; i < n1; i += kMT) s1 += m_load_fp4(ap + (size_t)i * 4) * m_load_fp4(op + (size_t)i * 4);
  for (int i = threadIdx.x; i < n2; i += kMT) s2 += m_load_fp4(ap + (size_t)(n1 + i) * 4) * m_load_fp4(op + (size_t)(n1 + i) * 4);
  const Fp4 r1 = m_block_sum(s1, red), r2 = m_block_sum(s2, red);
  if (threadIdx.x == 0) {
    m_store_fp4(a.bsum + ((size_t)b * 2 + 0) * 4, r1);
    m_store_fp4(a.bsum + ((size_t)b * 2 + 1) * 4, r2);
  }
}

__device__ __forceinline__ int64_t m_lazy_shrink(int64_t t) { return (int64_t)fps_fold(t) * (int64_t)kRModP; }

// lane = four consecutive LDE points; blockIdx.y = a chunk of `chunk_cols` columns of [prep | main | perm | quot]
// (kMReduceChunk of them; a batch below eight takes chunks up to eight times shorter - the scratch is sized for eight
// proofs - so that a single proof's 92-column CPU instance is 768 workgroups instead of 128)
__global__ __launch_bounds__(kMT) void mreduce_partial_kernel(MReduceArgs a, int nchunks, int n1, int chunk_cols) {
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  const size_t pt = ((size_t)blockIdx.x * kMT + threadIdx.x) * 4;
  if (pt >= n) return;
  const int chunk = blockIdx.y, b = blockIdx.z;
  const int i0 = chunk * chunk_cols, i1 = min(n1, i0 + chunk_cols);
  const int w0 = a.mats[0].width, w01 = w0 + a.mats[1].width, w012 = w01 + a.mats[2].width;
  const uint32_t* ap = a.af_pows + (size_t)b * a.af_bstride + a.pow_off * 4;
  int64_t acc1[4][4], acc2[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[q][j] = acc2[q][j] = 0;
  // four columns per trip: their loads are issued together, then the 2 x 16 signed 64-bit sums take the four products
  // each; every second trip brings the sums back to range
  for (int i = i0; i < i1; i += 4) {
    uint4 v[4];
    const uint32_t *al[4], *al2[4];
    bool two[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int iu = min(i + u, i1 - 1);
      const int mi = iu < w0 ? 0 : iu < w01 ? 1 : iu < w012 ? 2 : 3;
      const int col = iu - (mi == 0 ? 0 : mi == 1 ? w0 : mi == 2 ? w01 : w012);
      const Seg& sg = a.mats[mi];
      v[u] = *reinterpret_cast<const uint4*>(sg.p + (size_t)b * sg.bstride + (size_t)col * n + pt);
      if (i + u >= i1) v[u] = make_uint4(0u, 0u, 0u, 0u);
      al[u] = ap + (size_t)iu * 4;
      two[u] = mi == 1 || mi == 2;
      al2[u] = ap + (size_t)(n1 + iu - w0) * 4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // both factors centred: a 32 x 32 -> 64-bit signed multiply-add is one instruction (v_mad_i64_i32)
      const int32_t w[4] = {fps_centre(v[u].x), fps_centre(v[u].y), fps_centre(v[u].z), fps_centre(v[u].w)};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int32_t alpha = fps_centre(al[u][j]);
        const int32_t alpha2 = two[u] ? fps_centre(al2[u][j]) : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc1[q][j] += (int64_t)alpha * (int64_t)w[q];
          acc2[q][j] += (int64_t)alpha2 * (int64_t)w[q];
        }
      }
    }
    // |alpha|, |w| <= (p - 1) / 2: eight products and a shrunk sum (< 2^58.2) stay below 2^63
    if ((i - i0) & 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc1[q][j] = m_lazy_shrink(acc1[q][j]); acc2[q][j] = m_lazy_shrink(acc2[q][j]); }
    }
  }
  uint32_t* p1 = a.partial + ((((size_t)b * nchunks + chunk) * 2 + 0) * n + pt) * 4;
  uint32_t* p2 = a.partial + ((((size_t)b * nchunks + chunk) * 2 + 1) * n + pt) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    Fp4 r1, r2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r1.c[j] = Fp::raw(fps_canon(fps_fold(acc1[q][j])));
      r2.c[j] = Fp::raw(fps_canon(fps_fold(acc2[q][j])));
    }
    m_store_fp4(p1 + 4 * q, r1);
    m_store_fp4(p2 + 4 * q, r2);
  }
}

// Lane = four consecutive LDE points: their eight denominators (x - zeta, x - zeta w) are inverted with one
// extension-field inversion (an inversion is ~700 instructions, most of this kernel's work when taken per point)
__global__ __launch_bounds__(kMT) void mreduce_final_kernel(MReduceArgs a, int nchunks) {
  const size_t h = (size_t)1 << a.logh, n = 2 * h;
  const size_t pt0 = ((size_t)blockIdx.x * kMT + threadIdx.x) * 4;
  if (pt0 >= n) return;
  const int b = blockIdx.y;
  const Fp4 b1 = m_load_fp4(a.bsum + ((size_t)b * 2 + 0) * 4), b2 = m_load_fp4(a.bsum + ((size_t)b * 2 + 1) * 4);
  const Fp4 zeta = m_load_fp4(a.zeta + (size_t)b * 4), zn = zeta * Fp::raw(a.w_h);
  const int c = pt0 >= h ? 1 : 0;  // (h is a multiple of four: the four points lie on one coset)
  const size_t half = h >> 1;
  const Fp sh = Fp::raw(c ? a.shift[1] : a.shift[0]);
  Fp4 d[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t m = pt0 + i - (size_t)c * h;
    const Fp wm = m < half ? Fp::raw(a.tw_fwd[m]) : -Fp::raw(a.tw_fwd[m - half]);
    const Fp4 x = Fp4::from_base(sh * wm);
    d[2 * i] = x - zeta;
    d[2 * i + 1] = x - zn;
  }
  batch_inverse<8>(d);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t pt = pt0 + i;
    Fp4 s1 = Fp4::zero(), s2 = Fp4::zero();
    for (int ch = 0; ch < nchunks; ++ch) {
      s1 += m_load_fp4(a.partial + ((((size_t)b * nchunks + ch) * 2 + 0) * n + pt) * 4);
      s2 += m_load_fp4(a.partial + ((((size_t)b * nchunks + ch) * 2 + 1) * n + pt) * 4);
    }
    Fp4 g = (s1 - b1) * d[2 * i] + (s2 - b2) * d[2 * i + 1];
    uint32_t* o = a.out + (size_t)b * a.out_bstride + pt * 4;
    if (a.accumulate) g += m_load_fp4(o);
    m_store_fp4(o, g);
  }
}

void launch_machine_reduce(hipStream_t stream, const MReduceArgs& a) {
  const size_t n = (size_t)2 << a.logh;
  const int n1 = a.mats[0].width + a.mats[1].width + a.mats[2].width + a.mats[3].width;
  const int n2 = a.mats[1].width + a.mats[2].width;
  // (chunk lengths are multiples of 8: the kernel brings its sums back to range every eight columns of a chunk)
  const int split = a.batch >= 8 ? 1 : 8 / a.batch;
  // (at most sixteen chunks: the final kernel adds a point's partial sums one after the other)
  const int chunk_cols = std::min(kMReduceChunk, std::max({16, kMReduceChunk / split, (n1 / 16 + 7) & ~7}));
  const int nchunks = (n1 + chunk_cols - 1) / chunk_cols;
  hipLaunchKernelGGL(mreduce_bsum_kernel, dim3(a.batch), dim3(kMT), 0, stream, a, n1, n2);
  hipLaunchKernelGGL(mreduce_partial_kernel, dim3((unsigned)((n / 4 + kMT - 1) / kMT), nchunks, a.batch), dim3(kMT), 0, stream, a,
                     nchunks, n1, chunk_cols);
  hipLaunchKernelGGL(mreduce_final_kernel, dim3((unsigned)((n / 4 + kMT - 1) / kMT), a.batch), dim3(kMT), 0, stream, a, nchunks);
}

// ---- the same for a small batch, height by height (MReduceMulti, kernels_machine.h) ----
__global__ __launch_bounds__(kMT) void mreduce_multi_bsum_kernel(MReduceMulti a) {
  __shared__ Fp4 red[kMT / 64];
  const MRHeight& hh = a.heights[blockIdx.x];
  const int b = blockIdx.y;
  const uint32_t* ap = a.af_pows + (size_t)b * a.af_bstride;
  const uint32_t* op = a.opened + (size_t)b * a.opened_bstride;
  Fp4 s1 = Fp4::zero(), s2 = Fp4::zero();
  for (int c = 0; c < hh.nchips; ++c) {
    const MRChip ch = a.chips[hh.chip0 + c];
    for (int i = threadIdx.x; i < ch.n1; i += kMT) s1 += m_load_fp4(ap + (size_t)(ch.open_off + i) * 4) * m_load_fp4(op + (size_t)(ch.open_off + i) * 4);
    for (int i = threadIdx.x; i < ch.n2; i += kMT)
      s2 += m_load_fp4(ap + (size_t)(ch.open_off + ch.n1 + i) * 4) * m_load_fp4(op + (size_t)(ch.open_off + ch.n1 + i) * 4);
  }
  const Fp4 r1 = m_block_sum(s1, red), r2 = m_block_sum(s2, red);
  if (threadIdx.x == 0) {
    m_store_fp4(a.bsum + (((size_t)b * a.n_heights + blockIdx.x) * 2 + 0) * 4, r1);
    m_store_fp4(a.bsum + (((size_t)b * a.n_heights + blockIdx.x) * 2 + 1) * 4, r2);
  }
}
// A workgroup takes kMReduceMultiPoints LDE points of one height of one proof; lane (quad of points, part) adds the
// columns 4 part .. 4 part + 3, then sixteen further on, ... of every matrix of the height - four columns' loads in flight
// per trip, both factors centred, the signed 64-bit sums brought back to range every eight terms - the four parts are added
// through LDS and part 0 divides by the points' denominators (eight of them with one inversion).
__global__ __launch_bounds__(kMT) void mreduce_multi_kernel(MReduceMulti a) {
  constexpr int kQuads = kMReduceMultiPoints / 4, kParts = kMT / kQuads;
  static_assert(kParts == 4 && kQuads == 64, "64 quads of points x 4 parts");
  __shared__ uint32_t part_sum[kParts - 1][2][4][4][kQuads];  // [part][point of zeta / zeta w][point of the quad][limb][quad]
  int t = 0;
  while (t + 1 < a.n_heights && (int)blockIdx.x >= a.heights[t + 1].blk0) ++t;
  const MRHeight& hh = a.heights[t];
  const size_t h = (size_t)1 << hh.logh, n = 2 * h;
  const int per_proof = (int)((n + kMReduceMultiPoints - 1) / kMReduceMultiPoints), local = (int)blockIdx.x - hh.blk0;
  const int b = local / per_proof, lq = threadIdx.x % kQuads, part = threadIdx.x / kQuads;
  const size_t pt = ((size_t)(local % per_proof) * kQuads + lq) * 4;
  const bool in_range = pt < n;  // (n is a multiple of four)
  const size_t ptc = in_range ? pt : 0;
  const uint32_t* ap = a.af_pows + (size_t)b * a.af_bstride;
  int64_t acc1[4][4], acc2[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[q][j] = acc2[q][j] = 0;
  int pending = 0;
  for (int sgi = 0; sgi < hh.nseg; ++sgi) {
    const MRSeg sg = a.segs[hh.seg0 + sgi];
    const uint32_t* col = sg.p + (size_t)b * sg.bstride + ptc;
    const bool two = sg.pow2 >= 0;
    for (int j0 = 4 * part; j0 < sg.width; j0 += 4 * kParts) {
      uint4 v[4], p1[4], p2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + u, sg.width - 1);
        v[u] = *reinterpret_cast<const uint4*>(col + (size_t)j * n);
        if (j0 + u >= sg.width) v[u] = make_uint4(0u, 0u, 0u, 0u);
        p1[u] = *reinterpret_cast<const uint4*>(ap + (size_t)(sg.pow1 + j) * 4);
        p2[u] = two ? *reinterpret_cast<const uint4*>(ap + (size_t)(sg.pow2 + j) * 4) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int32_t w[4] = {fps_centre(v[u].x), fps_centre(v[u].y), fps_centre(v[u].z), fps_centre(v[u].w)};
        const int32_t al[4] = {fps_centre(p1[u].x), fps_centre(p1[u].y), fps_centre(p1[u].z), fps_centre(p1[u].w)};
        const int32_t al2[4] = {two ? fps_centre(p2[u].x) : 0, two ? fps_centre(p2[u].y) : 0, two ? fps_centre(p2[u].z) : 0,
                                two ? fps_centre(p2[u].w) : 0};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc1[q][j] += (int64_t)al[j] * (int64_t)w[q];
            acc2[q][j] += (int64_t)al2[j] * (int64_t)w[q];
          }
      }
      pending += 4;
      if (pending >= 8) {  // |alpha|, |w| <= (p - 1) / 2: eight products and a shrunk sum (< 2^58.2) stay below 2^63
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc1[q][j] = m_lazy_shrink(acc1[q][j]); acc2[q][j] = m_lazy_shrink(acc2[q][j]); }
        pending = 0;
      }
    }
  }
  if (part) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        part_sum[part - 1][0][q][j][lq] = fps_canon(fps_fold(acc1[q][j]));
        part_sum[part - 1][1][q][j][lq] = fps_canon(fps_fold(acc2[q][j]));
      }
  }
  __syncthreads();
  if (part != 0 || !in_range) return;
  const Fp4 b1 = m_load_fp4(a.bsum + (((size_t)b * a.n_heights + t) * 2 + 0) * 4), b2 = m_load_fp4(a.bsum + (((size_t)b * a.n_heights + t) * 2 + 1) * 4);
  const Fp4 zeta = m_load_fp4(a.zeta + (size_t)b * 4), zn = zeta * Fp::raw(hh.w_h);
  const int c = pt >= h ? 1 : 0;  // (h is a multiple of four: the four points lie on one coset)
  const size_t half = h >> 1;
  const Fp sh = Fp::raw(c ? hh.shift[1] : hh.shift[0]);
  Fp4 d[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t m = pt + i - (size_t)c * h;
    const Fp wm = m < half ? Fp::raw(hh.tw_fwd[m]) : -Fp::raw(hh.tw_fwd[m - half]);
    const Fp4 x = Fp4::from_base(sh * wm);
    d[2 * i] = x - zeta;
    d[2 * i + 1] = x - zn;
  }
  batch_inverse<8>(d);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    Fp4 s1, s2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1.c[j] = Fp::raw(fps_canon(fps_fold(acc1[q][j]))); s2.c[j] = Fp::raw(fps_canon(fps_fold(acc2[q][j]))); }
    for (int o = 0; o < kParts - 1; ++o) {
      Fp4 u1, u2;
#pragma unroll
      for (int j = 0; j < 4; ++j) { u1.c[j] = Fp::raw(part_sum[o][0][q][j][lq]); u2.c[j] = Fp::raw(part_sum[o][1][q][j][lq]); }
      s1 += u1;
      s2 += u2;
    }
    m_store_fp4(hh.out + (size_t)b * hh.out_bstride + (pt + q) * 4, (s1 - b1) * d[2 * q] + (s2 - b2) * d[2 * q + 1]);
  }
}
void launch_machine_reduce_multi(hipStream_t stream, const MReduceMulti& a) {
  hipLaunchKernelGGL(mreduce_multi_bsum_kernel, dim3(a.n_heights, a.batch), dim3(kMT), 0, stream, a);
  hipLaunchKernelGGL(mreduce_multi_kernel, dim3(a.total_blocks), dim3(kMT), 0, stream, a);
}

__global__ __launch_bounds__(kMT) void fri_add_kernel(uint32_t* __restrict__ layer, size_t layer_bstride,
                                                     const uint32_t* __restrict__ g, size_t g_bstride, size_t n) {
  const size_t i = (size_t)blockIdx.x * kMT + threadIdx.x;
  if (i >= n) return;
  uint32_t* p = layer + (size_t)blockIdx.y * layer_bstride + i * 4;
  m_store_fp4(p, m_load_fp4(p) + m_load_fp4(g + (size_t)blockIdx.y * g_bstride + i * 4));
}
void launch_fri_add(hipStream_t stream, uint32_t* layer, size_t layer_bstride, const uint32_t* g, size_t g_bstride, size_t n,
                    int batch) {
  hipLaunchKernelGGL(fri_add_kernel, dim3((unsigned)((n + kMT - 1) / kMT), batch), dim3(kMT), 0, stream, layer, layer_bstride, g,
                     g_bstride, n);
}

// ===========================================================================================
// proof assembly
// ===========================================================================================
__device__ __forceinline__ size_t m_layer_off(int logn, int layer) { return ((size_t)2 << logn) - ((size_t)2 << (logn - layer)); }
__device__ __forceinline__ uint32_t canon(uint32_t monty) { return Fp::raw(monty).to_canonical(); }

__global__ __launch_bounds__(kMT) void machine_assemble_kernel(MAssembleArgs a) {
  const int b = blockIdx.y, q = blockIdx.x, lm = a.lm, tid = threadIdx.x;
  uint32_t* body = a.body + (size_t)b * a.body_stride;
  const uint32_t* fl = a.fri_layers + (size_t)b * a.fri_layer_stride;
  const uint32_t* ft = a.fri_trees + (size_t)b * a.fri_tree_stride;
  const size_t n_open_words = a.n_open * 4;
  // main root 8 | perm root 8 | cumulative sums 4 * 7 | quotient root 8 | opened | FRI roots | final | witness | queries
  const size_t off_cum = 16, off_qroot = off_cum + 4 * kNumChips, off_opened = off_qroot + 8,
               off_fri_roots = off_opened + n_open_words, off_final = off_fri_roots + 8 * (size_t)lm, off_witness = off_final + 4,
               off_queries = off_witness + 1;
  const size_t hmax = (size_t)1 << lm;
  if (q == a.n_queries) {
    if (tid < 8) {
      for (int r = 1; r < 4; ++r) {
        const MRound& R = a.round[r];
        const size_t nr = (size_t)2 << R.lm;
        const uint32_t v = canon(R.tree[(size_t)b * R.tree_bstride + (2 * nr - 2) * 8 + tid]);
        body[(r == 1 ? 0 : r == 2 ? 8 : off_qroot) + tid] = v;
      }
    }
    for (int t = tid; t < 4 * kNumChips; t += kMT) body[off_cum + t] = canon(a.cum[(size_t)b * 4 * kNumChips + t]);
    const uint32_t* op = a.opened + (size_t)b * a.opened_bstride;
    for (size_t t = tid; t < n_open_words; t += kMT) body[off_opened + t] = canon(op[t]);
    size_t toff = 0, loff = 0;
    for (int k = 0; k < lm; ++k) {
      const size_t hk = hmax >> k;
      if (tid < 8) body[off_fri_roots + 8 * (size_t)k + tid] = canon(ft[(toff + 2 * hk - 2) * 8 + tid]);
      toff += 2 * hk - 1;
      loff += 2 * hk * 4;
    }
    if (tid < 4) body[off_final + tid] = canon(fl[loff + tid]);
    if (tid == 0) body[off_witness] = a.witness[b];
    return;
  }
  size_t perq = 0;
  for (int r = 0; r < 4; ++r) {
    for (int c = 0; c < kNumChips; ++c) perq += (size_t)a.round[r].seg[c].width;
    perq += 8 * ((size_t)a.round[r].lm + 1);
  }
  for (int k = 0; k < lm; ++k) perq += 8 + 8 * (size_t)(lm - k);
  uint32_t* dst = body + off_queries + perq * (size_t)q;
  const size_t idx = a.indices[(size_t)b * a.n_queries + q];
  const size_t cs = idx >> lm, m = idx & (hmax - 1);
  for (int r = 0; r < 4; ++r) {
    const MRound& R = a.round[r];
    for (int c = 0; c < kNumChips; ++c) {
      const size_t h = (size_t)1 << R.logh[c], mm = m & (h - 1);
      const Seg& sg = R.seg[c];
      if (!sg.width) continue;
      const uint32_t* src = sg.p + (size_t)b * sg.bstride + cs * h + mm;
      for (int i = tid; i < sg.width; i += kMT) dst[i] = canon(src[(size_t)i * 2 * h]);
      dst += sg.width;
    }
    const int logn = R.lm + 1;
    const size_t hm = (size_t)1 << R.lm;
    const size_t pos = cs * hm + (R.lm ? (size_t)(__brev((uint32_t)(m & (hm - 1))) >> (32 - R.lm)) : 0);
    const uint32_t* tree = R.tree + (size_t)b * R.tree_bstride;
    for (int t = tid; t < 8 * logn; t += kMT) {
      const int l = t >> 3, j = t & 7;
      dst[t] = canon(tree[(m_layer_off(logn, l) + ((pos >> l) ^ 1)) * 8 + j]);
    }
    dst += 8 * logn;
  }
  size_t toff = 0, loff = 0;
  for (int k = 0; k < lm; ++k) {
    const int loghk = lm - k;
    const size_t hk = hmax >> k, half = hk >> 1, mk = m & (half - 1), leaf = cs * half + mk;
    if (tid < 4) {
      dst[tid] = canon(fl[loff + (cs * hk + mk) * 4 + tid]);
      dst[4 + tid] = canon(fl[loff + (cs * hk + mk + half) * 4 + tid]);
    }
    dst += 8;
    for (int t = tid; t < 8 * loghk; t += kMT) {
      const int l = t >> 3, j = t & 7;
      dst[t] = canon(ft[(toff + m_layer_off(loghk, l) + ((leaf >> l) ^ 1)) * 8 + j]);
    }
    dst += 8 * loghk;
    toff += 2 * hk - 1;
    loff += 2 * hk * 4;
  }
}

void launch_machine_assemble(hipStream_t stream, const MAssembleArgs& a) {
  hipLaunchKernelGGL(machine_assemble_kernel, dim3(a.n_queries + 1, a.batch), dim3(kMT), 0, stream, a);
}

}  // namespace zksp
