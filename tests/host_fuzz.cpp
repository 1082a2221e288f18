// Sanitiser harness for the host side of the boundary (CPU build only; GPU ASan is not
// available on this pool): mutates ELF images, stdin buffers and proof bytes and drives
// load_elf / execute / parse_proof_header / verify_proof and, for machine proofs, parse_machine_header /
// verify_machine_proof under ASan + UBSan.
// Built and run by tests/test_host_sanitizers.py.  Exit code 0 = no finding.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <vector>

#include "executor.hpp"
#include "machine.hpp"
#include "mverifier.hpp"
#include "verifier.hpp"

using namespace zksp;

static std::vector<uint8_t> slurp(const char* p) {
  std::ifstream f(p, std::ios::binary);
  return {std::istreambuf_iterator<char>(f), {}};
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: host_fuzz ELF STDIN PROOF [iters [MACHINE_PROOF queries pow_bits]]\n");
    return 2;
  }
  const std::vector<uint8_t> elf = slurp(argv[1]), input = slurp(argv[2]), proof = slurp(argv[3]);
  const int iters = argc > 4 ? atoi(argv[4]) : 200;
  std::mt19937_64 rng(12345);
  ElfImage good;
  if (!load_elf(elf.data(), elf.size(), &good).empty()) return 3;
  uint32_t vk[8];
  memcpy(vk, proof.data() + 22 * 4, 32);
  std::string err;
  if (verify_proof(proof.data(), proof.size(), vk, 12, 8, &err) != 0) {
    fprintf(stderr, "baseline proof rejected: %s\n", err.c_str());
    return 4;
  }
  size_t elf_ok = 0, exec_halt = 0, ver_ok = 0;
  // machine proofs: the verifying key is rebuilt from the ELF, the baseline proof must verify
  std::vector<uint8_t> mproof;
  MachineVk mvk{};
  uint32_t mq = 0, mpow = 0;
  if (argc > 7) {
    mproof = slurp(argv[5]);
    mq = (uint32_t)atoi(argv[6]);
    mpow = (uint32_t)atoi(argv[7]);
    MachineProgram prog;
    if (!build_machine_program(good, KeccakMode::kReplace, &prog).empty()) return 6;
    machine_host_setup(prog, &mvk);
    if (verify_machine_proof(mproof.data(), mproof.size(), mvk, mq, mpow, &err) != 0) {
      fprintf(stderr, "baseline machine proof rejected: %s\n", err.c_str());
      return 7;
    }
  }
  for (int it = 0; it < iters; ++it) {
    // 1. corrupted ELF headers / bodies
    {
      std::vector<uint8_t> e = elf;
      int n = 1 + (int)(rng() % 8);
      for (int k = 0; k < n; ++k) {
        size_t pos = (it % 3 == 0) ? rng() % 256 : rng() % e.size();  // bias towards the headers
        e[pos] = (uint8_t)rng();
      }
      if (it % 7 == 0) e.resize(rng() % e.size());
      ElfImage img;
      if (load_elf(e.data(), e.size(), &img).empty()) {
        ++elf_ok;
        ExecOptions o;
        o.max_cycles = 200000;
        o.keccak_mode = (KeccakMode)(it % 3);
        ExecutionRecord r = execute(img, {input}, o);
        exec_halt += r.halted;
      }
    }
    // 1b. hostile section-header table: e_shoff / e_shentsize / e_shnum and the symtab's sh_link are
    // attacker-controlled; the loader must not read a linked string-table header outside the file
    {
      std::vector<uint8_t> e = elf;
      auto put16 = [&](size_t o, uint16_t v) { e[o] = (uint8_t)v; e[o + 1] = (uint8_t)(v >> 8); };
      auto put32 = [&](size_t o, uint32_t v) { for (int k = 0; k < 4; ++k) e[o + k] = (uint8_t)(v >> (8 * k)); };
      const uint32_t shoff = (uint32_t)e[32] | (uint32_t)e[33] << 8 | (uint32_t)e[34] << 16 | (uint32_t)e[35] << 24;
      const uint16_t shentsize = (uint16_t)(e[46] | e[47] << 8), shnum = (uint16_t)(e[48] | e[49] << 8);
      switch (it % 4) {
        case 0: put16(48, 0xffff); break;                         // e_shnum far beyond the table
        case 1: put16(46, (uint16_t)(rng() | 0x8000)); break;     // huge e_shentsize
        case 2: put32(32, (uint32_t)(e.size() - (rng() % 64))); break;  // table starts at the end of the file
        case 3:                                                   // every symtab links to a far section
          for (uint16_t i = 0; i < shnum; ++i) {
            size_t o = (size_t)shoff + (size_t)i * shentsize;
            if (o + 40 <= e.size() && e[o + 4] == 2) { put32(o + 24, 0xfff0 + (uint32_t)(rng() % 15)); put16(48, 0xffff); }
          }
          break;
      }
      ElfImage img;
      if (load_elf(e.data(), e.size(), &img).empty()) ++elf_ok;
    }
    // 2. corrupted stdin
    {
      std::vector<uint8_t> s = input;
      int n = 1 + (int)(rng() % 4);
      for (int k = 0; k < n; ++k) s[rng() % s.size()] = (uint8_t)rng();
      if (it % 5 == 0) s.resize(rng() % s.size());
      ExecOptions o;
      o.max_cycles = 3000000;
      o.keccak_mode = (KeccakMode)(it % 3);
      ExecutionRecord r = execute(good, {s}, o);
      exec_halt += r.halted;
    }
    // 3. corrupted proofs
    {
      std::vector<uint8_t> p = proof;
      int n = 1 + (int)(rng() % 4);
      for (int k = 0; k < n; ++k) {
        size_t pos = (it % 2 == 0) ? rng() % 200 : rng() % p.size();
        p[pos] ^= (uint8_t)(1u << (rng() % 8));
      }
      if (it % 11 == 0) p.resize(rng() % p.size());
      ProofHeader h;
      std::string e2;
      if (parse_proof_header(p.data(), p.size(), &h, &e2)) {
        int rc = verify_proof(p.data(), p.size(), vk, 12, 8, &e2);
        if (rc == 0) ++ver_ok;
      }
    }
    // 4. corrupted machine proofs: header words (heights, lengths), body words, truncations
    if (!mproof.empty() && it % 3 == 0) {  // a third of the iterations: one verification costs ~1 s under ASan
      std::vector<uint8_t> p = mproof;
      int n = 1 + (int)(rng() % 4);
      for (int k = 0; k < n; ++k) {
        size_t pos = (it % 2 == 0) ? rng() % 256 : rng() % p.size();
        p[pos] ^= (uint8_t)(1u << (rng() % 8));
      }
      if (it % 12 == 0) p.resize(rng() % p.size());
      if (it % 15 == 0) p.resize(p.size() + 4 * (rng() % 64), 0);
      MachineHeader h;
      std::string e2;
      if (parse_machine_header(p.data(), p.size(), &h, &e2) && verify_machine_proof(p.data(), p.size(), mvk, mq, mpow, &e2) == 0)
        ++ver_ok;
    }
  }
  printf("fuzz ok: %d iterations, %zu mutated ELFs loaded, %zu guest runs halted, %zu mutated proofs accepted\n", iters,
         elf_ok, exec_halt, ver_ok);
  return ver_ok == 0 ? 0 : 5;  // a mutated proof must never verify
}
