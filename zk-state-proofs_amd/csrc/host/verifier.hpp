#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "context.hpp"

namespace zksp {

constexpr uint32_t kProofVersion = 2;  // 2: LogUp bus + public I/O list

struct ProofHeader {
  uint32_t log_h, n_perms, exit_code, pv_len;
  uint32_t pv_digest[8], deferred_digest[8], vk_digest[8];
  size_t pv_offset, io_offset, body_offset;  // bytes; io = n_perms * (25 u64 in || 25 u64 out)
};

bool parse_proof_header(const uint8_t* bytes, size_t len, ProofHeader* h, std::string* err);

// 0 = accepted; 7 = malformed; 8 = rejected (codes match include/zksp.h)
int verify_proof(const uint8_t* bytes, size_t len, const uint32_t vk_digest[8], uint32_t num_queries, uint32_t pow_bits,
                 std::string* err);

}  // namespace zksp
