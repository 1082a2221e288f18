// Which vector extensions this host has - asked in a translation unit compiled WITHOUT them (plain C++, no -mavx2 / -mavx512f), so
// that the question itself cannot execute an instruction the CPU lacks.  The answers gate p2_avx2.cpp and p2_avx512.cpp
// (host_hash.hpp vector_level).
namespace zksp {
namespace p2avx2 {
bool usable() {
  __builtin_cpu_init();
  return __builtin_cpu_supports("avx2");
}
}  // namespace p2avx2
namespace p2avx512 {
bool usable() {
  __builtin_cpu_init();
  return __builtin_cpu_supports("avx512f");
}
}  // namespace p2avx512
}  // namespace zksp
