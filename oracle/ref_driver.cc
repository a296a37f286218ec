/*
 * ref_driver.cc -- TEST INFRASTRUCTURE ONLY.
 *
 * C-ABI wrapper around the reference's own CPU leaf functions, which the
 * Makefile compiles *in place* from /root/reference (no reference source is
 * copied into this repository).  Linked objects:
 *   samgraph/common/cpu/cpu_sampling_khop0.cc   CPUSampleKHop0
 *   samgraph/common/cpu/cpu_sampling_khop2.cc   CPUSampleKHop2
 *   samgraph/common/cpu/cpu_random.cc           RandomID
 *   samgraph/common/cpu/cpu_extraction.cc       CPUExtract
 *   samgraph/common/{run_config,logging,constant}.cc   (their link deps)
 *
 * The only symbols defined here on the reference's behalf are the two
 * getenv helpers below: their definitions live in common.cc:508-525, a file
 * that cannot be compiled in this image because it includes
 * <cuda_runtime.h>.  They are not on the arithmetic path (run_config.cc and
 * logging.cc read log level / env overrides through them).
 *
 * Everything the reference needs a Device for (CPUHashTable*, shufflers,
 * engines) is deliberately NOT built: that would need stand-ins for the
 * CUDA-backed Device classes.
 */
#include <cstdlib>
#include <string>

#include "common.h"
#include "cpu/cpu_function.h"
#include "run_config.h"

namespace samgraph {
namespace common {

std::string GetEnv(std::string key) {
  const char *v = getenv(key.c_str());
  return v ? std::string(v) : std::string("");
}

bool IsEnvSet(std::string key) {
  std::string val = GetEnv(key);
  return val == "ON" || val == "1";
}

}  // namespace common
}  // namespace samgraph

using namespace samgraph::common;

extern "C" {

void ref_set_omp_threads(int n) { RunConfig::omp_thread_num = n; }

/* cpu/cpu_sampling_khop0.cc:29-83 */
void ref_cpu_sample_khop0(const uint32_t *indptr, const uint32_t *indices,
                          const uint32_t *input, size_t num_input,
                          uint32_t *out_src, uint32_t *out_dst,
                          size_t *num_out, size_t fanout) {
  cpu::CPUSampleKHop0(indptr, indices, input, num_input, out_src, out_dst,
                      num_out, fanout);
}

/* cpu/cpu_sampling_khop2.cc:29-76 (permutes `indices` in place) */
void ref_cpu_sample_khop2(const uint32_t *indptr, uint32_t *indices,
                          const uint32_t *input, size_t num_input,
                          uint32_t *out_src, uint32_t *out_dst,
                          size_t *num_out, size_t fanout) {
  cpu::CPUSampleKHop2(indptr, indices, input, num_input, out_src, out_dst,
                      num_out, fanout);
}

/* cpu/cpu_random.cc:26-30 */
uint32_t ref_random_id(uint32_t lo, uint32_t hi) {
  return cpu::RandomID(lo, hi);
}

/* cpu/cpu_extraction.cc:66-90; dtype codes are the reference's DataType enum
 * (common.h:38-46) */
void ref_cpu_extract(void *dst, const void *src, const uint32_t *index,
                     size_t num_index, size_t dim, int dtype) {
  cpu::CPUExtract(dst, src, index, num_index, dim,
                  static_cast<DataType>(dtype));
}

int ref_dtype_code(const char *name) {
  std::string s(name);
  if (s == "f32") return kF32;
  if (s == "f64") return kF64;
  if (s == "f16") return kF16;
  if (s == "u8") return kU8;
  if (s == "i32") return kI32;
  if (s == "i8") return kI8;
  if (s == "i64") return kI64;
  return -1;
}

}  // extern "C"
