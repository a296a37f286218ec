// rocrand_pin.cc -- TEST INFRASTRUCTURE ONLY (checker of the checker).
//
// cuRAND is not in this image, so the XORWOW restatement in ggms_oracle.c (and the HIP Xorwow of
// xgnn_amd/csrc/ggms_device.h) cannot be run against it.  rocRAND ships a host-callable XORWOW
// (rocrand_device::xorwow_engine, /opt/rocm/include/rocrand/rocrand_xorwow.h:69-175) with the SAME
// recurrence (Marsaglia's xorwow: 5-word xorshift + Weyl counter, increment 362437) and the same
// base state {123456789, 362436069, 521288629, 88675123, 5783321; d = 6615241}, seeded through the
// same scramble structure with DIFFERENT salts / multipliers.  This file exposes that third-party
// implementation to the tests:
//   pin_xorwow_draws      load an arbitrary {d, v0..v4} into rocRAND's engine, draw n words
//   pin_rocrand_init      rocRAND's own init(seed, 0, 0): the state it produces
// Built with hipcc (host code only; the header needs the HIP runtime headers).
#include <cstddef>
#include <cstdint>

#include <rocrand/rocrand_xorwow.h>

namespace {
struct Probe : rocrand_device::xorwow_engine {
  Probe() : rocrand_device::xorwow_engine(0ull, 0ull, 0ull) {}
  void load(const uint32_t *s) {
    m_state.d = s[0];
    for (int i = 0; i < 5; ++i) m_state.x[i] = s[1 + i];
  }
  void store(uint32_t *s) const {
    s[0] = m_state.d;
    for (int i = 0; i < 5; ++i) s[1 + i] = m_state.x[i];
  }
};
} // namespace

extern "C" {

// state6 = {d, v0, v1, v2, v3, v4} (the layout of orc_xorwow_t and of the HIP pool); updated in place
void pin_xorwow_draws(uint32_t *state6, size_t n, uint32_t *out) {
  Probe p;
  p.load(state6);
  for (size_t i = 0; i < n; ++i) out[i] = p.next();
  p.store(state6);
}

// XOR-fold of n draws (cheap check of long streams without moving them)
uint32_t pin_xorwow_fold(uint32_t *state6, size_t n) {
  Probe p;
  p.load(state6);
  uint32_t acc = 0;
  for (size_t i = 0; i < n; ++i) acc = (acc << 1 | acc >> 31) ^ p.next();
  p.store(state6);
  return acc;
}

void pin_rocrand_init(uint64_t seed, uint32_t *state6) {
  Probe p;
  static_cast<rocrand_device::xorwow_engine &>(p) = rocrand_device::xorwow_engine(seed, 0ull, 0ull);
  p.store(state6);
}

} // extern "C"
