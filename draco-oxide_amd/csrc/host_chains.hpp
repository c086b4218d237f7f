// host_chains.hpp — one long rANS / rABS stream on one host core (see host_chains.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

#include "dmi_device.hpp"

namespace dmi {

// Grow-only byte output of one stream (kept by the job between encodes: no page faults after the first).
struct HostChainOut {
  uint8_t* data = nullptr;
  size_t cap = 0, len = 0;
  uint32_t err = 0;   // 0 ok, 1 StateTooLarge, 2 out of memory, 3 symbol outside the coding table
  HostChainOut() = default;
  HostChainOut(const HostChainOut&) = delete;
  HostChainOut& operator=(const HostChainOut&) = delete;
  ~HostChainOut();
  bool reserve(size_t need);
};

void host_rans_chain(const uint32_t* sym, uint64_t n, const RansEntry* table, uint32_t bins, uint32_t precision, HostChainOut& out);
void host_rans_chain16(const uint16_t* sym, uint64_t n, const RansEntry* table, uint32_t bins, uint32_t precision, HostChainOut& out);
void host_rabs_chain(const uint8_t* bits, uint64_t n, const RansEntry* entries /* [bit 0, bit 1] */, HostChainOut& out);

// the inverse coders (decoder side: dmi_decode.cpp)
bool host_rans_decode(const uint8_t* data, size_t len, const uint32_t* freq, uint32_t num_symbols, uint32_t precision, uint64_t n, uint32_t* out);
bool host_rabs_decode(const uint8_t* data, size_t len, uint32_t zero_prob, uint64_t n, uint8_t* out);

}  // namespace dmi
