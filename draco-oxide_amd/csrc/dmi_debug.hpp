// dmi_debug.hpp — the switches of the call a thread is working for (dmi_debug, include/draco_mi.h) and the process options, for host and device sources.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>

#include "../../include/draco_mi.h"

namespace dmi {
// Every C-ABI entry point that takes a dmi_config opens a DebugScope with its `debug` pointer (NULL keeps what the thread already has: a nested call, or the
// process default); objects that outlive the call (jobs, transcoders) keep a COPY and open the scope with that.  Worker threads do not inherit thread-locals:
// the helpers below (parallel_for, guarded_pool, with_debug) carry the spawning thread's pointer across.
const dmi_debug* dbg_ptr();                     // never null
inline const dmi_debug& dbg() { return *dbg_ptr(); }
inline bool dbg_on(uint64_t flag) { return (dbg().flags & flag) != 0; }
struct DebugScope {
  const dmi_debug* prev;
  explicit DebugScope(const dmi_debug* d);
  ~DebugScope();
  DebugScope(const DebugScope&) = delete;
  DebugScope& operator=(const DebugScope&) = delete;
};
// a thread body that works under the spawning thread's switches: dmi::Thread(with_debug([&] { … }))
template <class F> inline auto with_debug(F f) {
  const dmi_debug* cur = dbg_ptr();
  return [cur, f = std::move(f)](auto&&... args) mutable { DebugScope scope(cur); return f(std::forward<decltype(args)>(args)...); };
}
uint32_t process_flags();                       // DMI_PROCESS_* (dmi_configure_process)
size_t decode_budget_bytes();                   // bytes a decode call may allocate for what a file ASKS for before anything is known to be real (default 16 GiB)
size_t device_cache_limit_mb();                 // 0 = default (half of the device's memory, at most 64 GiB)
}  // namespace dmi
