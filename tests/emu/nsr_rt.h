// tests/emu/nsr_rt.h -- TEST INFRASTRUCTURE: host stand-in for the HIP runtime bits nsr_api.cpp uses.
#pragma once
namespace nsr {
inline const char *rt_check_last() { return nullptr; }
inline void rt_record(void *, void *) {}
inline int rt_current_device() { return 0; }
template <typename K>
inline const char *rt_allow_lds(K, int) { return nullptr; }
}  // namespace nsr
