// mickey_amd -- attention forward, one wave per SIMD (work in progress: the launcher reports "not handled" and the
// dispatcher in mk_attention.hip runs the two-waves-per-SIMD kernel).
#include "mk_common.hpp"

namespace mk {
bool launch_attn_w1(const void*, const void*, const void*, void*, int, int, int, int, int, int, hipStream_t) { return false; }
}  // namespace mk
