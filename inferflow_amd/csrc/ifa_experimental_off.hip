// ifa_experimental_off.hip -- what the default library answers for the launches parked under csrc/experimental/ (VERDICT r4 item 10):
//   * k_dec_persist   (experimental/ifa_decode_persist.h, option "persist"): all layers of a token as ONE launch, LDS-ring loader
//     waves + consumer waves, 8-byte granule hand-offs -- bit-identical to the five-launch step, 57.3 us per layer against 43.0
//     (round 3, profiles/r03_persist_phase_trace.log);
//   * k_dec_wo_ffn    (experimental/ifa_decode_wo_ffn.h, option "fuse_wo_ffn"): the Wo rows in front of the W1 / W3 launch --
//     bit-identical, 19.8-24.6 us against 4.6 + 12.6 for the two launches (round 4, profiles/r04_wo_ffn_fused_trace.log);
//   * the WO instances of k_dec_qkv_attn (option "fuse_wo"): the Wo rows behind the attention in the same launch -- bit-identical,
//     796 vs 800 tok/s (round 4, profiles/r04_ab_options.log).
// Each was built, verified and measured slower than the four launches it would replace; they stay in the tree as measured
// dead ends with their traces, out of the default build (14 fewer template instances, two minutes less compile per format).
// IFA_EXPERIMENTAL=1 python -m inferflow_amd.build --force builds them in (and leaves this unit out).
#include "ifa_host.h"
#include "experimental/ifa_decode_persist_launch.h"
#include "experimental/ifa_decode_wo_ffn.h"

namespace ifa {

bool dec_persist_has(int, int, int, int) { return false; }
int dec_persist_launch(int, int, int, int, int, const PsParams &, int, size_t, hipStream_t)
{
    return ifa_fail(IFA_ERR_STATE, "persistent decode launch: an experimental kernel this library was built without (IFA_EXPERIMENTAL=1)");
}
bool dec_wo_ffn_supported(int, int, int, int, int, int, bool, int) { return false; }
int dec_wo_ffn_launch(int, bool, const DecGemvParams &, const DecGemvParams &, const DecWoFfnExtra &, int, hipStream_t)
{
    return ifa_fail(IFA_ERR_STATE, "fused Wo + FFN launch: an experimental kernel this library was built without (IFA_EXPERIMENTAL=1)");
}

} // namespace ifa

extern "C" int ifa_experimental_built(void) { return 0; }
