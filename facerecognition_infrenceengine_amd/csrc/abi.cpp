// Error plumbing + version of the C ABI (include/frhip.h).
#include "common.h"
#include <cstring>

static thread_local char g_err[512] = "";

void fr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int fr_version(void) { return FR_ABI_VERSION; }
extern "C" const char* fr_last_error_string(void) { return g_err; }
extern "C" int fr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- fr_detect_sequence: a recorded run of detector calls replayed by ONE C call (the eager single-frame path is bound by
// the interpreter: ~50 ctypes calls per frame).  An entry names one of the detector's entry points and carries its
// arguments as 8-byte slots in declaration order (pointers and size_t as they are, ints sign-extended, floats as their bits).
namespace {
inline void* sp(const fr_call& c, int i) { return reinterpret_cast<void*>(static_cast<uintptr_t>(c.a[i])); }
inline int si(const fr_call& c, int i) { return static_cast<int>(static_cast<int64_t>(c.a[i])); }
inline float sf(const fr_call& c, int i) { union { uint32_t b; float f; } u; u.b = static_cast<uint32_t>(c.a[i]); return u.f; }
}
extern "C" int fr_detect_sequence(const fr_call* calls, int ncalls) {
    FR_REQUIRE(calls && ncalls > 0, "fr_detect_sequence: no calls");
    // arity of every entry first: nothing is launched from a list that holds a short or malformed call
    static const int arity[10] = {-1, 18, 21, 16, 18, 8, 13, 14, 2, 2};
    for (int k = 0; k < ncalls; ++k) {
        const int fn = calls[k].fn;
        FR_REQUIRE(fn >= 1 && fn <= 9, "fr_detect_sequence: call %d: unknown function id %d", k, fn);
        FR_REQUIRE(calls[k].nargs == arity[fn], "fr_detect_sequence: call %d (function id %d): %d arguments, expected %d", k, fn,
                   calls[k].nargs, arity[fn]);
    }
    for (int k = 0; k < ncalls; ++k) {
        const fr_call& c = calls[k];
        int rc;
#define PF(i) static_cast<const float*>(sp(c, i))
#define PFM(i) static_cast<float*>(sp(c, i))
#define PI(i) static_cast<const int32_t*>(sp(c, i))
#define PIM(i) static_cast<int32_t*>(sp(c, i))
        switch (c.fn) {
            case FR_FN_DCONV_MFMA:
                rc = fr_dconv_mfma_f32(si(c, 0), PF(1), PF(2), PF(3), PF(4), PFM(5), si(c, 6), si(c, 7), si(c, 8), PF(9), PF(10),
                                       static_cast<const uint8_t*>(sp(c, 11)), si(c, 12), si(c, 13), PI(14), si(c, 15), sp(c, 16), sp(c, 17));
                break;
            case FR_FN_PNET23:
                rc = fr_pnet23_split_f16(PF(0), sp(c, 1), si(c, 2), si(c, 3), si(c, 4), PF(5), PF(6), PF(7), PF(8), PF(9), PF(10), PF(11), PF(12),
                                         PFM(13), si(c, 14), sf(c, 15), sf(c, 16), PIM(17), sp(c, 18), static_cast<size_t>(c.a[19]), sp(c, 20));
                break;
            case FR_FN_PNET_CANDIDATES:
                rc = fr_pnet_candidates(PF(0), si(c, 1), si(c, 2), si(c, 3), sf(c, 4), sf(c, 5), si(c, 6), PFM(7), PFM(8), PFM(9), PIM(10), PIM(11),
                                        PFM(12), PF(13), sf(c, 14), sp(c, 15));
                break;
            case FR_FN_SORT_NMS:
                rc = fr_sort_nms(PF(0), PF(1), PF(2), si(c, 3), PI(4), si(c, 5), si(c, 6), si(c, 7), si(c, 8), sf(c, 9), si(c, 10), si(c, 11),
                                 PFM(12), PFM(13), PFM(14), PIM(15), si(c, 16), sp(c, 17));
                break;
            case FR_FN_BOX_REFINE:
                rc = fr_box_refine(PFM(0), PF(1), si(c, 2), PI(3), si(c, 4), si(c, 5), si(c, 6), sp(c, 7));
                break;
            case FR_FN_CROP_CONV1:
                rc = fr_crop_conv1_f32(si(c, 0), static_cast<const uint8_t*>(sp(c, 1)), si(c, 2), si(c, 3), si(c, 4), PF(5), PI(6), si(c, 7),
                                       PF(8), PF(9), PF(10), PFM(11), sp(c, 12));
                break;
            case FR_FN_STAGE_SELECT:
                rc = fr_stage_select(PF(0), PF(1), si(c, 2), PI(3), si(c, 4), si(c, 5), sf(c, 6), PFM(7), PFM(8), PFM(9), si(c, 10), PIM(11),
                                     PFM(12), sp(c, 13));
                break;
            case FR_FN_EVENT_RECORD:      // (event, stream): the pyramid levels of a recorded call run on side streams, forked from
            case FR_FN_STREAM_WAIT: {     // (stream, event)  and joined to the caller's stream by the caller's own events
                const hipError_t e = c.fn == FR_FN_EVENT_RECORD
                    ? hipEventRecord(static_cast<hipEvent_t>(sp(c, 0)), static_cast<hipStream_t>(sp(c, 1)))
                    : hipStreamWaitEvent(static_cast<hipStream_t>(sp(c, 0)), static_cast<hipEvent_t>(sp(c, 1)), 0);
                if (e != hipSuccess) { fr_set_error("fr_detect_sequence: call %d: %s", k, hipGetErrorString(e)); return FR_E_LAUNCH; }
                rc = FR_OK;
                break;
            }
            default:
                FR_REQUIRE(false, "fr_detect_sequence: call %d: unknown function id %d", k, c.fn);
        }
#undef PF
#undef PFM
#undef PI
#undef PIM
        if (rc != FR_OK) return rc;
    }
    return FR_OK;
}
