// golap_variants.h -- which form of the hot kernels a launch takes by default, and why.
//
// Five kernels exist in two forms: one whose loads and stores are `asm volatile` statements with hand-counted s_waitcnt vmcnt(N)
// (golap_hidden.h) and one that leaves every memory operation to the compiler.  The hidden forms are faster but their
// correctness rests on properties of the GENERATED code -- which registers hipcc leaves alone, how many memory instructions sit
// between a load and its wait -- that tests/test_isa_audit.py checks on the assembly and tests/test_join_gpu.py::
// test_hidden_forms_match_plain_forms checks on the GPU, for ONE compiler.  A library built by any other compiler has not had that
// audit: it launches the plain forms (3-5 % slower, no assumptions) until somebody has rerun both tests and moved the three numbers
// below.  Host-only header: no HIP runtime calls, includable from a plain C++ probe (tests/test_variants_cpu.py does that with a
// faked compiler version).
#pragma once
#include <hip/hip_version.h>

// the hipcc the hidden forms were audited with (hipcc --version: "HIP version: 7.2.26015")
#define GOLAP_HIDDEN_AUDITED_HIP_MAJOR 7
#define GOLAP_HIDDEN_AUDITED_HIP_MINOR 2
#define GOLAP_HIDDEN_AUDITED_HIP_PATCH 26015
// the hipcc building this translation unit (tests override the three to stand in for a compiler upgrade)
#ifndef GOLAP_BUILD_HIP_MAJOR
#define GOLAP_BUILD_HIP_MAJOR HIP_VERSION_MAJOR
#define GOLAP_BUILD_HIP_MINOR HIP_VERSION_MINOR
#define GOLAP_BUILD_HIP_PATCH HIP_VERSION_PATCH
#endif

namespace golap {

constexpr bool kHiddenFormsAudited = GOLAP_BUILD_HIP_MAJOR == GOLAP_HIDDEN_AUDITED_HIP_MAJOR && GOLAP_BUILD_HIP_MINOR == GOLAP_HIDDEN_AUDITED_HIP_MINOR &&
                                     GOLAP_BUILD_HIP_PATCH == GOLAP_HIDDEN_AUDITED_HIP_PATCH;

// default selector values of the five kernels (the GOLAP_P1N / GOLAP_P2R / GOLAP_P1W / GOLAP_P2W / GOLAP_AGG_RSV knobs override them one by one)
struct KernelForms {
    int p1n;              // first pass of the tuple pipeline: 118 = k_part1_narrow<18, .., ASM>, 18 = the same tile with compiler-visible memory operations
    int p2r;              // ring pass: 104 = k_part2_ring<.., ASM>, 4 = plain
    int p1w;              // first pass of the 64-bit-row pipeline: 1 = hidden, 2 = plain
    int p2w;              // last pass of the 64-bit-row pipeline (k_part2_wide): 1 = hidden, 2 = plain
    bool agg_rsv_hidden;  // GROUP BY reserving scatter k_agg_scatter_rsv: hidden / plain
    bool hidden;          // what was asked for
};
constexpr KernelForms kernel_forms(bool hidden) {
    return hidden ? KernelForms{118, 104, 1, 1, true, true} : KernelForms{18, 4, 2, 2, false, false};
}
// `override_knob`: the GOLAP_HIDDEN_FORMS knob ("0" = plain forms, "1" = hidden forms whatever built the library -- what a developer
// sets to run the GPU half of the audit on a new compiler's build), or null
inline KernelForms default_kernel_forms(const char* override_knob) {
    bool hidden = kHiddenFormsAudited;
    if (override_knob && override_knob[0] == '0') hidden = false;
    if (override_knob && override_knob[0] == '1') hidden = true;
    return kernel_forms(hidden);
}

}  // namespace golap
