"""Which kernel forms a build launches by default (csrc/golap_variants.h): the hand-scheduled forms with hidden memory operations
only when the library was built by the hipcc they were audited with, the plain forms otherwise.  A 15-line C++ probe includes the
very header the library uses -- once as it is, once with the build's compiler version faked -- and prints the defaults."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpu_olap_engine_amd", "csrc")
HDR = open(os.path.join(CSRC, "golap_variants.h")).read()

PROBE = r"""
#include <cstdio>
#include "golap_variants.h"
int main(int argc, char** argv) {
    const golap::KernelForms f = golap::default_kernel_forms(argc > 1 ? argv[1] : nullptr);
    printf("%d %d %d %d %d %d %d\n", (int)golap::kHiddenFormsAudited, f.p1n, f.p2r, f.p1w, f.p2w, (int)f.agg_rsv_hidden, (int)f.hidden);
    return 0;
}
"""


def _run(tmp_path, defines, *args):
    src = tmp_path / "probe.cpp"
    src.write_text(PROBE)
    exe = tmp_path / ("probe" + "".join(c for c in "".join(defines) if c.isalnum()))
    subprocess.check_call(["g++", "-std=c++17", "-I", CSRC, "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", *defines, str(src), "-o", str(exe)])
    return [int(x) for x in subprocess.check_output([str(exe), *args], text=True).split()]


def _audited():
    return tuple(int(re.search(r"#define GOLAP_HIDDEN_AUDITED_HIP_" + k + r" (\d+)", HDR).group(1)) for k in ("MAJOR", "MINOR", "PATCH"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/hip/hip_version.h"), reason="ROCm headers not installed")
def test_this_images_compiler_is_the_audited_one_and_selects_the_hidden_forms(tmp_path):
    assert _run(tmp_path, []) == [1, 118, 104, 1, 1, 1, 1]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/hip/hip_version.h"), reason="ROCm headers not installed")
@pytest.mark.parametrize("fake", [(0, 1, 0), (0, 0, 1), (1, 0, 0)])
def test_a_library_built_by_another_compiler_selects_the_plain_forms(tmp_path, fake):
    """a compiler upgrade (patch, minor or major) without a new audit: every one of the five kernels takes its plain form"""
    mj, mn, pt = (a + d for a, d in zip(_audited(), fake))
    defs = [f"-DGOLAP_BUILD_HIP_MAJOR={mj}", f"-DGOLAP_BUILD_HIP_MINOR={mn}", f"-DGOLAP_BUILD_HIP_PATCH={pt}"]
    assert _run(tmp_path, defs) == [0, 18, 4, 2, 2, 0, 0]
    # the developer's switch to run the GPU half of the audit on such a build, and the way back
    assert _run(tmp_path, defs, "1") == [0, 118, 104, 1, 1, 1, 1]
    assert _run(tmp_path, [], "0") == [1, 18, 4, 2, 2, 0, 0]


def test_the_isa_audit_and_the_header_name_the_same_compiler():
    import test_isa_audit
    assert test_isa_audit.VALIDATED_HIPCC == ".".join(str(v) for v in _audited())


def test_the_selectors_take_their_defaults_from_the_header():
    """no launch site hard-codes a hidden form as its default any more"""
    join = open(os.path.join(CSRC, "golap_join.hip")).read()
    ops = open(os.path.join(CSRC, "golap_ops.hip")).read()
    for fn, field in (("p1n_variant", "p1n"), ("p2r_variant", "p2r"), ("p1w_variant", "p1w")):
        m = re.search(r"static int " + fn + r"\(\) \{([^}]*)\}", join)
        assert m and f"kernel_forms_now().{field}" in m.group(1), fn
    assert "kernel_forms_now().agg_rsv_hidden" in ops
