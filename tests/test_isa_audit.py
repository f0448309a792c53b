"""Static audit of the kernels that hide memory operations from hipcc (csrc/golap_hidden.h: the ASM forms of csrc/golap_part_narrow.h
and csrc/golap_part_wide.h, and k_agg_scatter_rsv in csrc/golap_ops.hip).

Those kernels issue their loads and stores in `asm volatile` statements and wait for them with hand-counted `s_waitcnt vmcnt(N)`
(cdna_hip_programming.md 5.7).  That is only sound while the compiler (1) never spills -- a spilled or reloaded register may be the
destination of a load that has not landed (wrong data, or a memory fault when it held an address: seen once on MI355X with a build
that had 36 bytes of scratch), (2) never touches a hidden load's destination registers between the load and the counted wait, and
(3) adds no `vmcnt` wait of its own inside the tile loop (it would drain the prefetch on every tile).  All three are properties of
the generated code, so they are checked on the generated code: golap_join.hip and golap_ops.hip are compiled to gfx950 assembly (no
GPU needed)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpu_olap_engine_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
# The instantiations the product is known to launch by default (k_part1_narrow<18, 18432, VEC2, ASM> and the four ring lengths): they
# must be present; every OTHER instantiation with ASM = true found in the assembly is audited as well (discovered, not listed).
EXPECTED = ["k_part1_narrowILi18ELi18432ELb1ELb1ELi0ELi0ELi0EEE", "k_part1_narrowILi18ELi18432ELb1ELb1ELi0ELi1ELi0EEE", "k_part1_narrowILi14ELi14336ELb1ELb1ELi0ELi2ELi0EEE",
            "k_part1_narrowILi12ELi12288ELb1ELb1ELi0ELi0ELi1EEE", "k_part2_wideILi12ELb1EEE", "k_part2_ringILi4ELi32ELb1ELi0EEE",
            "k_part2_ringILi4ELi64ELb1ELi0EEE", "k_part2_ringILi4ELi128ELb1ELi0EEE", "k_part2_ringILi4ELi256ELb1ELi0EEE",
            "k_agg_scatter_rsvIiLb1ELb1EEE", "k_agg_scatter_rsvIiLb0ELb1EEE", "k_agg_scatter_rsvIxLb1ELb1EEE", "k_agg_scatter_rsvIxLb0ELb1EEE"]
# template <int RPT, int WIN, bool VEC2, bool ASM, int DIAG, int DICT, int WIDE> k_part1_narrow / template <int RPT, int R, bool ASM, int DIAG> k_part2_ring
# template <typename KeyT, bool VAL_F64, bool ASM> k_agg_scatter_rsv; template <int RPT, bool ASM> k_part2_wide (csrc/golap_part_wide.h)
# (the whole template argument list is part of the key, up to the EEE that closes it: a new template parameter shows up as a new key)
ASM_FORMS = re.compile(r"^_ZN5golap\d+(k_part1_narrowILi\d+ELi\d+ELb[01]ELb1E(?:Li\d+E)+EE|k_part2_ringILi\d+ELi\d+ELb1E(?:Li\d+E)+EE|"
                       r"k_agg_scatter_rsvI[a-z]Lb[01]ELb1E(?:L[a-z]\d+E)*EE|k_part2_wideILi\d+ELb1E(?:L[a-z]\d+E)*EE)\w*:", re.M)
# The audit reads ONE compiler's output: the hand-counted waits and the fixed-register scheme were validated (on the GPU, row by row:
# tests/test_join_gpu.py::test_hidden_forms_match_plain_forms) with exactly this hipcc.  A different compiler must be re-validated
# there before this constant is changed.
# the compiler the hidden forms were audited with: csrc/golap_variants.h is the one place that names it (a library built by any other
# compiler launches the plain forms by itself, tests/test_variants_cpu.py)
VALIDATED_HIPCC = ".".join(re.search(r"#define GOLAP_HIDDEN_AUDITED_HIP_" + k + r" (\d+)", open(os.path.join(CSRC, "golap_variants.h")).read()).group(1)
                           for k in ("MAJOR", "MINOR", "PATCH"))


def _compile(tmp_path_factory, extra=()):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-inline-asm", "-Wno-pass-failed", *extra]   # the Makefile's flags
    d, text = tmp_path_factory.mktemp("isa"), ""
    for src in ("golap_join", "golap_ops"):
        subprocess.check_call([HIPCC, *flags, "-S", "--cuda-device-only", os.path.join(CSRC, src + ".hip"), "-o", str(d / (src + ".s"))],
                              stderr=subprocess.DEVNULL)
        text += open(d / (src + ".s")).read()
    return text


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    return _compile(tmp_path_factory)


def _discover(asm):
    return sorted(set(ASM_FORMS.findall(asm)))


def test_compiler_is_the_validated_one():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    ver = subprocess.check_output([HIPCC, "--version"], text=True)
    m = re.search(r"HIP version:\s*([\d.]+)", ver)
    assert m and m.group(1).startswith(VALIDATED_HIPCC), (
        f"hipcc {m.group(1) if m else ver!r} is not the compiler the hidden-memory kernels were validated with ({VALIDATED_HIPCC}): run "
        "tests/test_join_gpu.py::test_hidden_forms_match_plain_forms on a GPU with this compiler's build, then update VALIDATED_HIPCC")


def test_every_hidden_memory_instantiation_is_discovered(asm):
    found = _discover(asm)
    for k in EXPECTED:
        assert k in found, f"{k} (a default launch form) is missing from the assembly"
    # the shipped library must not contain the timing-only diagnostic forms (DIAG != 0: wrong results by design)
    diag = re.findall(r"^_ZN5golap\d+(k_part1_narrowILi\d+ELi\d+ELb[01]ELb[01]ELi[1-9]\d*ELi\d+ELi\d+EEE|k_part2_ringILi\d+ELi\d+ELb[01]ELi[1-9]\d*EEE)\w*:", asm, re.M)
    assert not diag, f"diagnostic kernel forms in the default build: {sorted(set(diag))}"


def _body(asm, key):
    m = re.search(r"^(_ZN5golap\d+" + re.escape(key) + r"\w*):\s*;", asm, re.M)
    assert m, f"kernel {key} not found in the assembly"
    name = m.group(1)
    end = asm.index(".amdhsa_kernel " + name)
    meta = asm[asm.index(f".name:           {name}") if f".name:           {name}" in asm else 0:]
    return name, asm[m.start():end].split("\n")


def _regs(line):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", line):
        out |= set(range(int(a), int(b) + 1))
    out |= {int(x) for x in re.findall(r"\bv(\d+)\b", line)}
    return out


def test_hidden_memory_kernels_have_no_scratch_and_no_stray_waits(asm):
    keys = _discover(asm)
    assert keys
    for key in keys:
        _audit(asm, key)


def test_the_audit_sees_a_planted_violation(asm):
    """the control-flow scan must fail on a compiler-style read of a register with a hidden load in flight (here: planted right
    behind the first prefetch of the tile loop, i.e. in front of the back edge that leads to the counted wait)"""
    key = "k_agg_scatter_rsvIiLb1ELb1EEE"
    name, lines = _body(asm, key)
    loads = [i for i, l in enumerate(lines) if re.search(r"global_load_dwordx4 v\[\d+:\d+\]", l)]
    at = loads[len(loads) // 2]   # the second group of loads is the prefetch inside the loop
    reg = re.search(r"global_load_dwordx4 v\[(\d+):", lines[at]).group(1)
    end = next(i for i in range(at, len(lines)) if "ASMEND" in lines[i])
    bad = lines[:end + 1] + [f"\tv_mov_b32_e32 v1, v{reg}"] + lines[end + 1:]
    start = asm.index("\n".join(lines))
    planted = asm[:start] + "\n".join(bad) + asm[start + len("\n".join(lines)):]
    with pytest.raises(AssertionError, match="hidden load"):
        _audit(planted, key)


def test_the_audit_sees_a_planted_stray_wait(asm):
    """a compiler-style `s_waitcnt vmcnt(0)` in a block of the tile loop (it would drain the prefetch on every tile) must fail the
    audit; the same line in a cold block behind s_endpgm must not"""
    key = "k_part2_wideILi12ELb1EEE"
    name, lines = _body(asm, key)
    at = next(i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l) and "in Loop" in l)
    bad = lines[:at + 1] + ["\ts_waitcnt vmcnt(0)"] + lines[at + 1:]
    start = asm.index("\n".join(lines))
    planted = asm[:start] + "\n".join(bad) + asm[start + len("\n".join(lines)):]
    with pytest.raises(AssertionError, match="compiler-inserted vmcnt waits"):
        _audit(planted, key)
    end = max(i for i, l in enumerate(lines) if "s_endpgm" in l)
    cold = lines[:end + 1] + [".LBB999_999:", "\ts_waitcnt vmcnt(0)", "\ts_endpgm"] + lines[end + 1:]
    _audit(asm[:start] + "\n".join(cold) + asm[start + len("\n".join(lines)):], key)


def _audit(asm, key):
    name, lines = _body(asm, key)
    # (1) no spills, no scratch
    assert not [l for l in lines if "scratch_" in l], f"{key}: scratch instructions (spills) next to hidden loads"
    md = re.search(r"\.name:\s+" + re.escape(name) + r"\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", asm)
    sp = re.search(re.escape(name) + r"\.private_seg_size,\s*(\d+)", asm)
    if sp:
        assert int(sp.group(1)) == 0, f"{key}: private segment {sp.group(1)} bytes"
    # (2) between a hidden load (or returning atomic) and the next hand-counted wait, nothing outside the asm statements touches
    #     its destination registers: a copy placed there (e.g. a phi copy at a control-flow merge) reads data that has not
    #     landed, and the original registers -- free again in the compiler's eyes -- get overwritten by the late data while they
    #     hold something else (seen on MI355X: a store address, i.e. a memory fault).  Checked along the control-flow graph of the
    #     kernel text (labels, s_branch / s_cbranch_*, fall-through; loop back edges included): the set of registers with a load
    #     possibly in flight is propagated to a fixed point, and every compiler-issued instruction is checked against the set that
    #     reaches it on ANY path.
    ins, labels, in_asm = [], {}, False   # ins: (text line number, instruction, inside an asm statement)
    for ln, l in enumerate(lines):
        t = l.split(";")[0].strip()
        if "ASMSTART" in l:
            in_asm = True
        elif "ASMEND" in l:
            in_asm = False
        elif re.match(r"\.LBB\d+_\d+:", t):
            labels[t[:-1]] = len(ins)
        elif t and not t.startswith("."):
            ins.append((ln, t, in_asm))
    hidden, reach, work = set(), {0: frozenset()}, [0]
    load_re = re.compile(r"global_(?:load_dword(?:x[24])?|atomic_add) (v\[(\d+):(\d+)\]|v(\d+)),")
    while work:
        i = work.pop()
        dirty = set(reach[i])
        ln, t, asm_stmt = ins[i]
        if asm_stmt:
            m = load_re.match(t)
            if m:
                r = set(range(int(m.group(2)), int(m.group(3)) + 1)) if m.group(2) else {int(m.group(4))}
                hidden |= r
                dirty |= r
            if t.startswith("s_waitcnt vmcnt"):
                dirty = set()
        else:
            touched = _regs(t) & dirty
            assert not touched, f"{key}: line {ln} `{t}` uses v{sorted(touched)} while a hidden load into them may still be in flight"
        succ = []
        op = t.split()[0]
        if op == "s_branch":
            succ = [labels[t.split()[1]]]
        elif op.startswith("s_cbranch"):
            succ = [labels[t.split()[1]], i + 1]
        elif op not in ("s_endpgm", "s_setpc_b64"):
            succ = [i + 1]
        for j in succ:
            if j < len(ins) and not (j in reach and dirty <= reach[j]):
                reach[j] = frozenset(dirty | reach.get(j, frozenset()))
                work.append(j)
    assert hidden, f"{key}: no hidden loads found (is this still the ASM form?)"
    # (4) a hidden store of more than 8 bytes keeps reading its data registers for two more cycles on gfx940+ (hipcc pads compiler-
    #     visible stores itself but does not look inside asm statements): every such store must carry its own wait states
    for i, l in enumerate(lines):
        if "global_store_dwordx4" in l or "global_store_dwordx3" in l:
            nxt = [x.strip() for x in lines[i + 1:i + 3]]
            assert nxt and nxt[0].startswith("s_nop"), f"{key}: line {i} `{l.strip()}` is not followed by s_nop (store-data hazard)"
    if "k_part2_ring" in key:
        # the ring pass loads into fixed registers outside hipcc's budget (golap_hidden.h, ld8_fixed): hipcc itself must
        # never name them, and the kernel descriptor must cover them
        src = open(os.path.join(CSRC, "golap_hidden.h")).read()
        first = int(re.search(r"P2R_FIXED_FIRST = (\d+)", src).group(1))
        pairs = int(re.search(r"P2R_FIXED_PAIRS = (\d+)", src).group(1))
        fixed = set(range(first, first + 2 * pairs))
        assert hidden == fixed, f"{key}: hidden loads go to {sorted(hidden)}, expected the fixed registers {first}..{first + 2 * pairs - 1}"
        in_asm = False
        for ln, l in enumerate(lines):
            if "ASMSTART" in l:
                in_asm = True
            elif "ASMEND" in l:
                in_asm = False
            elif not in_asm and not l.strip().startswith(";"):
                assert not (_regs(l) & fixed), f"{key}: line {ln} `{l.strip()}`: hipcc allocated a register reserved for hidden loads"
        nv = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(?:.*\n)*?\s+\.amdhsa_next_free_vgpr (\d+)", asm)
        assert nv and int(nv.group(1)) >= first + 2 * pairs, f"{key}: kernel descriptor does not cover the fixed registers"
    # (3) every vmcnt wait in the kernel is one of ours (inside an asm statement), except before the main loop / after it
    #     (which block an instruction belongs to, and whether that block is part of a loop, is what hipcc's own block comments say:
    #     `.LBBn_m:  ; in Loop: Header=...`, `; %bb.k:  ; in Loop: ...`, `; =>This Loop Header`, `; Parent Loop ...` -- cold blocks
    #     placed behind s_endpgm carry no such comment and are not part of the tile loop, whatever stands in the lines above them)
    stray, in_asm, in_loop = [], False, False
    for i, l in enumerate(lines):
        if re.match(r"\s*(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
            nxt = lines[i + 1] if i + 1 < len(lines) else ""
            in_loop = any(w in l for w in ("in Loop", "Loop Header", "Parent Loop")) or (nxt.strip().startswith(";") and "Loop Header" in nxt)
        if "ASMSTART" in l:
            in_asm = True
        elif "ASMEND" in l:
            in_asm = False
        elif "s_waitcnt" in l and "vmcnt" in l and not in_asm and in_loop:
            stray.append((i, l.strip()))
    assert not stray, f"{key}: compiler-inserted vmcnt waits inside the tile loop: {stray[:3]}"
