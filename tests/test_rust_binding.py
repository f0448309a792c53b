"""The Rust binding crate (bindings/rust/golap-hip-sys) cannot be compiled here (no Rust toolchain): this keeps its `#[repr(C)]`
structs and `extern "C"` declarations in step with include/golap.h by parsing both."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "golap.h")).read()
RS = open(os.path.join(ROOT, "bindings", "rust", "golap-hip-sys", "src", "lib.rs")).read()

C2R = {"uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "void*": "*mut c_void",
       "const uint64_t*": "*const u64", "const int32_t*": "*const i32", "float": "f32", "double": "f64"}
CONSTS = {"GOLAP_SHARD_STAGES": "8"}


def c_struct(name):
    body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", HDR, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl or "(" in decl:          # (function pointers are compared by name only)
            m = re.search(r"\(\*(\w+)\)", decl)
            if m:
                fields.append((m.group(1), "fn"))
            continue
        ty, names = decl.rsplit(" ", 1)[0], decl.rsplit(" ", 1)[1]
        for nm in [n.strip() for n in (ty.split(" ", 1)[1] + " " + names if "," in decl else names).split(",")] if "," in decl else [names]:
            base = ty.split(" ")[0] if "," in decl else ty
            for cname, cval in CONSTS.items():
                nm = nm.replace(cname, cval)
            arr2 = re.match(r"(\w+)\[(\d+)\]\[(\d+)\]", nm)
            arr = re.match(r"(\w+)\[(\d+)\]", nm)
            if arr2:
                fields.append((arr2.group(1), f"[[{C2R[base]}; {arr2.group(3)}]; {arr2.group(2)}]"))
            elif arr:
                fields.append((arr.group(1), f"[{C2R[base]}; {arr.group(2)}]"))
            else:
                fields.append((nm, C2R.get(base, base)))
    return fields


def rs_struct(name):
    body = re.search(r"pub struct " + name + r" \{(.*?)\n\}", RS, re.S).group(1)
    fields = []
    for m in re.finditer(r"pub (\w+): ([^,\n]+(?:\n\s+[^,\n]+)*),", body):
        ty = " ".join(m.group(2).split())
        fields.append((m.group(1), "fn" if ty.startswith("Option<unsafe extern") else ty))
    return fields


def test_structs_match_the_header():
    # every #[repr(C)] struct with fields that lib.rs declares (discovered, not listed), field by field
    names = [n for n in re.findall(r"#\[repr\(C\)\]\n(?:#\[derive[^\n]*\]\n)?pub struct (golap_\w+) \{", RS) if n not in ("golap_ctx", "golap_comm")]
    for must in ["golap_engine_config", "golap_join_params", "golap_aggregate_params", "golap_sharded_join_params", "golap_sharded_join_info",
                 "golap_sharded_plan", "golap_join_partitioned_plan", "golap_partition_params"]:
        assert must in names, must
    for s in names:
        assert rs_struct(s) == c_struct(s), s


def test_every_bound_function_exists_with_the_same_arity():
    decls = dict()
    for m in re.finditer(r"^(?:int32_t|int64_t|uint32_t|uint64_t|void|const char\*|golap_cmp_op|golap_join_strategy)\s+(golap_\w+)\((.*?)\);", HDR, re.S | re.M):
        args = re.sub(r"/\*.*?\*/", "", m.group(2), flags=re.S).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    bound = re.findall(r"pub fn (golap_\w+)\((.*?)\)(?: ->|;)", RS, re.S)
    assert len(bound) >= 25
    for name, args in bound:
        assert name in decls, f"{name} is not declared in golap.h"
        n = 0 if not args.strip() else args.count(":")
        assert n == decls[name], f"{name}: {n} arguments in lib.rs, {decls[name]} in golap.h"


def test_abi_version_constant():
    assert re.search(r"#define GOLAP_ABI_VERSION (\d+)", HDR).group(1) == re.search(r"GOLAP_ABI_VERSION: u32 = (\d+)", RS).group(1)
