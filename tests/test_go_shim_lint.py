"""The Go side of the drop-in (integration/go: hipengine, hipprobe, appcore_core.patch) cannot be compiled here -- the
image has no Go toolchain -- so it is linted against the C header it binds: every C.ipcr_* function it calls exists in
include/ipcr_hip.h with that many parameters, every C.IPCR_* constant and C.ipcr_* type exists, every struct field it
names is a field of that struct (cgo spells a field called like a Go keyword with a leading underscore: `_type`).
The test fails when the header drifts away from the shim, or the shim from the header."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ipcr_hip.h")
GO_DIR = os.path.join(ROOT, "integration", "go")
GO_KEYWORDS = {"type", "func", "range", "map", "chan", "go", "select", "defer", "var", "package", "import", "interface",
               "fallthrough", "default", "switch", "case", "const", "struct", "goto", "return", "break", "continue", "if",
               "else", "for"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def parse_header():
    h = strip_comments(open(HEADER).read())
    funcs = {}
    for m in re.finditer(r"\b(ipcr_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", h, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        if "(*" in h[max(0, m.start() - 2):m.start() + len(name) + 2]:
            continue
        n = 0 if params in ("", "void") else len(split_args(params))
        funcs[name] = n
    consts = set(re.findall(r"#define\s+(IPCR_[A-Z0-9_]+)", h))
    for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*\w+\s*;", h, flags=re.S):
        consts |= set(re.findall(r"\b(IPCR_[A-Z0-9_]+)\b", m.group(1)))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*(\w*)\s*\{(.*?)\}\s*(\w+)\s*;", h, flags=re.S):
        fields = set()
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in split_args(decl):
                name = re.findall(r"(\w+)\s*(?:\[[^\]]*\])*\s*$", part)
                if name:
                    fields.add(name[0])
        structs[m.group(3)] = fields
    opaque = set(re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", h))
    fnptr = set(re.findall(r"typedef\s+\w+\s*\(\s*\*\s*(\w+)\s*\)", h))
    return funcs, consts, structs, opaque | fnptr


def go_sources():
    out = {}
    for d, _, files in os.walk(GO_DIR):
        for f in files:
            if f.endswith(".go"):
                out[os.path.relpath(os.path.join(d, f), ROOT)] = open(os.path.join(d, f)).read()
    return out


def go_code(src):
    """Go source without comments, string literals and the cgo preamble"""
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)
    return re.sub(r"`[^`]*`", "``", src)


def call_args(code, start):
    """arguments of the call whose '(' is at code[start]"""
    depth, i = 0, start
    while True:
        if code[i] in "([{":
            depth += 1
        elif code[i] in ")]}":
            depth -= 1
            if depth == 0:
                return split_args(code[start + 1:i])
        i += 1


def test_header_parses():
    funcs, consts, structs, opaque = parse_header()
    assert funcs["ipcr_scan_chunk"] == 6 and funcs["ipcr_probe_scratch_products"] == 5 and funcs["ipcr_device_count"] == 0
    assert {"IPCR_OK", "IPCR_MAX_MM", "IPCR_MAX_PRIMER_LEN", "IPCR_ERR_ABORTED"} <= consts
    assert {"start", "end", "type", "fwd_idx", "n_rev_idx"} <= structs["ipcr_product"] and {"found", "strand", "pos", "mm"} == structs["ipcr_probe_hit"]
    assert {"ipcr_panel", "ipcr_scratch", "ipcr_genome", "ipcr_emit_fn"} <= opaque
    from ipcr_amd import _lib
    assert set(funcs) == set(_lib.SYMBOLS), sorted(set(funcs) ^ set(_lib.SYMBOLS))   # the ctypes table and the header agree too


def test_go_shim_uses_only_what_the_header_declares():
    funcs, consts, structs, opaque = parse_header()
    srcs = go_sources()
    assert {os.path.basename(p) for p in srcs} >= {"hipengine.go", "hipprobe.go"}
    used_funcs = set()
    for path, src in srcs.items():
        assert "//go:build hip" in src.splitlines()[0], path
        assert '#include "ipcr_hip.h"' in src, path
        code = go_code(src)
        for m in re.finditer(r"\bC\.(\w+)", code):
            name = m.group(1)
            after = code[m.end():m.end() + 1]
            if name.startswith("ipcr_"):
                if after == "(":
                    assert name in funcs, f"{path}: C.{name}() is not declared in include/ipcr_hip.h"
                    n = len(call_args(code, m.end()))
                    assert n == funcs[name], f"{path}: C.{name} called with {n} arguments, the header declares {funcs[name]}"
                    used_funcs.add(name)
                else:
                    assert name in structs or name in opaque, f"{path}: C.{name} is not a type of include/ipcr_hip.h"
                    if after == "{":   # composite literal: its keys are fields
                        body = code[m.end() + 1:code.index("}", m.end())]
                        for key in re.findall(r"(\w+)\s*:", body):
                            assert key in structs[name], f"{path}: C.{name} has no field {key}"
            elif name.startswith("IPCR_"):
                assert name in consts, f"{path}: C.{name} is not defined in include/ipcr_hip.h"
            else:
                assert name in {"CString", "GoString", "free", "int32_t", "int64_t", "uint32_t", "uint64_t", "uint8_t", "int"}, f"{path}: unexpected C.{name}"
        # field accesses on values of the C structs (variable -> struct, as the sources name them)
        for var, struct in (("cpr", "ipcr_product"), ("cfg", "ipcr_config"), ("ccfg", "ipcr_config"), ("h", "ipcr_probe_hit"), ("cw", "ipcr_chunk_window")):
            if not re.search(r"\b%s\b" % var, code):
                continue
            for fld in re.findall(r"(?<![\w.])%s\.(\w+)" % var, code):
                cname = fld[1:] if fld.startswith("_") else fld
                assert cname in structs[struct], f"{path}: {var}.{fld}: {struct} has no field {cname}"
                assert (cname in GO_KEYWORDS) == fld.startswith("_"), f"{path}: {var}.{fld}: cgo spells a field named like a Go keyword with a leading underscore"
    # the path's entry points are all bound
    assert {"ipcr_panel_create", "ipcr_panel_destroy", "ipcr_scratch_create_on", "ipcr_scratch_destroy", "ipcr_scan_chunk",
            "ipcr_scratch_products", "ipcr_probe_scratch_products", "ipcr_device_count", "ipcr_last_error"} <= used_funcs


def test_go_shim_implements_the_pipeline_interfaces():
    """method sets of internal/pipeline/sim.go:11-39 (names and parameter counts) on both engines"""
    want = {"SimulateBatch": 3, "CompilePanel": 1, "SimulateCompiled": 3, "NewSimulationScratch": 1,
            "SimulateCompiledWithScratch": 4, "ForEachCompiledProduct": 5}
    eng = go_code(open(os.path.join(GO_DIR, "internal", "hipengine", "hipengine.go")).read())
    got = {}
    for m in re.finditer(r"func \(e \*Engine\) (\w+)\(", eng):
        got[m.group(1)] = len(call_args(eng, m.end() - 1))
    for name, n in want.items():
        assert got.get(name) == n, (name, got.get(name))
    prb = go_code(open(os.path.join(GO_DIR, "internal", "hipprobe", "hipprobe.go")).read())
    assert "*hipengine.Engine" in prb                      # the rest of the method set is embedded
    for name in ("ForEachCompiledProduct", "SimulateCompiledWithScratch", "SimulateCompiled", "SimulateBatch"):
        assert re.search(r"func \(e \*Engine\) %s\(" % name, prb), name
    assert re.search(r"func \(v Visitor\) Visit\(p engine\.Product\) \(bool, probeoutput\.AnnotatedProduct, error\)", prb)


def test_patch_applies_to_the_reference(tmp_path):
    ref = "/root/reference"
    if not os.path.isdir(ref) or shutil.which("patch") is None:
        pytest.skip("no reference checkout / no patch(1) here")
    for rel in ("internal/appcore/core.go", "internal/probeapp/app.go"):
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        shutil.copy(os.path.join(ref, rel), tmp_path / rel)
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(GO_DIR, "appcore_core.patch")], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr
