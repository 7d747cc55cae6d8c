"""The C++ host-side mirror of the reference's simulator interfaces (include/ipcr_hip.hpp) -- header-only over the C
ABI.  Without a GPU: it compiles as C++17 against the header alone.  With one: a native program drives it through the
reference's own engine literals, the streaming form, an emit error and a worker pool (tests/native/engine_contract.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "engine_contract.cpp")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "ipcr_amd")


def test_cxx_host_header_compiles_without_hip(tmp_path):
    """a C++ host needs g++ and the two headers, nothing of HIP: the mirror is plain C++17 over plain C"""
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", INC, SRC])
    hdr = open(os.path.join(INC, "ipcr_hip.hpp")).read()
    for name in ("SimulateBatch", "CompilePanel", "SimulateCompiled", "NewSimulationScratch", "SimulateCompiledWithScratch",
                 "ForEachCompiledProduct"):       # internal/pipeline/sim.go:11-39, every method of the interface family
        assert name in hdr


@pytest.mark.gpu
def test_cxx_host_mirror_runs_the_reference_literals(tmp_path):
    exe = str(tmp_path / "engine_contract")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", INC, SRC, "-o", exe, "-L", LIBDIR, "-lipcr_hip", "-lpthread",
                           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, IPCR_JIT_ASYNC=os.environ.get("IPCR_JIT_ASYNC", "0"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
