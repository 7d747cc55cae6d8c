"""Build check for the run-time specialised filter: emit its HIP source for the benchmark
panel (config C2) and an IUPAC k=3 panel (config C3) and compile both for gfx950 with hipcc."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from ipcr_amd import engine, primer  # noqa: E402
from ipcr_amd.workloads import c2_pairs, c3_pairs  # noqa: E402


def main() -> None:
    cases = [("c2", engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12), c2_pairs()),
             ("c3", engine.Config(MaxMM=3, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12, Circular=True), c3_pairs())]
    outdir = os.path.join(ROOT, "ipcr_amd", "csrc", "build")
    os.makedirs(outdir, exist_ok=True)
    for name, cfg, pairs in cases:
        cp = engine.New(cfg).CompilePanel(pairs)
        for mode in (0, 1):
            src = cp.filter_source(mode)
            assert src, f"{name}: panel unexpectedly not specialisable"
            path = os.path.join(outdir, f"filter_{name}_m{mode}.hip")
            with open(path, "w") as f:
                f.write(src)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-c", path,
                                   "-o", path[:-4] + ".o"])
        cp.close()
    # the hand-over relies on each 16-byte half of a published hit record being ONE store instruction (each half
    # carries the scan's tag): check the ISA of the C2 kernel for the dwordx4 pairs of hits[] and publish()
    path = os.path.join(outdir, "filter_c2_m0.hip")
    asm = subprocess.check_output(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only",
                                   "-o", "-", path], stderr=subprocess.DEVNULL).decode()
    lines = asm.splitlines()
    pairs16 = sum(1 for a, b in zip(lines, lines[1:]) if "global_store_dwordx4" in a and "global_store_dwordx4" in b
                  and b.rstrip().endswith("offset:16"))
    assert pairs16 >= 4, f"hit records are no longer stored as two dwordx4 halves ({pairs16} pairs found)"
    # seed-index filter of a large panel (config C4)
    from ipcr_amd.workloads import c4_pairs
    cp = engine.New(engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)).CompilePanel(c4_pairs(64))
    for mode in (2, 3):
        src = cp.filter_source(mode)
        assert "ipcr_index_filter" in src
        path = os.path.join(outdir, f"index_c4_m{mode - 2}.hip")
        with open(path, "w") as f:
            f.write(src)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-c", path, "-o", path[:-4] + ".o"])
    cp.close()
    print("jit_check: specialised filter sources compile for gfx950")


if __name__ == "__main__":
    main()
