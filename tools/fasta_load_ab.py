#!/usr/bin/env python3
"""dev tool (GPU box): load time of a 1 GB FASTA file (80-column lines, page cache) into a resident genome under
environment knobs -- one child process per variant, A/B/A/B.   python3 tools/fasta_load_ab.py VAR=a VAR=b ..."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = "/tmp/ipcr_ab.fa"


def child():
    from ipcr_amd import engine
    for _ in range(4):
        t0 = time.perf_counter()
        g = engine.Genome(1_000_000_000 + (1 << 20), max_records=12)
        g.add_fasta(PATH)
        t1 = time.perf_counter()
        print("  load %.1f ms (%d bases)" % ((t1 - t0) * 1e3, g.total_bases), flush=True)
        g.close()


def make():
    import numpy as np
    from ipcr_amd import engine
    import torch
    n = 125_000_000
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    with open(PATH, "wb") as fh:
        for r in range(8):
            engine.lcg_fill_device(buf.data_ptr(), n, 0x5eed1234, r * n)
            torch.cuda.synchronize()
            seq = buf.cpu().numpy()
            fh.write(b">chr%d synthetic LCG record\n" % (r + 1))
            body = np.empty((n // 80, 81), dtype=np.uint8)
            body[:, :80] = seq[: (n // 80) * 80].reshape(-1, 80)
            body[:, 80] = 10
            fh.write(body.tobytes())
            if n % 80:
                fh.write(seq[(n // 80) * 80:].tobytes() + b"\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    if not os.path.exists(PATH):
        subprocess.check_call([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tools.fasta_load_ab as m; m.make()" % ROOT])
    for rep in range(2):
        for var in sys.argv[1:] or [""]:
            env = dict(os.environ)
            for kv in var.split(","):
                if "=" in kv:
                    k, v = kv.split("=", 1)
                    env[k] = v
            print(var or "(default)", flush=True)
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child"], env=env)
    os.unlink(PATH)
