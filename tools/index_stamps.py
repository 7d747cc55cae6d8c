"""dev tool: summarise IPCR_INDEX_STAMPS files (per-wave start / end of every ipcr_index_filter sweep, 100 MHz ticks)."""
import struct
import sys

import numpy as np


def sweeps(path):
    data = open(path, "rb").read()
    off = 0
    while off + 8 <= len(data):
        (n,) = struct.unpack_from("<Q", data, off)
        off += 8
        a = np.frombuffer(data, dtype="<u8", count=n, offset=off).reshape(-1, 2)
        off += 8 * n
        yield a


def main():
    for path in sys.argv[1:]:
        for i, a in enumerate(sweeps(path)):
            a = a[(a[:, 0] > 0) & (a[:, 1] > 0)]
            if not len(a):
                continue
            t0 = a[:, 0].min()
            start = (a[:, 0] - t0) / 100.0      # us
            end = (a[:, 1] - t0) / 100.0
            span = end.max()
            q = np.percentile(end, [0, 1, 10, 50, 90, 99, 100])
            print("%s sweep %d: %d waves, span %.0f us; starts <= %.0f us; ends (min,1,10,50,90,99,max %%): %s; mean residency %.3f of span"
                  % (path, i, len(a), span, start.max(), " ".join("%.0f" % x for x in q), float(((end - start) / span).mean())))
            # per workgroup (16 waves): when its last wave ends
            if len(a) % 16 == 0:
                wg = end.reshape(-1, 16).max(axis=1)
                print("   per-workgroup end: min %.0f median %.0f max %.0f us" % (wg.min(), np.median(wg), wg.max()))


if __name__ == "__main__":
    main()
