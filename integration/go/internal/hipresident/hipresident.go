//go:build hip

// Package hipresident is the pipeline without the chunk stream: pipeline.ForEachProduct (internal/pipeline/pipeline.go:33-196)
// reads every FASTA file on one goroutine, hands rolling chunks to workers and collects; on an MI355X host that reader
// (~1 Gbases/s) is what a run waits for -- the drop-in workers behind it take 150 Gbases/s.  Here every file goes through the
// device loader (ipcr_genome_add_fasta: 44-48 Gbases/s from the page cache), the resident tiles are swept ONCE, and with
// --chunk-size every record's hits are cut into the same rolling windows and every window is joined as its own
// ForEachCompiledProduct call (ipcr_scan_genome_chunked): the same products, window-local coordinates and "id:start-end" IDs the
// workers would have sent, handed to the same collector logic (chunk-local -> record-global, bounded de-duplication).
//
// Same signature as pipeline.ForEachProduct minus the simulator.  The hook is three lines in cmdutil.RunStream:
//
//	if _, ok := sim.(*hipengine.Engine); ok && !pc.Circular {
//		return hipresident.ForEachProduct(ctx, cfg, seqFiles, pairs, engineCfg, visit)
//	}
//
// Shipped as source (no Go toolchain in the build image); tests/test_go_shim_lint.py checks its C identifiers against
// include/ipcr_hip.h.
package hipresident

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/ipcr_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/ipcr_hip -lipcr_hip -Wl,-rpath,${SRCDIR}/../../third_party/ipcr_hip
#include <stdlib.h>
#include "ipcr_hip.h"
*/
import "C"

import (
	"context"
	"fmt"
	"os"
	"unsafe"

	"ipcr-core/engine"
	"ipcr-core/primer"
	"ipcr/internal/common"
	"ipcr/internal/pipeline"
	"ipcr/internal/runutil"
)

func lastErr() error { return fmt.Errorf("ipcr_hip: %s", C.GoString(C.ipcr_last_error())) }

// ForEachProduct: every product of every file, in the collector's order per file (records in file order, windows in record
// order), coordinates record-global, products found in two overlapping windows once.
func ForEachProduct(ctx context.Context, pc pipeline.Config, seqFiles []string, pairs []primer.Pair, ec engine.Config,
	visit func(engine.Product) error) error {
	ccfg := C.ipcr_config{max_mm: C.int32_t(ec.MaxMM), terminal_window: C.int32_t(ec.TerminalWindow),
		min_len: C.int32_t(ec.MinLen), max_len: C.int32_t(ec.MaxLen), hit_cap: C.int32_t(ec.HitCap), seed_len: C.int32_t(ec.SeedLen)}
	if ec.Circular {
		ccfg.circular = 1
	}
	cps := make([]C.ipcr_pair, len(pairs))
	var frees []unsafe.Pointer
	defer func() {
		for _, p := range frees {
			C.free(p)
		}
	}()
	for i, p := range pairs {
		id, f, r := C.CString(p.ID), C.CString(p.Forward), C.CString(p.Reverse)
		frees = append(frees, unsafe.Pointer(id), unsafe.Pointer(f), unsafe.Pointer(r))
		cps[i] = C.ipcr_pair{id: id, forward: f, reverse: r, min_product: C.int32_t(p.MinProduct), max_product: C.int32_t(p.MaxProduct)}
	}
	var panel *C.ipcr_panel
	var first *C.ipcr_pair
	if len(cps) > 0 {
		first = &cps[0]
	}
	if C.ipcr_panel_create(&ccfg, first, C.int32_t(len(cps)), &panel) != C.IPCR_OK {
		return lastErr()
	}
	defer C.ipcr_panel_destroy(panel)
	var scratch *C.ipcr_scratch
	if C.ipcr_scratch_create(panel, &scratch) != C.IPCR_OK {
		return lastErr()
	}
	defer C.ipcr_scratch_destroy(scratch)

	seen := runutil.NewLRUSet[pipeline.Key](pc.DedupCap)
	for _, fa := range seqFiles {
		if err := ctx.Err(); err != nil {
			return err
		}
		if err := scanFile(fa, pc, pairs, ec, panel, scratch, seen, visit); err != nil {
			return err
		}
	}
	return nil
}

func scanFile(fa string, pc pipeline.Config, pairs []primer.Pair, ec engine.Config, panel *C.ipcr_panel, scratch *C.ipcr_scratch,
	seen *runutil.LRUSet[pipeline.Key], visit func(engine.Product) error) error {
	capacity := uint64(1 << 28) // stdin, or a file whose size is not known
	if st, err := os.Stat(fa); err == nil && st.Size() > 0 {
		capacity = uint64(st.Size())
		if len(fa) > 3 && fa[len(fa)-3:] == ".gz" {
			capacity *= 8
		}
	}
	var g *C.ipcr_genome
	if C.ipcr_genome_create(C.uint64_t(capacity+(1<<20)), 1<<16, &g) != C.IPCR_OK {
		return lastErr()
	}
	defer C.ipcr_genome_destroy(g)
	cpath := C.CString(fa)
	defer C.free(unsafe.Pointer(cpath))
	var added C.uint32_t
	if C.ipcr_genome_add_fasta(g, cpath, &added, nil, 0, nil) != C.IPCR_OK {
		return lastErr()
	}
	// one sweep; with --chunk-size every rolling window is its own ForEachCompiledProduct call
	if C.ipcr_scan_genome_chunked(panel, scratch, g, C.int64_t(pc.ChunkSize), C.int64_t(pc.Overlap), nil, nil) != C.IPCR_OK {
		return lastErr() // IPCR_ERR_UNSUPPORTED (a capped scan that ran in segments): the caller falls back to pipeline.ForEachProduct
	}
	var prods *C.ipcr_product
	var n C.int64_t
	if C.ipcr_scratch_products(scratch, &prods, &n) != C.IPCR_OK {
		return lastErr()
	}
	var wins *C.ipcr_chunk_window
	var nw C.int64_t
	if C.ipcr_scratch_chunk_windows(scratch, &wins, &nw) != C.IPCR_OK {
		return lastErr()
	}
	windows := unsafe.Slice(wins, int(nw))
	for _, cpr := range unsafe.Slice(prods, int(n)) {
		cw := windows[int(cpr.record)]
		pair := pairs[int(cpr.pair)]
		// what the worker would have sent: the window's ID and window-local coordinates ...
		id := C.GoString(C.ipcr_genome_record_id(g, cw.record))
		seqID := id
		if cw.plain == 0 {
			seqID = fmt.Sprintf("%s:%d-%d", id, uint64(cw.start), uint64(cw.end))
		}
		pr := engine.Product{ExperimentID: pair.ID, SequenceID: seqID, SourceFile: fa,
			Start: int(cpr.start), End: int(cpr.end), Length: int(cpr.length),
			Type: "forward", FwdMM: int(cpr.fwd_mm), RevMM: int(cpr.rev_mm), FwdPrimer: pair.Forward, RevPrimer: pair.Reverse}
		if cpr._type == 1 {
			pr.Type, pr.FwdPrimer, pr.RevPrimer = "revcomp", pair.Reverse, pair.Forward
		}
		pr.FwdMismatchIdx = idx(cpr.fwd_idx[:], int(cpr.n_fwd_idx))
		pr.RevMismatchIdx = idx(cpr.rev_idx[:], int(cpr.n_rev_idx))
		if pc.NeedSeq || ec.NeedSites { // the window's bases, read back from the resident tiles (pipeline.go:80-89, engine.go:175-183)
			amp := make([]byte, pr.End-pr.Start)
			if len(amp) > 0 && C.ipcr_genome_read(g, cw.record, C.uint64_t(uint64(cw.start)+uint64(pr.Start)), (*C.uint8_t)(unsafe.Pointer(&amp[0])), C.uint64_t(len(amp))) != C.IPCR_OK {
				return lastErr()
			}
			if pc.NeedSeq {
				pr.Seq = string(amp)
			}
			if ec.NeedSites {
				if fl := len(pr.FwdPrimer); fl <= len(amp) {
					pr.FwdSite = string(amp[:fl])
				}
				if rl := len(pr.RevPrimer); rl <= len(amp) {
					pr.RevSite = string(primer.RevComp(amp[len(amp)-rl:]))
				}
			}
		}
		// ... and the collector's part (pipeline.go:127-161)
		base, off, ok := common.SplitChunkSuffix(pr.SequenceID)
		if !ok {
			base, off = pr.SequenceID, 0
		}
		gs, ge := pr.Start+off, pr.End+off
		if seen.Add(pipeline.Key{Base: base, File: fa, Start: gs, End: ge, Type: pr.Type, Exp: pr.ExperimentID}) {
			continue
		}
		if ok {
			pr.SequenceID, pr.Start, pr.End = base, gs, ge
		}
		if err := visit(pr); err != nil {
			return err
		}
	}
	return nil
}

func idx(a []C.uint8_t, n int) []int {
	if n == 0 {
		return nil
	}
	out := make([]int, n)
	for i := range out {
		out[i] = int(a[i])
	}
	return out
}
