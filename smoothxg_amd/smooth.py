"""ctypes binding of include/sxg_smooth.h (libsxgsmooth.so): the host-side rows around the POA --
sequence collection / padding / dedup (A2-A4), block-graph normalisation (A9, A10), lacing and GFA
I/O.  The POA provider is a C function pointer: `gpu_provider(engine)` hands the
library `sxg_poa_batch_run` of libsxgpoa.so and the engine handle, so a smoothing iteration runs
collect -> one batched GPU call -> lace without Python in the loop."""
import ctypes as C
import os

from . import build as _build
from . import poa as _poa

_lib = None

RUN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_poa.BatchIn), C.POINTER(_poa.BatchOut))
FREE_FN = C.CFUNCTYPE(None, C.POINTER(_poa.BatchOut))


class SmoothParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("poa_m", C.c_int32), ("poa_n", C.c_int32), ("poa_g", C.c_int32), ("poa_e", C.c_int32),
                ("poa_q", C.c_int32), ("poa_c", C.c_int32), ("local_alignment", C.c_int32),
                ("poa_padding_fraction", C.c_float), ("max_block_depth_for_padding_more", C.c_uint64),
                ("add_consensus", C.c_int32), ("consensus_base_name", C.c_char_p),
                ("adaptive_poa_params", C.c_int32), ("kmer_size", C.c_int32), ("use_abpoa", C.c_int32),
                ("abpoa_band_local", C.c_int32), ("poa_spoa_order", C.c_int32)]


EXPORTS = ["sxg_smooth_abi_version", "sxg_smooth_default_params", "sxg_smooth_last_error", "sxg_smooth_free", "sxg_graph_from_gfa",
           "sxg_graph_free", "sxg_graph_node_count", "sxg_graph_path_count", "sxg_blockset_by_path_windows",
           "sxg_blockset_free", "sxg_blockset_size", "sxg_block_collect_text", "sxg_block_graph_gfa",
           "sxg_smooth_gfa", "sxg_adaptive_poa_scores", "sxg_block_identity_threshold",
           "sxg_block_maf_rows", "sxg_block_maf", "sxg_blockset_from_ranges", "sxg_blockset_block_size",
           "sxg_blockset_block_ranges", "sxg_blockset_smoothable", "sxg_blockset_break", "sxg_blockset_break_ex", "sxg_merge_default_params", "sxg_smooth_maf_gfa"]


class MergeParams(C.Structure):
    """sxg_merge_params: -M / -J / -N of smoothxg (src/main.cpp:282,297-298)."""
    _fields_ = [("merge_blocks", C.c_int32), ("contiguous_path_jaccard", C.c_double), ("preserve_unmerged_consensus", C.c_int32),
                ("max_merged_groups_in_memory", C.c_uint64), ("maf_header", C.c_char_p)]


class PathRange(C.Structure):
    """path_range_t (src/blocks.hpp:29-33): steps [step_begin, step_end) of path `path`, `length` bases."""
    _fields_ = [("path", C.c_int64), ("step_begin", C.c_int64), ("step_end", C.c_int64), ("length", C.c_int64)]


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    so = _build.SMOOTH_SO
    if not os.path.exists(so):
        _build.build_smooth()
    L = C.CDLL(so)
    vp = C.c_void_p
    L.sxg_smooth_last_error.restype = C.c_char_p
    L.sxg_smooth_free.argtypes = [vp]
    L.sxg_smooth_default_params.argtypes = [C.POINTER(SmoothParams)]
    L.sxg_graph_from_gfa.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.sxg_graph_free.argtypes = [vp]
    L.sxg_graph_node_count.restype = C.c_int64
    L.sxg_graph_node_count.argtypes = [vp]
    L.sxg_graph_path_count.restype = C.c_int64
    L.sxg_graph_path_count.argtypes = [vp]
    L.sxg_blockset_by_path_windows.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.sxg_blockset_from_ranges.argtypes = [vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(PathRange), C.POINTER(vp)]
    L.sxg_blockset_block_size.restype = C.c_int64
    L.sxg_blockset_block_size.argtypes = [vp, C.c_int64]
    L.sxg_blockset_block_ranges.argtypes = [vp, C.c_int64, C.POINTER(PathRange)]
    L.sxg_blockset_smoothable.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.sxg_blockset_break.argtypes = [vp, vp, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.sxg_blockset_break_ex.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.sxg_blockset_free.argtypes = [vp]
    L.sxg_blockset_size.restype = C.c_int64
    L.sxg_blockset_size.argtypes = [vp]
    L.sxg_block_collect_text.argtypes = [vp, vp, C.c_int64, C.POINTER(SmoothParams), C.POINTER(vp)]
    L.sxg_block_graph_gfa.argtypes = [vp, vp, C.c_int64, C.POINTER(SmoothParams), vp, vp, vp, C.POINTER(vp)]
    L.sxg_smooth_gfa.argtypes = [vp, vp, C.POINTER(SmoothParams), vp, vp, vp, C.POINTER(vp)]
    L.sxg_merge_default_params.argtypes = [C.POINTER(MergeParams)]
    L.sxg_smooth_maf_gfa.argtypes = [vp, vp, C.POINTER(SmoothParams), C.POINTER(MergeParams), vp, vp, vp, C.POINTER(vp), C.POINTER(vp),
                                     C.POINTER(C.c_int64)]
    L.sxg_block_maf_rows.argtypes = [vp, vp, C.c_int64, C.POINTER(SmoothParams), vp, vp, vp, C.POINTER(vp)]
    L.sxg_block_maf.argtypes = [vp, vp, C.c_int64, C.POINTER(SmoothParams), vp, vp, vp, C.POINTER(vp)]
    L.sxg_adaptive_poa_scores.restype = None
    L.sxg_adaptive_poa_scores.argtypes = [C.c_float, C.POINTER(C.c_int32 * 6), C.POINTER(C.c_int32 * 6)]
    L.sxg_block_identity_threshold.argtypes = [vp, vp, C.c_int64, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    _lib = L
    return L


class SmoothError(RuntimeError):
    pass


def default_params(**kw):
    p = SmoothParams()
    load_library().sxg_smooth_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def adaptive_poa_scores(est_identity_threshold, set_scores=(1, 4, 6, 2, 26, 1)):
    """A14: the score tier (m, n, g, e, q, c; CLI convention) of src/smooth.cpp:2032-2069."""
    a, o = (C.c_int32 * 6)(*set_scores), (C.c_int32 * 6)()
    load_library().sxg_adaptive_poa_scores(est_identity_threshold, C.byref(a), C.byref(o))
    return tuple(o)


def gpu_provider(engine, sharded=False):
    """(run, free, ctx) backed by the GPU engine: raw C entry points of libsxgpoa.so.  sharded=True: the multi-GPU
    entry (sxg_poa_batch_run_sharded over the engine's communicator): every rank runs the same iteration, rank 0 laces."""
    L = engine.lib
    run = C.cast(L.sxg_poa_batch_run_sharded if sharded else L.sxg_poa_batch_run, C.c_void_p)
    fre = C.cast(L.sxg_poa_batch_free, C.c_void_p)
    return run, fre, engine.h


class Smoother:
    """An input GFA + a blockset; collect / block graph / full iteration through the C ABI."""

    def __init__(self, gfa_text, target_bp=None, blocks=None, discover=None):
        """discover: block discovery as smoothxg does it -- a dict with target_poa_length and n_haps (and optionally
        max_path_jump, max_edge_jump, max_poa_length, repeats): smoothable_blocks then the cutting half of break_blocks
        (repeat-aware cut lengths with the reference's defaults unless repeats=None).
        blocks: the caller's own blockset -- a list of blocks, each a list of (path, step_begin, step_end)
        or (path, step_begin, step_end, length) in alignment order (sxg_blockset_from_ranges); otherwise the
        demo partition into path windows of target_bp."""
        self.L = load_library()
        data = gfa_text.encode() if isinstance(gfa_text, str) else gfa_text
        g = C.c_void_p()
        if self.L.sxg_graph_from_gfa(data, len(data), C.byref(g)):
            raise SmoothError(self.L.sxg_smooth_last_error().decode())
        self.g = g
        b = C.c_void_p()
        if discover is not None:
            tl = int(discover["target_poa_length"])
            raw = C.c_void_p()
            rc = self.L.sxg_blockset_smoothable(g, int(discover.get("max_block_weight", tl * int(discover["n_haps"]))), tl,
                                                int(discover.get("max_path_jump", 100)), int(discover.get("max_edge_jump", 0)), 1,
                                                C.byref(raw))
            if not rc:
                rep = discover.get("repeats", (1000, 20000, 5, 50))   # (min_copy_length, max_copy_length, min_autocorr_z, autocorr_stride) or None
                if rep is None:
                    rc = self.L.sxg_blockset_break_ex(g, raw, int(discover.get("max_poa_length", 2 * tl)), 0, 1000, 20000, 5.0, 50, 1, C.byref(b))
                else:
                    rc = self.L.sxg_blockset_break_ex(g, raw, int(discover.get("max_poa_length", 2 * tl)), 1, int(rep[0]), int(rep[1]),
                                                      float(rep[2]), int(rep[3]), 1, C.byref(b))
                self.L.sxg_blockset_free(raw)
        elif blocks is not None:
            flat = [r for blk in blocks for r in blk]
            arr = (PathRange * max(len(flat), 1))(*[PathRange(r[0], r[1], r[2], r[3] if len(r) > 3 else 0) for r in flat])
            off = (C.c_int64 * (len(blocks) + 1))()
            for k, blk in enumerate(blocks):
                off[k + 1] = off[k] + len(blk)
            rc = self.L.sxg_blockset_from_ranges(g, len(blocks), off, arr, C.byref(b))
        else:
            rc = self.L.sxg_blockset_by_path_windows(g, target_bp, C.byref(b))
        if rc:
            msg = self.L.sxg_smooth_last_error().decode()
            self.L.sxg_graph_free(g)
            self.g = None
            raise SmoothError(msg)
        self.b = b

    def block_ranges(self, block_id):
        """The ranges of a block as (path, step_begin, step_end, length) tuples."""
        n = self.L.sxg_blockset_block_size(self.b, block_id)
        if n < 0:
            raise SmoothError("no such block")
        arr = (PathRange * max(n, 1))()
        if self.L.sxg_blockset_block_ranges(self.b, block_id, arr):
            raise SmoothError(self.L.sxg_smooth_last_error().decode())
        return [(arr[k].path, arr[k].step_begin, arr[k].step_end, arr[k].length) for k in range(n)]

    def close(self):
        if getattr(self, "b", None):
            self.L.sxg_blockset_free(self.b)
            self.b = None
        if getattr(self, "g", None):
            self.L.sxg_graph_free(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_blocks(self):
        return self.L.sxg_blockset_size(self.b)

    def _text(self, rc, out):
        if rc:
            raise SmoothError(self.L.sxg_smooth_last_error().decode())
        try:
            return C.string_at(out).decode()
        finally:
            self.L.sxg_smooth_free(out)

    def identity_threshold(self, block_id, kmer_size=17):
        """A14: (threshold, sequences used); the threshold only counts when more than one was used."""
        thr, n = C.c_float(), C.c_int32()
        if self.L.sxg_block_identity_threshold(self.g, self.b, block_id, kmer_size, C.byref(thr), C.byref(n)):
            raise SmoothError(self.L.sxg_smooth_last_error().decode())
        return thr.value, n.value

    def collect_text(self, block_id, params):
        out = C.c_void_p()
        return self._text(self.L.sxg_block_collect_text(self.g, self.b, block_id, C.byref(params), C.byref(out)), out)

    def block_graph_gfa(self, block_id, params, provider):
        run, fre, ctx = provider
        out = C.c_void_p()
        return self._text(self.L.sxg_block_graph_gfa(self.g, self.b, block_id, C.byref(params), run, fre, ctx, C.byref(out)), out)

    def block_maf_rows(self, block_id, params, provider):
        run, fre, ctx = provider
        out = C.c_void_p()
        return self._text(self.L.sxg_block_maf_rows(self.g, self.b, block_id, C.byref(params), run, fre, ctx, C.byref(out)), out)

    def block_maf(self, block_id, params, provider):
        run, fre, ctx = provider
        out = C.c_void_p()
        return self._text(self.L.sxg_block_maf(self.g, self.b, block_id, C.byref(params), run, fre, ctx, C.byref(out)), out)

    def smooth_maf_gfa(self, params, provider, merge_blocks=False, jaccard=1.0, preserve_unmerged=False, max_groups=50, header=None):
        """The iteration with the in-order MAF consumer (block merging, flips): -> (GFA text, MAF text, flipped blocks)."""
        run, fre, ctx = provider
        mp = MergeParams()
        self.L.sxg_merge_default_params(C.byref(mp))
        mp.merge_blocks, mp.contiguous_path_jaccard, mp.preserve_unmerged_consensus = int(merge_blocks), jaccard, int(preserve_unmerged)
        mp.max_merged_groups_in_memory = max_groups
        mp.maf_header = header.encode() if header is not None else None
        gfa, maf, nf = C.c_void_p(), C.c_void_p(), C.c_int64()
        rc = self.L.sxg_smooth_maf_gfa(self.g, self.b, C.byref(params), C.byref(mp), run, fre, ctx, C.byref(gfa), C.byref(maf), C.byref(nf))
        if rc:
            raise SmoothError(self.L.sxg_smooth_last_error().decode())
        try:
            return C.string_at(gfa).decode(), C.string_at(maf).decode(), nf.value
        finally:
            self.L.sxg_smooth_free(gfa)
            self.L.sxg_smooth_free(maf)

    def smooth_gfa(self, params, provider):
        """One smoothing iteration -> GFA text; None on a rank of a multi-GPU provider that does not lace."""
        run, fre, ctx = provider
        out = C.c_void_p()
        rc = self.L.sxg_smooth_gfa(self.g, self.b, C.byref(params), run, fre, ctx, C.byref(out))
        if rc == 1:   # SXG_NOT_ROOT
            return None
        return self._text(rc, out)
