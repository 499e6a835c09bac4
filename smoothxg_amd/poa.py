"""ctypes binding of the C ABI in include/sxg_poa.h (libsxgpoa.so, HIP/gfx950).

This is the host-side mirror of the reference's per-block POA call sequence
(src/smooth.cpp:752-786): `PoaEngine.run_blocks` takes what smooth_spoa has after its
dedup step (sequences, weights, the six scores in spoa's sign convention, local/global) and
returns what build_odgi_SPOA consumes (nodes, edges, per-sequence node paths, consensus).
There is NO CPU fallback: a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MODE_LOCAL, MODE_GLOBAL = 0, 1


class Params(C.Structure):
    _fields_ = [("m", C.c_int8), ("n", C.c_int8), ("g", C.c_int8), ("e", C.c_int8),
                ("q", C.c_int8), ("c", C.c_int8), ("mode", C.c_uint8), ("banded", C.c_uint8)]


def params_from_cli(m=1, n=4, g=6, e=2, q=26, c=1, local=True):
    """smoothxg CLI values (positive penalties, src/main.cpp:322-361) -> spoa convention
    exactly as src/smooth.cpp:2098-2106 negates them."""
    return Params(m, -n, -g, -e, -q, -c, MODE_LOCAL if local else MODE_GLOBAL, 0)


_i32p, _i64p, _u8p, _u32p, _u64p = (C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8),
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint64))


class BatchIn(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("blk_off", _i32p), ("seq_off", _i64p), ("bases", _u8p),
                ("weights", _u32p), ("params", C.POINTER(Params)), ("per_block_params", C.c_int32),
                ("want_consensus", C.c_int32), ("want_msa", C.c_int32),
                ("want_block_graph", C.c_int32), ("bg_consensus_visited_only", C.c_int32), ("bg_trim", _i32p)]


class BatchOut(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("n_seqs", C.c_int64), ("status", _i32p), ("node_off", _i64p),
                ("node_code", _u8p), ("node_rank", _i32p), ("node_group", _i32p), ("edge_off", _i64p),
                ("edge_tail", _i32p), ("edge_head", _i32p), ("edge_weight", _u32p),
                ("seq_path_nodes", _i32p), ("score", _i32p), ("cells", _u64p), ("cons_off", _i64p),
                ("cons_nodes", _i32p), ("msa_off", _i64p), ("msa_cols", _i32p), ("msa", C.c_void_p),
                ("bg_node_off", _i64p), ("bg_node_len", _i32p), ("bg_node_outdeg", _i32p), ("bg_node_indeg", _u8p),
                ("bg_seq_off", _i64p), ("bg_seq", C.c_void_p), ("bg_edge_off", _i64p), ("bg_edge_to", _i32p),
                ("bg_step_off", _i64p), ("bg_steps", _i32p), ("bg_cons_off", _i64p), ("bg_cons_steps", _i32p),
                ("block_cycles", _u64p), ("_owner", C.c_void_p)]


class AlignIn(C.Structure):
    _fields_ = [("n", C.c_int32), ("row_off", _i64p), ("row_code", _u8p), ("row_sink", _u8p),
                ("pred_off", _i64p), ("preds", _i32p), ("seq_off", _i64p), ("bases", _u8p),
                ("params", C.POINTER(Params)), ("per_problem_params", C.c_int32)]


class AlignOut(C.Structure):
    _fields_ = [("n", C.c_int32), ("status", _i32p), ("score", _i32p), ("pair_off", _i64p),
                ("pair_row", _i32p), ("pair_pos", _i32p), ("_owner", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("cells", C.c_uint64), ("dp_launches", C.c_uint64),
                ("algo_bytes", C.c_uint64), ("n_slots", C.c_int32), ("retries", C.c_int32),
                ("device_bytes", C.c_uint64), ("dom_kernel_ms", C.c_double), ("dom_cells", C.c_uint64),
                ("dom_algo_bytes", C.c_uint64), ("dom_threads", C.c_int32), ("dom_cols_per_lane", C.c_int32),
                ("dom_row_mode", C.c_int32), ("dom_clock_mhz", C.c_int32), ("bg_ms", C.c_double)]


class DeviceView(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("n_seqs", C.c_int64), ("n_bases", C.c_int64),
                ("status", C.c_void_p), ("n_nodes", C.c_void_p), ("n_edges", C.c_void_p),
                ("node_code", C.c_void_p), ("node_rank", C.c_void_p), ("node_group", C.c_void_p),
                ("edge_tail", C.c_void_p), ("edge_head", C.c_void_p), ("edge_weight", C.c_void_p),
                ("seq_path_nodes", C.c_void_p), ("score", C.c_void_p)]


EXPORTS = ["sxg_poa_batch_device_view", "sxg_poa_abi_version", "sxg_poa_device_count", "sxg_poa_last_error", "sxg_poa_create",
           "sxg_poa_destroy", "sxg_poa_batch_run", "sxg_poa_batch_upload", "sxg_poa_batch_execute",
           "sxg_poa_batch_download", "sxg_poa_batch_free", "sxg_poa_align_batch", "sxg_poa_align_free",
           "sxg_poa_get_stats", "sxg_poa_set_memory_budget", "sxg_xxh64", "sxg_poa_comm_unique_id", "sxg_poa_comm_init",
           "sxg_poa_comm_attach", "sxg_poa_comm_destroy", "sxg_poa_batch_run_sharded", "sxg_poa_batch_run_sharded_local",
           "sxg_poa_batch_upload_sharded", "sxg_poa_batch_execute_sharded", "sxg_poa_batch_download_sharded", "sxg_poa_sharded_info",
           "sxg_poa_sharded_timing", "sxg_poa_measure_copy", "sxg_poa_roctx_available"]
COMM_ID_BYTES = 128
NOT_ROOT = 1

_lib = None


def load_library(build_if_missing=True):
    """Loads libsxgpoa.so; raises if it is missing and cannot be built."""
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("SXG_POA_LIB", _build.SO)  # override: A/B builds of the same source
    if not os.path.exists(so):
        if not build_if_missing:
            raise RuntimeError("libsxgpoa.so is missing: run `python -m smoothxg_amd.build`")
        _build.build()
    L = C.CDLL(so)
    vp = C.c_void_p
    L.sxg_poa_last_error.restype = C.c_char_p
    L.sxg_poa_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.sxg_poa_destroy.argtypes = [vp]
    L.sxg_poa_batch_run.argtypes = [vp, C.POINTER(BatchIn), C.POINTER(BatchOut)]
    L.sxg_poa_batch_upload.argtypes = [vp, C.POINTER(BatchIn)]
    L.sxg_poa_batch_execute.argtypes = [vp]
    L.sxg_poa_batch_download.argtypes = [vp, C.POINTER(BatchOut)]
    L.sxg_poa_batch_free.argtypes = [C.POINTER(BatchOut)]
    L.sxg_poa_batch_device_view.argtypes = [vp, C.POINTER(DeviceView)]
    L.sxg_poa_align_batch.argtypes = [vp, C.POINTER(AlignIn), C.POINTER(AlignOut)]
    L.sxg_poa_align_free.argtypes = [C.POINTER(AlignOut)]
    L.sxg_poa_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.sxg_poa_set_memory_budget.argtypes = [vp, C.c_uint64]
    L.sxg_poa_measure_copy.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    L.sxg_poa_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.sxg_poa_comm_init.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int, C.c_int]
    L.sxg_poa_comm_attach.argtypes = [vp, vp, C.c_int, C.c_int]
    L.sxg_poa_comm_destroy.argtypes = [vp]
    L.sxg_poa_comm_destroy.restype = None
    L.sxg_poa_batch_run_sharded.argtypes = [vp, C.POINTER(BatchIn), C.POINTER(BatchOut)]
    L.sxg_poa_batch_run_sharded_local.argtypes = [vp, C.POINTER(BatchIn), C.c_int, C.POINTER(BatchOut)]
    L.sxg_poa_batch_upload_sharded.argtypes = [vp, C.POINTER(BatchIn)]
    L.sxg_poa_batch_execute_sharded.argtypes = [vp]
    L.sxg_poa_batch_download_sharded.argtypes = [vp, C.POINTER(BatchIn), C.POINTER(BatchOut)]
    L.sxg_poa_sharded_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    L.sxg_poa_sharded_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.sxg_xxh64.restype = C.c_uint64
    L.sxg_xxh64.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64]
    _lib = L
    return L


class PoaError(RuntimeError):
    pass


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class BlockResult:
    """POA result of one block (what build_odgi_SPOA reads from spoa::Graph)."""
    __slots__ = ("status", "node_code", "node_rank", "node_group", "edge_tail", "edge_head", "edge_weight",
                 "paths", "scores", "cells", "consensus", "msa", "bg", "device_cycles")


class BlockGraph:
    """Normalised block graph of one block (want_block_graph): node sequences, forward edges (from, to) sorted, one path
    of node ids per dedup'd sequence, the consensus path."""
    __slots__ = ("node_seq", "node_indeg", "edges", "paths", "consensus")

    def gfa(self, names, revs=None, consensus_name=None):
        """GFA text in the convention of sxg_block_graph_gfa (names[i]: names of the duplicates of sequence i,
        revs[i][j]: collected in reverse)."""
        o = ["H\tVN:Z:1.0"]
        o += ["S\t%d\t%s" % (i + 1, s) for i, s in enumerate(self.node_seq)]
        o += ["L\t%d\t+\t%d\t+\t0M" % (a + 1, b + 1) for a, b in self.edges]
        for i, p in enumerate(self.paths):
            for j, nm in enumerate(names[i]):
                st = ["%d-" % (v + 1) for v in p[::-1]] if revs is not None and revs[i][j] else ["%d+" % (v + 1) for v in p]
                o.append("P\t%s\t%s\t*" % (nm, ",".join(st)))
        if consensus_name is not None:
            o.append("P\t%s\t%s\t*" % (consensus_name, ",".join("%d+" % (v + 1) for v in self.consensus)))
        return "\n".join(o) + "\n"


class PoaEngine:
    """One engine = one GPU + one stream (sxg_poa_handle)."""

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.sxg_poa_device_count() <= 0:
            raise PoaError("no HIP device: the blocked-POA engine has no CPU fallback")
        h = C.c_void_p()
        rc = self.lib.sxg_poa_create(device, C.byref(h))
        if rc:
            raise PoaError("sxg_poa_create: %s" % self.lib.sxg_poa_last_error().decode())
        self.h = h
        self._keep = None
        self._shape = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.sxg_poa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, what):
        return PoaError("%s: %s" % (what, self.lib.sxg_poa_last_error().decode()))

    def set_memory_budget(self, nbytes):
        self.lib.sxg_poa_set_memory_budget(self.h, int(nbytes))

    # -- flat batch -----------------------------------------------------------------
    def _mk_in(self, bases, seq_off, blk_off, weights, params, want_consensus, want_msa, block_graph=0, bg_trim=None,
               bg_cons_visited_only=False):
        bases = np.ascontiguousarray(bases, np.uint8)
        seq_off = np.ascontiguousarray(seq_off, np.int64)
        blk_off = np.ascontiguousarray(blk_off, np.int32)
        nb = len(blk_off) - 1
        if isinstance(params, Params):
            parr = (Params * 1)(params)
            per = 0
        else:
            parr = (Params * max(len(params), 1))(*params)
            per = 1
        w = None if weights is None else np.ascontiguousarray(weights, np.uint32)
        bpad = bases if len(bases) else np.zeros(1, np.uint8)
        trim = None if bg_trim is None else np.ascontiguousarray(bg_trim, np.int32)
        bi = BatchIn(nb, _p(blk_off, C.c_int32), _p(seq_off, C.c_int64), _p(bpad, C.c_uint8),
                     _p(w, C.c_uint32) if w is not None else None, parr, per, int(want_consensus),
                     int(want_msa), int(block_graph), int(bool(bg_cons_visited_only)),
                     _p(trim, C.c_int32) if trim is not None else None)
        self._keep = (bpad, seq_off, blk_off, w, parr, trim)
        return bi

    def upload(self, bases, seq_off, blk_off, weights, params, want_consensus=False, want_msa=False, block_graph=0,
               bg_trim=None, bg_cons_visited_only=False):
        """block_graph: 1 = also the normalised block graphs (BlockResult.bg), 2 = ... without the per-base paths,
        3 = ... and without the raw POA graphs (node_* / edge_* / consensus of the results are None)."""
        bi = self._mk_in(bases, seq_off, blk_off, weights, params, want_consensus, want_msa, block_graph, bg_trim,
                         bg_cons_visited_only)
        if self.lib.sxg_poa_batch_upload(self.h, C.byref(bi)):
            raise self._err("sxg_poa_batch_upload")
        self._shape = (np.asarray(blk_off).copy(), np.asarray(seq_off).copy())

    def execute(self, check=True):
        rc = self.lib.sxg_poa_batch_execute(self.h)
        if rc and (check or rc != -4):
            raise self._err("sxg_poa_batch_execute")
        return rc

    def stats(self):
        s = Stats()
        self.lib.sxg_poa_get_stats(self.h, C.byref(s))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_}

    def device_view(self):
        """Raw HBM pointers of the executed batch's results (see sxg_poa_device_view)."""
        v = DeviceView()
        if self.lib.sxg_poa_batch_device_view(self.h, C.byref(v)):
            raise self._err("sxg_poa_batch_device_view")
        return v

    def download(self):
        out = BatchOut()
        if self.lib.sxg_poa_batch_download(self.h, C.byref(out)):
            raise self._err("sxg_poa_batch_download")
        try:
            return self._unpack(out)
        finally:
            self.lib.sxg_poa_batch_free(C.byref(out))

    def _unpack(self, out):
        blk_off, seq_off = self._shape
        nb = out.n_blocks
        ns = int(out.n_seqs)
        status = _arr(out.status, nb, np.int32)
        node_off = _arr(out.node_off, nb + 1, np.int64)
        edge_off = _arr(out.edge_off, nb + 1, np.int64)
        nn, ne = (int(node_off[-1]), int(edge_off[-1])) if nb else (0, 0)
        has_raw = bool(out.node_code) or nn == 0      # (block_graph=3: the raw POA graphs are left out)
        node_code = _arr(out.node_code, nn, np.uint8) if has_raw else None
        node_rank = _arr(out.node_rank, nn, np.int32) if has_raw else None
        node_group = _arr(out.node_group, nn, np.int32) if has_raw else None
        et = _arr(out.edge_tail, ne, np.int32) if has_raw else None
        eh = _arr(out.edge_head, ne, np.int32) if has_raw else None
        ew = _arr(out.edge_weight, ne, np.uint32) if has_raw else None
        nbases = int(seq_off[-1]) if ns else 0
        paths = _arr(out.seq_path_nodes, nbases, np.int32) if out.seq_path_nodes else None
        bg = None
        if out.bg_node_off:
            bno = _arr(out.bg_node_off, nb + 1, np.int64)
            bso = _arr(out.bg_seq_off, nb + 1, np.int64)
            beo = _arr(out.bg_edge_off, nb + 1, np.int64)
            bpo = _arr(out.bg_step_off, ns + 1, np.int64)
            bg = dict(no=bno, so=bso, eo=beo, po=bpo, ln=_arr(out.bg_node_len, int(bno[-1]), np.int32),
                      od=_arr(out.bg_node_outdeg, int(bno[-1]), np.int32), idg=_arr(out.bg_node_indeg, int(bno[-1]), np.uint8),
                      sq=C.string_at(out.bg_seq, int(bso[-1])) if bso[-1] else b"", to=_arr(out.bg_edge_to, int(beo[-1]), np.int32),
                      st=_arr(out.bg_steps, int(bpo[-1]), np.int32))
            if out.bg_cons_off:
                bg["co"] = _arr(out.bg_cons_off, nb + 1, np.int64)
                bg["cs"] = _arr(out.bg_cons_steps, int(bg["co"][-1]), np.int32)
        score = _arr(out.score, ns, np.int32)
        cells = _arr(out.cells, ns, np.uint64)
        cycles = _arr(out.block_cycles, nb, np.uint64) if out.block_cycles else None
        cons_off = _arr(out.cons_off, nb + 1, np.int64) if out.cons_off else None
        cons = _arr(out.cons_nodes, int(cons_off[-1]), np.int32) if cons_off is not None and nb and out.cons_nodes else None
        msa_off = _arr(out.msa_off, nb + 1, np.int64) if out.msa_off else None
        msa_cols = _arr(out.msa_cols, nb, np.int32) if out.msa_cols else None
        msa_raw = None
        if msa_off is not None and nb and msa_off[-1] > 0:
            msa_raw = C.string_at(out.msa, int(msa_off[-1]))
        res = []
        for b in range(nb):
            r = BlockResult()
            r.status = int(status[b])
            a, z = node_off[b], node_off[b + 1]
            r.node_code, r.node_rank, r.node_group = (node_code[a:z], node_rank[a:z], node_group[a:z]) if has_raw else (None, None, None)
            a, z = edge_off[b], edge_off[b + 1]
            r.edge_tail, r.edge_head, r.edge_weight = (et[a:z], eh[a:z], ew[a:z]) if has_raw else (None, None, None)
            s0, s1 = int(blk_off[b]), int(blk_off[b + 1])
            r.paths = [paths[int(seq_off[s]):int(seq_off[s + 1])] for s in range(s0, s1)] if paths is not None else None
            r.bg = None
            if bg is not None:
                g = BlockGraph()
                a, z = int(bg["no"][b]), int(bg["no"][b + 1])
                ln, od = bg["ln"][a:z], bg["od"][a:z]
                g.node_indeg = bg["idg"][a:z]
                so = int(bg["so"][b]) + np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
                g.node_seq = [bg["sq"][int(so[i]):int(so[i + 1])].decode() for i in range(z - a)]
                to = bg["to"][int(bg["eo"][b]):int(bg["eo"][b + 1])]
                g.edges = list(zip(np.repeat(np.arange(z - a), od).tolist(), to.tolist()))
                g.paths = [bg["st"][int(bg["po"][s]):int(bg["po"][s + 1])] for s in range(s0, s1)]
                g.consensus = bg["cs"][int(bg["co"][b]):int(bg["co"][b + 1])] if "co" in bg else None
                r.bg = g
            r.scores, r.cells = score[s0:s1], cells[s0:s1]
            r.device_cycles = int(cycles[b]) if cycles is not None else None   # (sxg_poa_batch_out::block_cycles)
            r.consensus = cons[cons_off[b]:cons_off[b + 1]] if cons is not None else None
            r.msa = None
            if msa_raw is not None and r.status == 0:
                ncol = int(msa_cols[b])
                raw = msa_raw[int(msa_off[b]):int(msa_off[b + 1])]
                r.msa = [raw[i * ncol:(i + 1) * ncol].decode() for i in range(len(raw) // ncol)] if ncol else []
            res.append(r)
        return res

    def run_flat(self, bases, seq_off, blk_off, weights, params, want_consensus=False, want_msa=False,
                 check=True, block_graph=0, bg_trim=None, bg_cons_visited_only=False):
        self.upload(bases, seq_off, blk_off, weights, params, want_consensus, want_msa, block_graph, bg_trim,
                    bg_cons_visited_only)
        self.execute(check=check)
        return self.download()

    # -- multi-GPU (one engine per rank) ---------------------------------------------------
    def comm_unique_id(self):
        """Rank 0: the id every rank passes to comm_init (send it through any side channel)."""
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        if self.lib.sxg_poa_comm_unique_id(buf):
            raise self._err("sxg_poa_comm_unique_id")
        return bytes(buf)

    def comm_init(self, comm_id, nranks, rank):
        buf = (C.c_uint8 * COMM_ID_BYTES)(*comm_id)
        if self.lib.sxg_poa_comm_init(self.h, buf, nranks, rank):
            raise self._err("sxg_poa_comm_init")

    def run_flat_sharded(self, bases, seq_off, blk_off, weights, params, want_consensus=False, want_msa=False, check=True,
                         simulate_ranks=0, block_graph=0, bg_trim=None, bg_cons_visited_only=False):
        """sxg_poa_batch_run_sharded: every rank passes the SAME batch; rank 0 gets all results (list), the others None.
        simulate_ranks > 0: the test entry that plays that many ranks on this one GPU."""
        bi = self._mk_in(bases, seq_off, blk_off, weights, params, want_consensus, want_msa, block_graph, bg_trim, bg_cons_visited_only)
        self._shape = (np.asarray(blk_off).copy(), np.asarray(seq_off).copy())
        out = BatchOut()
        if simulate_ranks:
            rc = self.lib.sxg_poa_batch_run_sharded_local(self.h, C.byref(bi), simulate_ranks, C.byref(out))
        else:
            rc = self.lib.sxg_poa_batch_run_sharded(self.h, C.byref(bi), C.byref(out))
        if rc == NOT_ROOT:
            return None
        try:
            if rc and (check or rc != -4):
                raise self._err("sxg_poa_batch_run_sharded")
            return self._unpack(out)
        finally:
            self.lib.sxg_poa_batch_free(C.byref(out))

    # -- the sharded run in stages (inputs stay resident between executes) ------------------
    def upload_sharded(self, bases, seq_off, blk_off, weights, params, want_consensus=False, want_msa=False):
        """Every rank, SAME batch: deal the blocks by cost over the communicator's ranks, upload this rank's share."""
        self._sh_in = self._mk_in(bases, seq_off, blk_off, weights, params, want_consensus, want_msa)
        self._shape = (np.asarray(blk_off).copy(), np.asarray(seq_off).copy())
        if self.lib.sxg_poa_batch_upload_sharded(self.h, C.byref(self._sh_in)):
            raise self._err("sxg_poa_batch_upload_sharded")

    def execute_sharded(self):
        """Collective: align the uploaded share, pack it, bring every rank's blob to rank 0 (RCCL)."""
        if self.lib.sxg_poa_batch_execute_sharded(self.h):
            raise self._err("sxg_poa_batch_execute_sharded")

    def download_sharded(self, check=True):
        """Rank 0: all results in the batch's block order; other ranks: None."""
        out = BatchOut()
        rc = self.lib.sxg_poa_batch_download_sharded(self.h, C.byref(self._sh_in), C.byref(out))
        if rc == NOT_ROOT:
            return None
        try:
            if rc and (check or rc != -4):
                raise self._err("sxg_poa_batch_download_sharded")
            return self._unpack(out)
        finally:
            self.lib.sxg_poa_batch_free(C.byref(out))

    def measure_copy(self, nbytes=1 << 30, reps=5):
        """GB/s (read + written) of a streaming device-to-device copy on this engine's device (sxg_poa_measure_copy)."""
        g = C.c_double()
        if self.lib.sxg_poa_measure_copy(self.h, C.c_uint64(int(nbytes)), int(reps), C.byref(g)):
            raise self._err("sxg_poa_measure_copy")
        return g.value

    def roctx_available(self):
        return int(self.lib.sxg_poa_roctx_available())

    def sharded_info(self):
        n, b = C.c_int32(), C.c_uint64()
        self.lib.sxg_poa_sharded_info(self.h, C.byref(n), C.byref(b))
        pk, ex = C.c_double(), C.c_double()
        self.lib.sxg_poa_sharded_timing(self.h, C.byref(pk), C.byref(ex))
        return {"ranks_seen": n.value, "bytes_received": b.value, "pack_ms": pk.value, "exchange_ms": ex.value}

    def run_blocks(self, blocks, params, weights=None, want_consensus=False, want_msa=False, check=True):
        """blocks: list of lists of uint8 code arrays (one inner list per block, alignment order)."""
        seqs = [s for blk in blocks for s in blk]
        bases = np.concatenate([np.asarray(s, np.uint8) for s in seqs]) if seqs else np.zeros(0, np.uint8)
        seq_off = np.zeros(len(seqs) + 1, np.int64)
        if seqs:
            seq_off[1:] = np.cumsum([len(s) for s in seqs])
        blk_off = np.zeros(len(blocks) + 1, np.int32)
        if blocks:
            blk_off[1:] = np.cumsum([len(b) for b in blocks])
        w = None if weights is None else np.concatenate([np.asarray(x, np.uint32) for x in weights])
        return self.run_flat(bases, seq_off, blk_off, w, params, want_consensus, want_msa, check)

    # -- stand-alone Align(sequence, graph) --------------------------------------------
    def align(self, problems, params, check=True):
        """problems: list of (codes, off, pred, sink, seq) with the CSR of oracle `rows()`.
        Returns list of (pair_row, pair_pos, score, status)."""
        n = len(problems)
        row_off = np.zeros(n + 1, np.int64)
        seq_off = np.zeros(n + 1, np.int64)
        codes, sinks, preds, poffs, bases = [], [], [], [np.zeros(1, np.int64)], []
        eo = 0
        for k, (cd, off, pr, sk, sq) in enumerate(problems):
            row_off[k + 1] = row_off[k] + len(cd)
            seq_off[k + 1] = seq_off[k] + len(sq)
            codes.append(np.asarray(cd, np.uint8))
            sinks.append(np.asarray(sk, np.uint8))
            preds.append(np.asarray(pr, np.int32))
            bases.append(np.asarray(sq, np.uint8))
            poffs.append(np.asarray(off[1:], np.int64) + eo)
            eo += int(off[-1]) if len(off) else 0

        def cat(xs, dt):
            return np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(0, dt), dt)

        def pad(a):
            return a if len(a) else np.zeros(1, a.dtype)

        codes, sinks, preds, bases = cat(codes, np.uint8), cat(sinks, np.uint8), cat(preds, np.int32), cat(bases, np.uint8)
        poff = cat(poffs, np.int64)
        if isinstance(params, Params):
            parr, per = (Params * 1)(params), 0
        else:
            parr, per = (Params * max(len(params), 1))(*params), 1
        codes, sinks, preds, bases = pad(codes), pad(sinks), pad(preds), pad(bases)
        ai = AlignIn(n, _p(row_off, C.c_int64), _p(codes, C.c_uint8), _p(sinks, C.c_uint8),
                     _p(poff, C.c_int64), _p(preds, C.c_int32), _p(seq_off, C.c_int64),
                     _p(bases, C.c_uint8), parr, per)
        out = AlignOut()
        rc = self.lib.sxg_poa_align_batch(self.h, C.byref(ai), C.byref(out))
        try:
            if rc and (check or rc != -4):
                raise self._err("sxg_poa_align_batch")
            status = _arr(out.status, n, np.int32)
            score = _arr(out.score, n, np.int32)
            po = _arr(out.pair_off, n + 1, np.int64)
            tot = int(po[-1]) if n else 0
            pr_ = _arr(out.pair_row, tot, np.int32)
            pp_ = _arr(out.pair_pos, tot, np.int32)
            return [(pr_[po[k]:po[k + 1]], pp_[po[k]:po[k + 1]], int(score[k]), int(status[k]))
                    for k in range(n)]
        finally:
            self.lib.sxg_poa_align_free(C.byref(out))


def xxh64(data: bytes, seed=0):
    return load_library().sxg_xxh64(data, len(data), seed)
